"""``ModelHandle`` - the object that owns one compiled model and its (batched) data, and the only place where ``Env``
touches the engine's lifecycle calls.

Public surface = the reference's (``mujoco_template/model.py:11-105``): construction from XML, ``forward`` / ``step`` /
``reset`` / ``reset_keyframe``, actuator-group enabling through ``opt.disableactuator``.  Differences that come with
the engine: the constructors forward keyword-only creation arguments (``batch``, ``dtype``, ``device``, ``lanes``,
``nconmax``, ``nefcmax``, ``env0``, ``specialize``) to ``MjData``; every physics call is ONE C-ABI call for the whole batch;
``from_binary_path`` / ``save_binary`` use the engine's own flat model table (``mjb_model_save`` / ``mjb_model_load``), so a host
without the Python MJCF compiler can load a model compiled once.
"""

from __future__ import annotations

from collections.abc import Iterable

import numpy as np

from . import mj
from .exceptions import CompatibilityError, ConfigError, NameLookupError

_MAX_GROUP = 31          # opt.disableactuator is a 31-bit mask


class ModelHandle:
    # -- construction -----------------------------------------------------------------------------
    def __init__(self, model: mj.MjModel, data: mj.MjData | None = None, **data_kwargs):
        if data is not None and data.model is not model:
            raise ConfigError("Provided mj.MjData must reference the supplied model.")
        self.model = model
        self.data = data if data is not None else mj.MjData(model, **data_kwargs)

    @classmethod
    def from_xml_path(cls, xml_path: str, **data_kwargs) -> "ModelHandle":
        return cls(mj.MjModel.from_xml_path(xml_path), **data_kwargs)

    @classmethod
    def from_xml_string(cls, xml_text: str, **data_kwargs) -> "ModelHandle":
        return cls(mj.MjModel.from_xml_string(xml_text), **data_kwargs)

    @classmethod
    def from_model_and_data(cls, model: mj.MjModel, data: mj.MjData) -> "ModelHandle":
        """Adopt an existing pair (no new device buffers)."""
        return cls(model, data=data)

    @classmethod
    def from_binary_path(cls, mjb_path: str, **data_kwargs) -> "ModelHandle":
        """A model written by :meth:`save_binary` (C ABI ``mjb_model_load``): this engine's own flat table format, not MuJoCo's .mjb."""
        return cls(mj.MjModel.from_binary_path(mjb_path), **data_kwargs)

    def save_binary(self, mjb_path: str) -> None:
        try:
            mj.mj_saveModel(self.model, mjb_path, None)
        except Exception as exc:
            raise ConfigError(f"Could not save model to {mjb_path}: {exc}") from exc

    # -- what was built -----------------------------------------------------------------------------
    @property
    def batch(self) -> int:
        """Number of independent environments that share this model."""
        return int(getattr(self.data, "batch", 1))

    @property
    def sim(self):
        """The ``BatchSim`` behind ``data`` (device pointers, stream, counters, fused rollouts)."""
        return self.data.sim

    def describe(self) -> dict[str, object]:
        """Creation facts worth logging next to a benchmark number."""
        sim = self.sim
        return {"batch": self.batch, "dtype": sim.dtype, "device": sim.device, "lanes_per_env": sim.lanes,
                "nconmax": sim.nconmax, "nefcmax": sim.nefcmax, "lds_bytes_per_env": sim.lds_bytes_per_env,
                "specialized_kernel": bool(sim.specialized), "nq": self.model.nq, "nv": self.model.nv, "nu": self.model.nu}

    def keyframe_names(self) -> list[str | None]:
        return [mj.mj_id2name(self.model, mj.mjtObj.mjOBJ_KEY, k) for k in range(self.model.nkey)]

    # -- lifecycle: one launch per call, whole batch -----------------------------------------------
    def forward(self) -> None:
        mj.mj_forward(self.model, self.data)

    def step(self, nstep: int = 1) -> None:
        mj.mj_step(self.model, self.data, nstep)

    def reset(self) -> None:
        mj.mj_resetData(self.model, self.data)

    def reset_keyframe(self, key: int | str) -> None:
        mj.mj_resetDataKeyframe(self.model, self.data, self._keyframe_index(key))

    def _keyframe_index(self, key: int | str) -> int:
        if isinstance(key, str):
            found = mj.mj_name2id(self.model, mj.mjtObj.mjOBJ_KEY, key)
            if found < 0:
                raise NameLookupError(f"Keyframe name not found: {key}")
            return found
        index = int(key)
        if index < 0 or index >= self.model.nkey:
            raise ConfigError(f"Keyframe index out of range: {index}")
        return index

    # -- actuator groups ------------------------------------------------------------------------------
    @property
    def actuator_groups(self) -> np.ndarray:
        return np.array(self.model.actuator_group, dtype=int)

    def enabled_actuator_mask(self) -> np.ndarray:
        """Boolean ``[nu]``: actuator not switched off by ``opt.disableactuator``."""
        bits = (int(self.model.opt.disableactuator) >> self.actuator_groups) & 1
        return bits == 0

    def set_enabled_actuator_groups(self, enabled_groups: Iterable[int]) -> None:
        """Enable exactly the listed groups (all other groups present in the model are disabled), then re-propagate."""
        keep = {int(g) for g in enabled_groups}
        if not keep:
            raise CompatibilityError("At least one actuator group must be enabled.")
        if min(keep) < 0 or max(keep) > _MAX_GROUP:
            raise ConfigError(f"Actuator groups must be in [0, {_MAX_GROUP}].")
        if self.model.nu == 0:
            raise CompatibilityError("Model has no actuators (nu=0).")
        present = {int(g) for g in self.actuator_groups}
        if keep.isdisjoint(present):
            raise CompatibilityError("None of the requested groups exist in this model.")
        self.model.opt.disableactuator = sum(1 << g for g in present - keep)
        self.forward()
        if not self.enabled_actuator_mask().any():
            raise CompatibilityError("All actuators disabled by group selection.")


__all__ = ["ModelHandle"]
