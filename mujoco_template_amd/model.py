"""``ModelHandle`` — owner of (model, data); the seam where ``Env`` meets the engine.

Mirrors reference ``mujoco_template/model.py:11-105`` (same method names, arguments and
errors) with the batched engine underneath: ``step``/``forward``/``reset`` are one C-ABI call
each for the whole batch.  ``from_xml_*`` gain keyword-only ``batch``/``dtype``/``device``.
"""

from __future__ import annotations

from collections.abc import Iterable

import numpy as np

from . import mj
from .exceptions import CompatibilityError, ConfigError, NameLookupError


class ModelHandle:
    def __init__(self, model: mj.MjModel, data: mj.MjData | None = None, **data_kwargs):
        self.model = model
        if data is None:
            data = mj.MjData(model, **data_kwargs)
        elif data.model is not model:
            raise ConfigError("Provided mj.MjData must reference the supplied model.")
        self.data = data

    @classmethod
    def from_xml_path(cls, xml_path: str, **data_kwargs) -> "ModelHandle":
        return cls(mj.MjModel.from_xml_path(xml_path), **data_kwargs)

    @classmethod
    def from_xml_string(cls, xml_text: str, **data_kwargs) -> "ModelHandle":
        return cls(mj.MjModel.from_xml_string(xml_text), **data_kwargs)

    @classmethod
    def from_binary_path(cls, mjb_path: str) -> "ModelHandle":
        raise ConfigError("This build has no from_binary_path(): compile from XML instead.")

    @classmethod
    def from_model_and_data(cls, model: mj.MjModel, data: mj.MjData) -> "ModelHandle":
        return cls(model, data=data)

    def save_binary(self, mjb_path: str) -> None:
        raise ConfigError("This build has no mj_saveModel().")

    # -- physics: one launch per call for the whole batch ---------------------------
    def forward(self) -> None:
        mj.mj_forward(self.model, self.data)

    def step(self, nstep: int = 1) -> None:
        mj.mj_step(self.model, self.data, nstep)

    def reset(self) -> None:
        mj.mj_resetData(self.model, self.data)

    def reset_keyframe(self, key: int | str) -> None:
        if isinstance(key, str):
            idx = mj.mj_name2id(self.model, mj.mjtObj.mjOBJ_KEY, key)
            if idx < 0:
                raise NameLookupError(f"Keyframe name not found: {key}")
        else:
            idx = int(key)
            if not 0 <= idx < self.model.nkey:
                raise ConfigError(f"Keyframe index out of range: {idx}")
        mj.mj_resetDataKeyframe(self.model, self.data, idx)

    # -- actuator groups ---------------------------------------------------------------
    @property
    def actuator_groups(self) -> np.ndarray:
        return np.array(self.model.actuator_group, dtype=int)

    def set_enabled_actuator_groups(self, enabled_groups: Iterable[int]) -> None:
        wanted = [int(g) for g in enabled_groups]
        if not wanted:
            raise CompatibilityError("At least one actuator group must be enabled.")
        if any(g < 0 or g > 31 for g in wanted):
            raise ConfigError("Actuator groups must be in [0, 31].")
        if self.model.nu == 0:
            raise CompatibilityError("Model has no actuators (nu=0).")
        present = {int(g) for g in self.model.actuator_group[: self.model.nu]}
        if not present.intersection(wanted):
            raise CompatibilityError("None of the requested groups exist in this model.")
        mask = 0
        for grp in present.difference(wanted):
            mask |= 1 << grp
        self.model.opt.disableactuator = mask
        mj.mj_forward(self.model, self.data)
        if self.enabled_actuator_mask().sum() == 0:
            raise CompatibilityError("All actuators disabled by group selection.")

    def enabled_actuator_mask(self) -> np.ndarray:
        disabled = int(self.model.opt.disableactuator)
        return np.array([not ((disabled >> int(g)) & 1) for g in self.actuator_groups], dtype=bool)


__all__ = ["ModelHandle"]
