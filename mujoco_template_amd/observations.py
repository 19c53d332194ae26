"""What an environment reports after a step, declared once.

``ObservationSpec`` / ``ObservationProducer`` / ``ObservationExtractor`` carry the reference's names, fields and defaults
(``mujoco_template/observations.py:20-174``).  The extractor serves two consumers:

* **host** - ``extractor(data)`` returns the mapping ``field -> array`` (views that alias the data object's host mirrors
  unless ``copy`` asks otherwise; position blocks are always fresh arrays) or, with ``as_dict=False``, one flat vector
  laid out in ``sorted(field names)`` order.  With ``batch > 1`` every array has a leading batch axis and the flat form
  is ``[batch, obs_dim]``;
* **device** - ``gather_device`` / ``device_spec``: the same flat layout written by a gather kernel (``mjb_obs_gather``)
  or straight into the observation ring of a fused rollout, ready for the RCCL all-gather.

The state fields are described by one table (flag on the spec, attribute on the data object); name -> id resolution of
sites / bodies / geoms happens once, at construction, and fails with ``NameLookupError`` like the reference.
"""

from __future__ import annotations

import warnings
from collections.abc import Callable, Mapping, Sequence
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from . import mj
from ._typing import Observation, ObservationDict
from .exceptions import ConfigError, NameLookupError

# (spec flag, observation key = attribute of the data object) for the plain state arrays, in the reference's order
_STATE_FIELDS = (("include_qpos", "qpos"), ("include_qvel", "qvel"), ("include_act", "act"), ("include_ctrl", "ctrl"))
# bit of each flag in the device gather kernel (include/mjbatch.h mjb_obs_spec_create)
_DEVICE_BITS = {"include_qpos": 0, "include_qvel": 1, "include_ctrl": 2, "include_sensordata": 3, "include_time": 4, "bodies_inertial": 6}


@dataclass(frozen=True)
class ObservationProducer:
    """A user-computed observation entry; ``copy=None`` defers to the spec-wide ``copy`` flag."""

    fn: Callable[[Any, Any], np.ndarray | Sequence[float]]
    copy: bool | None = None

    def produce(self, model: Any, data: Any, default_copy: bool) -> np.ndarray:
        raw = self.fn(model, data)
        arr = raw if isinstance(raw, np.ndarray) else np.asarray(raw)
        detach = default_copy if self.copy is None else bool(self.copy)
        return arr.copy() if detach else arr


@dataclass
class ObservationSpec:
    include_qpos: bool = True
    include_qvel: bool = True
    include_act: bool = False
    include_ctrl: bool = False
    include_sensordata: bool = False
    include_time: bool = False
    sites_pos: Sequence[str] = field(default_factory=tuple)
    bodies_pos: Sequence[str] = field(default_factory=tuple)
    geoms_pos: Sequence[str] = field(default_factory=tuple)
    subtree_com: Sequence[str] = field(default_factory=tuple)
    as_dict: bool = True
    bodies_inertial: bool = False
    extras: Mapping[str, ObservationProducer | Callable[[Any, Any], np.ndarray | Sequence[float]]] = field(default_factory=dict)
    copy: bool = False


def _as_producer(key: str, value: Any) -> ObservationProducer:
    if isinstance(value, ObservationProducer):
        return value
    if callable(value):
        return ObservationProducer(value)
    raise TypeError(f"extras[{key!r}] must be callable or ObservationProducer")


class ObservationExtractor:
    def __init__(self, model: Any, spec: ObservationSpec):
        self.model, self.spec = model, spec

        def ids(kind: int, names: Sequence[str]) -> tuple[int, ...]:
            found = []
            for name in names:
                idx = int(mj.mj_name2id(model, kind, name))
                if idx < 0:
                    raise NameLookupError(f"Name not found in model: {name}")
                found.append(idx)
            return tuple(found)

        self.site_ids = ids(mj.mjtObj.mjOBJ_SITE, spec.sites_pos)
        self.body_ids = ids(mj.mjtObj.mjOBJ_BODY, spec.bodies_pos)
        self.geom_ids = ids(mj.mjtObj.mjOBJ_GEOM, spec.geoms_pos)
        self.subtree_ids = ids(mj.mjtObj.mjOBJ_BODY, spec.subtree_com)
        self.extra_items = tuple((key, _as_producer(key, value)) for key, value in spec.extras.items())
        self._warned_missing_sensordata = False
        self._dev_spec = None
        self._dev_out = None

    # -- host ------------------------------------------------------------------------------------
    def _sensordata(self, data: Any, batched: bool) -> np.ndarray:
        if self.model.nsensordata > 0:
            return data.sensordata.copy() if self.spec.copy else data.sensordata
        if not self._warned_missing_sensordata:
            warnings.warn("ObservationSpec requested sensordata but model has none; returning an empty array instead.", RuntimeWarning)
            self._warned_missing_sensordata = True
        return np.zeros((data.batch, 0)) if batched else np.zeros(0, dtype=float)

    def _position_blocks(self, data: Any) -> ObservationDict:
        """``[k, 3]`` (``[batch, k, 3]``) selections of the kinematic outputs - always fresh arrays."""
        spec = self.spec
        if self.subtree_ids:
            mj.mj_subtreeCoM(self.model, data)
        sources = (("sites_pos", self.site_ids, "site_xpos"),
                   ("bodies_pos", self.body_ids, "xipos" if spec.bodies_inertial else "xpos"),
                   ("geoms_pos", self.geom_ids, "geom_xpos"),
                   ("subtree_com", self.subtree_ids, "subtree_com"))
        return {key: np.array(getattr(data, attr)[..., list(sel), :], dtype=float) for key, sel, attr in sources if sel}

    def __call__(self, data: Any) -> Observation:
        spec = self.spec
        batched = getattr(data, "batch", 1) > 1
        out: ObservationDict = {}
        for flag, key in _STATE_FIELDS:
            if getattr(spec, flag):
                arr = getattr(data, key)
                out[key] = arr.copy() if spec.copy else arr
        if spec.include_sensordata:
            out["sensordata"] = self._sensordata(data, batched)
        if spec.include_time:
            clock = np.array(data.time, dtype=float)
            out["time"] = clock.reshape(-1, 1) if batched else clock.reshape(1)
        out.update(self._position_blocks(data))
        for key, producer in self.extra_items:
            if key in out:
                raise ValueError(f"extras[{key!r}] duplicates an existing observation key")
            out[key] = producer.produce(self.model, data, spec.copy)
        if spec.as_dict:
            return out
        if not out:
            return np.zeros(0)
        parts = [np.asarray(out[key]) for key in sorted(out)]
        if batched:
            return np.concatenate([part.reshape(data.batch, -1) for part in parts], axis=1)
        return np.concatenate([part.ravel() for part in parts])

    # -- device path ------------------------------------------------------------------------
    def device_flags(self) -> int:
        if self.extra_items:
            raise ConfigError("user extras are host callables: use the host extractor for specs with extras")
        return sum(1 << bit for name, bit in _DEVICE_BITS.items() if getattr(self.spec, name))

    def device_spec(self, data: Any):
        if self._dev_spec is None or self._dev_spec.sim is not data.sim:
            self._dev_spec = data.sim.make_obs_spec(self.device_flags(), self.site_ids, self.body_ids, self.geom_ids, self.subtree_ids)
        return self._dev_spec

    @property
    def obs_dim(self) -> int:
        m, s = self.model, self.spec
        return (3 * (len(self.site_ids) + len(self.body_ids) + len(self.geom_ids) + len(self.subtree_ids)) + m.nq * int(s.include_qpos)
                + m.nv * int(s.include_qvel) + m.nu * int(s.include_ctrl) + m.nsensordata * int(s.include_sensordata) + int(s.include_time))

    def gather_device(self, data: Any, out=None):
        """Flat observation ``[batch, obs_dim]`` as a torch tensor on the data's GPU (dtype of the data)."""
        import torch

        spec = self.device_spec(data)
        if out is None:
            if self._dev_out is None or self._dev_out.shape != (data.batch, spec.dim):
                self._dev_out = torch.empty((data.batch, spec.dim), device=f"cuda:{data.sim.device}",
                                            dtype=torch.float32 if data.sim.dtype == "float32" else torch.float64)
            out = self._dev_out
        data.push_host_edits()
        data.sim.obs_gather(spec, out.data_ptr())
        return out


__all__ = ["ObservationSpec", "ObservationExtractor", "ObservationProducer"]
