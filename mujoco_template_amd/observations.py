"""Declarative observations (reference ``mujoco_template/observations.py:34-174``).

``ObservationExtractor.__call__`` keeps the reference's host semantics (dict of zero-copy
views, or a flat array concatenated in ``sorted(keys)`` order) over the batched data proxy.
``gather_device`` is the batched fast path: one HIP gather kernel writes the same flat
layout ``[batch, obs_dim]`` on the GPU (``mjb_obs_gather``), ready for an RCCL all-gather.
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


@dataclass(frozen=True)
class ObservationProducer:
    """User-defined observation slice; ``copy=None`` inherits the spec-wide flag."""

    fn: Callable[[Any, Any], np.ndarray | Sequence[float]]
    copy: bool | None = None

    def produce(self, model: Any, data: Any, default_copy: bool) -> np.ndarray:
        value = self.fn(model, data)
        want_copy = default_copy if self.copy is None else bool(self.copy)
        arr = value if isinstance(value, np.ndarray) else np.asarray(value)
        return np.array(arr, copy=True) if want_copy else arr


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


def _maybe_copy(arr: np.ndarray, copy: bool) -> np.ndarray:
    return np.array(arr, copy=True) if copy else arr


class ObservationExtractor:
    def __init__(self, model: Any, spec: ObservationSpec):
        self.model = model
        self.spec = spec
        self.site_ids = tuple(self._name2id(mj.mjtObj.mjOBJ_SITE, n) for n in spec.sites_pos)
        self.body_ids = tuple(self._name2id(mj.mjtObj.mjOBJ_BODY, n) for n in spec.bodies_pos)
        self.geom_ids = tuple(self._name2id(mj.mjtObj.mjOBJ_GEOM, n) for n in spec.geoms_pos)
        self.subtree_ids = tuple(self._name2id(mj.mjtObj.mjOBJ_BODY, n) for n in spec.subtree_com)
        self.extra_items = tuple((name, self._normalize_extra(name, p)) for name, p in spec.extras.items())
        self._warned_missing_sensordata = False
        self._dev_spec = None
        self._dev_out = None

    def _name2id(self, objtype: int, name: str) -> int:
        idx = int(mj.mj_name2id(self.model, objtype, name))
        if idx < 0:
            raise NameLookupError(f"Name not found in model: {name}")
        return idx

    @staticmethod
    def _normalize_extra(name: str, producer: Any) -> ObservationProducer:
        if isinstance(producer, ObservationProducer):
            return producer
        if callable(producer):
            return ObservationProducer(producer)
        raise TypeError(f"extras[{name!r}] must be callable or ObservationProducer")

    # -- host path (reference semantics) ------------------------------------------------
    def __call__(self, data: Any) -> Observation:
        spec = self.spec
        batched = getattr(data, "batch", 1) > 1
        out: ObservationDict = {}
        if spec.include_qpos:
            out["qpos"] = _maybe_copy(data.qpos, spec.copy)
        if spec.include_qvel:
            out["qvel"] = _maybe_copy(data.qvel, spec.copy)
        if spec.include_act:
            out["act"] = _maybe_copy(data.act, spec.copy)
        if spec.include_ctrl:
            out["ctrl"] = _maybe_copy(data.ctrl, spec.copy)
        if spec.include_sensordata:
            if self.model.nsensordata == 0:
                if not self._warned_missing_sensordata:
                    warnings.warn("ObservationSpec requested sensordata but model has none; returning an empty array instead.", RuntimeWarning)
                    self._warned_missing_sensordata = True
                out["sensordata"] = np.zeros((data.batch, 0)) if batched else np.zeros(0, dtype=float)
            else:
                out["sensordata"] = _maybe_copy(data.sensordata, spec.copy)
        if spec.include_time:
            out["time"] = np.array(data.time, dtype=float).reshape(-1, 1) if batched else np.array([data.time], dtype=float)

        def rows(src: np.ndarray, ids: tuple[int, ...]) -> np.ndarray:   # always a fresh array, like the reference
            return np.array(src[..., list(ids), :], dtype=float)

        if self.site_ids:
            out["sites_pos"] = rows(data.site_xpos, self.site_ids)
        if self.body_ids:
            out["bodies_pos"] = rows(data.xipos if spec.bodies_inertial else data.xpos, self.body_ids)
        if self.geom_ids:
            out["geoms_pos"] = rows(data.geom_xpos, self.geom_ids)
        if self.subtree_ids:
            mj.mj_subtreeCoM(self.model, data)
            out["subtree_com"] = rows(data.subtree_com, self.subtree_ids)
        for name, producer in self.extra_items:
            if name in out:
                raise ValueError(f"extras[{name!r}] duplicates an existing observation key")
            out[name] = producer.produce(self.model, data, spec.copy)
        if spec.as_dict:
            return out
        keys = sorted(out.keys())
        if not keys:
            return np.zeros(0)
        if batched:
            return np.concatenate([np.asarray(out[k]).reshape(data.batch, -1) for k in keys], axis=1)
        return np.concatenate([np.asarray(out[k]).ravel() for k in keys])

    # -- device path ------------------------------------------------------------------------
    def device_flags(self) -> int:
        s = self.spec
        if self.extra_items:
            raise ConfigError("user extras are host callables: use the host extractor for specs with extras")
        return (int(s.include_qpos) | int(s.include_qvel) << 1 | int(s.include_ctrl) << 2 | int(s.include_sensordata) << 3
                | int(s.include_time) << 4 | int(s.bodies_inertial) << 6)

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
