"""Save and put back the mutable state of a (batched) ``MjData`` - used around every computation that has to
leave the simulation untouched (``steady_ctrl0``, finite differences on the host).

Same two private helpers as the reference (``mujoco_template/state_utils.py:9-31``); the fields are whatever
``_FIELDS`` lists, each copied with its batch axis, and ``time`` is restored through the data object's setter so the
device copy follows.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from ._typing import StateSnapshot

_FIELDS = ("qpos", "qvel", "ctrl")          # writable arrays that exist on every data object
_OPTIONAL = ("act",)                          # present only for models with actuator state (na > 0)


def _snapshot_state(data: Any) -> StateSnapshot:
    snap: StateSnapshot = {name: np.array(getattr(data, name)) for name in _FIELDS}
    for name in _OPTIONAL:
        snap[name] = np.array(getattr(data, name)) if hasattr(data, name) else None
    snap["time"] = np.array(data.time, dtype=float).copy()
    return snap


def _restore_state(data: Any, snap: StateSnapshot) -> None:
    for name in _FIELDS:
        getattr(data, name)[...] = snap[name]
    when = snap.get("time")
    if when is None:
        data.time = 0.0
    else:
        data.time = float(when) if np.ndim(when) == 0 else np.asarray(when, dtype=float)


__all__ = ["_snapshot_state", "_restore_state"]
