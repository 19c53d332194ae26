"""In-memory snapshot/restore of the batch state (reference ``mujoco_template/state_utils.py:9-31``)."""

from __future__ import annotations

from typing import Any

import numpy as np

from ._typing import StateSnapshot


def _snapshot_state(data: Any) -> StateSnapshot:
    return {
        "qpos": np.array(data.qpos),
        "qvel": np.array(data.qvel),
        "act": np.array(data.act) if hasattr(data, "act") else None,
        "ctrl": np.array(data.ctrl),
        "time": np.array(data.time, dtype=float).copy(),
    }


def _restore_state(data: Any, snap: StateSnapshot) -> None:
    data.qpos[...] = snap["qpos"]
    data.qvel[...] = snap["qvel"]
    data.ctrl[...] = snap["ctrl"]
    t = snap.get("time")
    data.time = 0.0 if t is None else (float(t) if np.ndim(t) == 0 else np.asarray(t, dtype=float))


__all__ = ["_snapshot_state", "_restore_state"]
