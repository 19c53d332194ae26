"""Equilibrium controls by inverse dynamics, for one state or a whole batch.

``steady_ctrl0(model, data, qpos0, qvel0=None)`` has the reference's signature, argument checks and exception types
(``mujoco_template/setpoints.py:10-58``): put the system at ``(qpos0, qvel0)``, ask for zero acceleration, read the
generalized force that inverse dynamics says is needed, and realise it with the actuators through the pseudo-inverse of
the actuator moment matrix.  Here the inverse dynamics AND the dense moment matrix come from one device pass
(``mjb_inverse``); the ``nu x nv`` pseudo-inverse and the conditioning guard are evaluated for all environments at once
with stacked ``numpy.linalg`` calls.  ``qpos0`` / ``qvel0`` may be single vectors (applied to every environment) or carry a
batch axis; the result is ``[nu]`` for ``batch == 1`` and ``[batch, nu]`` otherwise.  The caller's state is restored.
"""

from __future__ import annotations

import numpy as np

from . import mj
from .exceptions import CompatibilityError, ConfigError, TemplateError
from .state_utils import _restore_state, _snapshot_state

_RCOND_GUARD = 1e-12          # smallest / largest singular value below this = unusable moment matrix


def _checked(vec, n: int, what: str) -> np.ndarray:
    arr = np.asarray(vec, dtype=float)
    if arr.shape[-1] != n:
        raise ConfigError(f"{what} must have length model.{'nq' if what == 'qpos0' else 'nv'}")
    return arr


def _solve_for_ctrl(moment: np.ndarray, qfrc: np.ndarray) -> np.ndarray:
    """``u`` with ``u @ moment = qfrc`` in the least-squares sense, per environment: ``[B, nu, nv]``, ``[B, nv]`` -> ``[B, nu]``."""
    sv = np.linalg.svd(moment, compute_uv=False)                               # [B, min(nu, nv)]
    worst = np.where(sv.max(axis=-1) > 0, sv.min(axis=-1) / np.maximum(sv.max(axis=-1), 1e-300), 0.0)
    if sv.shape[-1] == 0 or np.any(worst < _RCOND_GUARD):
        raise TemplateError("Actuator moment matrix is singular or ill-conditioned at this state.")
    return np.einsum("bv,bvu->bu", qfrc, np.linalg.pinv(moment))


def steady_ctrl0(model: "mj.MjModel", data: "mj.MjData", qpos0: np.ndarray, qvel0: np.ndarray | None = None) -> np.ndarray:
    qpos0 = _checked(qpos0, model.nq, "qpos0")
    qvel0 = np.zeros(model.nv) if qvel0 is None else _checked(qvel0, model.nv, "qvel0")
    batch = int(getattr(data, "batch", 1))
    saved = _snapshot_state(data)
    try:
        mj.mj_resetData(model, data)
        data.qpos[...] = qpos0
        data.qvel[...] = qvel0
        mj.mj_forward(model, data)
        data.qacc[...] = 0.0
        mj.mj_inverse(model, data)                      # fills qfrc_inverse and the dense actuator moment
        if model.nu == 0:
            raise CompatibilityError("No actuators to realize inverse dynamics (nu=0).")
        qfrc = np.array(data.qfrc_inverse, dtype=float).reshape(batch, model.nv)
        moment = np.array(data.actuator_moment, dtype=float).reshape(batch, model.nu, model.nv)
        ctrl = _solve_for_ctrl(moment, qfrc)
        return ctrl[0] if batch == 1 else ctrl
    finally:
        _restore_state(data, saved)
        mj.mj_forward(model, data)


__all__ = ["steady_ctrl0"]
