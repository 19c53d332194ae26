"""Equilibrium controls by inverse dynamics (reference ``mujoco_template/setpoints.py:10-58``), batched.

``steady_ctrl0(model, data, qpos0, qvel0=None)`` keeps the reference's signature, checks and exceptions.  With a
batched ``MjData`` the set-point may be one ``[nq]`` vector (applied to every environment) or ``[batch, nq]``;
the result is ``[nu]`` for ``batch == 1`` and ``[batch, nu]`` otherwise.  The inverse dynamics and the dense
actuator moment come from ONE device pass (``mjb_inverse``); the ``nu x nv`` pseudo-inverse stays on the host,
as in the reference (design-time code, a few hundred flops per environment).
"""

from __future__ import annotations

import numpy as np

from . import mj
from .exceptions import CompatibilityError, ConfigError, TemplateError
from .state_utils import _restore_state, _snapshot_state


def steady_ctrl0(model: "mj.MjModel", data: "mj.MjData", qpos0: np.ndarray, qvel0: np.ndarray | None = None) -> np.ndarray:
    qpos0 = np.asarray(qpos0, dtype=float)
    if qpos0.shape[-1] != model.nq:
        raise ConfigError("qpos0 must have length model.nq")
    if qvel0 is None:
        qvel0 = np.zeros(model.nv)
    qvel0 = np.asarray(qvel0, dtype=float)
    if qvel0.shape[-1] != model.nv:
        raise ConfigError("qvel0 must have length model.nv")

    snap = _snapshot_state(data)
    try:
        mj.mj_resetData(model, data)
        data.qpos[...] = qpos0
        data.qvel[...] = qvel0
        mj.mj_forward(model, data)
        data.qacc[...] = 0.0
        mj.mj_inverse(model, data)
        qfrc = np.array(data.qfrc_inverse, dtype=float)

        if model.nu == 0:
            raise CompatibilityError("No actuators to realize inverse dynamics (nu=0).")

        batch = getattr(data, "batch", 1)
        moments = np.array(data.actuator_moment, dtype=float).reshape(batch, model.nu, model.nv)
        qf = qfrc.reshape(batch, model.nv)
        out = np.zeros((batch, model.nu))
        for e in range(batch):
            M = np.zeros((model.nu, model.nv))
            mj.mju_sparse2dense(M, moments[e].reshape(-1), np.full(model.nu, model.nv), np.arange(model.nu) * model.nv,
                                np.tile(np.arange(model.nv), model.nu))
            s = np.linalg.svd(M, compute_uv=False)
            cond_guard = (s.size == 0) or ((s.min() / s.max()) if (s.max() > 0) else 0.0) < 1e-12
            if cond_guard:
                raise TemplateError("Actuator moment matrix is singular or ill-conditioned at this state.")
            out[e] = (np.atleast_2d(qf[e]) @ np.linalg.pinv(M)).ravel()
        return out[0] if batch == 1 else out
    finally:
        _restore_state(data, snap)
        mj.mj_forward(model, data)


__all__ = ["steady_ctrl0"]
