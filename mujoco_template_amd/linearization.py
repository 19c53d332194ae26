"""Discrete linearisation ``x' = A x + B u`` with ``x = (dq, qvel)`` in tangent space
(reference ``mujoco_template/linearization.py:16-135``).

``use_native=True`` runs the batched ``mjd_transitionFD`` on the GPU in float64: every
perturbed replica (2(2nv+nu)+1 per environment) is one lane-group of one launch.
``use_native=False`` reproduces the reference's Python fallback loop shape (``horizon_steps``)
on top of the engine.  Note: the reference fallback's ``_dqpos`` passes (after, base) to
``mj_differentiatePos`` which yields ``(base - after)``, i.e. sign-flipped position rows versus
the native path; this implementation uses the native convention for both.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from . import mj
from .exceptions import LinearizationError
from .state_utils import _restore_state, _snapshot_state


def _native_transition_fd(model: Any, data: Any, eps: float = 1e-6, centered: bool = True) -> tuple[np.ndarray, np.ndarray]:
    nv, nu, B = model.nv, model.nu, data.batch
    nx = 2 * nv
    A = np.zeros((nx, nx)) if B == 1 else np.zeros((B, nx, nx))
    Bm = np.zeros((nx, nu)) if B == 1 else np.zeros((B, nx, nu))
    try:
        mj.mjd_transitionFD(model, data, float(eps), bool(centered), A, Bm, None, None)
    except TypeError as exc:
        raise LinearizationError(f"mjd_transitionFD failed: {exc}") from exc
    return A, Bm


def _fd_linearization(model: Any, data: Any, eps: float = 1e-6, horizon_steps: int = 1) -> tuple[np.ndarray, np.ndarray]:
    if data.batch != 1:
        raise LinearizationError("the Python FD fallback supports batch=1; use use_native=True for batches")
    nv, nu = model.nv, model.nu
    nx = 2 * nv
    snap = _snapshot_state(data)
    base_q, base_v, base_u = np.array(data.qpos), np.array(data.qvel), np.array(data.ctrl)

    def rollout() -> np.ndarray:
        mj.mj_step(model, data, horizon_steps)
        dq = np.zeros(nv)
        mj.mj_differentiatePos(model, dq, 1.0, base_q, np.array(data.qpos))
        return np.concatenate([dq, np.array(data.qvel) - base_v])

    def perturbed(kind: str, idx: int, sign: float) -> np.ndarray:
        _restore_state(data, snap)
        if kind == "q":
            q = base_q.copy()
            e = np.zeros(nv)
            e[idx] = sign * eps
            mj.mj_integratePos(model, q, e, 1.0)
            data.qpos[...] = q
        elif kind == "v":
            data.qvel[idx] += sign * eps
        else:
            data.ctrl[idx] = base_u[idx] + sign * eps
        return rollout()

    try:
        A = np.zeros((nx, nx))
        Bm = np.zeros((nx, nu))
        for i in range(nv):
            A[:, i] = (perturbed("q", i, 1.0) - perturbed("q", i, -1.0)) / (2.0 * eps)
            A[:, nv + i] = (perturbed("v", i, 1.0) - perturbed("v", i, -1.0)) / (2.0 * eps)
        for i in range(nu):
            Bm[:, i] = (perturbed("u", i, 1.0) - perturbed("u", i, -1.0)) / (2.0 * eps)
        return A, Bm
    finally:
        _restore_state(data, snap)
        mj.mj_forward(model, data)


def linearize_discrete(model: Any, data: Any, use_native: bool = True, eps: float = 1e-6, horizon_steps: int = 1) -> tuple[np.ndarray, np.ndarray]:
    if use_native:
        try:
            return _native_transition_fd(model, data, eps=eps, centered=True)
        except LinearizationError:
            pass
    return _fd_linearization(model, data, eps=eps, horizon_steps=horizon_steps)


__all__ = ["linearize_discrete"]
