"""Built-in controllers.  ``ZeroController`` and ``PositionTargetDemo`` follow the reference
(``mujoco_template/controllers.py:12-46``); ``RandomCtrlController`` is the synthetic
random-ctrl driver of BASELINE.json's rollout configs.  Zero and random run on the device
(``device_ctrl_mode``), and their host ``__call__`` writes exactly the same values.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np

from .control import ControlSpace, ControllerCapabilities
from .exceptions import CompatibilityError, ConfigError, TemplateError


def _ctrl_rows(data: Any) -> np.ndarray:
    ctrl = data.ctrl
    return ctrl.reshape(1, -1) if ctrl.ndim == 1 else ctrl


@dataclass
class ZeroController:
    capabilities: ControllerCapabilities = ControllerCapabilities(control_space=ControlSpace.TORQUE)
    device_ctrl_mode: str = "zero"

    def prepare(self, model: Any, data: Any) -> None:
        if model.nu == 0:
            raise CompatibilityError("ZeroController requires nu>0 to write controls.")

    def __call__(self, model: Any, data: Any, t: float) -> None:
        if _ctrl_rows(data).shape[-1] != model.nu:
            raise TemplateError("data.ctrl size does not match model.nu")
        data.ctrl[...] = 0.0


@dataclass
class PositionTargetDemo:
    targets: np.ndarray | None = None
    capabilities: ControllerCapabilities = ControllerCapabilities(control_space=ControlSpace.POSITION)

    def prepare(self, model: Any, data: Any) -> None:
        if model.nu == 0:
            raise CompatibilityError("PositionTargetDemo requires nu>0.")
        if self.targets is None:
            row = _ctrl_rows(data)[0]
            self.targets = np.array(row) if row.size == model.nu else np.zeros(model.nu)
        if self.targets.shape[-1] != model.nu:
            raise ConfigError("targets must have length model.nu")

    def __call__(self, model: Any, data: Any, t: float) -> None:
        if _ctrl_rows(data).shape[-1] != model.nu:
            raise TemplateError("data.ctrl size does not match model.nu")
        data.ctrl[...] = self.targets


def philox_uniform(seed: int, env: np.ndarray, step: int, nu: int) -> np.ndarray:
    """u[env, actuator] in [0,1): Philox4x32-10 keyed by (seed, 0x5EED), counter (env, step, actuator, 0).

    Bit-identical to ``philox_first`` in ``csrc/mjb_device.hpp``.
    """
    c0 = np.repeat(np.asarray(env, dtype=np.uint64)[:, None], nu, axis=1) & 0xFFFFFFFF
    c1 = np.full_like(c0, step & 0xFFFFFFFF)
    c2 = np.broadcast_to(np.arange(nu, dtype=np.uint64)[None, :], c0.shape).copy()
    c3 = np.zeros_like(c0)
    k0, k1 = seed & 0xFFFFFFFF, 0x5EED
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return (c0 >> np.uint64(8)).astype(np.float64) / 16777216.0


@dataclass
class RandomCtrlController:
    """ctrl[a] = mid + half * scale * (2u - 1), u ~ Philox(seed; global env index, step, a)."""

    seed: int = 0
    scale: float = 1.0
    capabilities: ControllerCapabilities = ControllerCapabilities(control_space=ControlSpace.TORQUE)
    device_ctrl_mode: str = "random"
    step_count: int = field(default=0)
    env0: int = 0

    def prepare(self, model: Any, data: Any) -> None:
        if model.nu == 0:
            raise CompatibilityError("RandomCtrlController requires nu>0 to write controls.")
        self.step_count = 0

    def __call__(self, model: Any, data: Any, t: float) -> None:
        rows = _ctrl_rows(data)
        env = np.arange(rows.shape[0]) + self.env0
        u = philox_uniform(self.seed, env, self.step_count, model.nu)
        lo = np.where(model.actuator_ctrllimited, model.actuator_ctrlrange[:, 0], -1.0)
        hi = np.where(model.actuator_ctrllimited, model.actuator_ctrlrange[:, 1], 1.0)
        rows[...] = 0.5 * (lo + hi) + 0.5 * (hi - lo) * self.scale * (2.0 * u - 1.0)
        self.step_count += 1


@dataclass
class LinearFeedbackController:
    """``ctrl = clip(ctrl0 - K [q (-) q_goal ; qvel - qvel_goal] + std o P[step])`` — the LQR feedback law of the reference's
    examples (``examples/humanoid/controllers/lqr.py:147-170``, ``examples/drone2/main.py:400-471``), including its optional
    pre-drawn ctrl noise (``lqr.py:160-165``).  Three evaluations of the same law:

    * inside the fused rollout kernel (``device_ctrl_mode = "feedback"``, ``Env.rollout`` / ``run_passive_headless``);
    * ``__call__`` in a host-driven loop: ONE batched kernel for all environments (``mjb_feedback_ctrl``: ``K dx`` as an MFMA GEMM
      ``[batch, 2nv] x [2nv, nu]`` in fp32) writes ``ctrl`` on the device — no per-environment Python;
    * :meth:`host_law`: numpy on the host mirrors (what the tests compare the two device forms with).
    """

    K: np.ndarray = None            # [nu, 2 nv]
    ctrl0: np.ndarray = None        # [nu]
    qpos_goal: np.ndarray = None    # [nq]
    qvel_goal: np.ndarray | None = None
    ctrl_noise_std: np.ndarray | None = None     # [nu]
    perturbations: np.ndarray | None = None      # [nsteps, nu]: the reference's pre-drawn table, indexed by the step (mod nsteps)
    env_stride: int = 0                          # phase offset of the table per environment (0: every environment sees the same noise)
    # optional clipped integrators (the yaw-integral term of the drone example, examples/drone2/main.py:437-452): per step and environment
    #   z_j <- clip(z_j + (integ_rows[j] . dx) * timestep, +-integ_limit[j]);   ctrl -= sum_j z_j * integ_gain[j]
    # A law with integrators carries state between steps: it is evaluated for the whole batch with numpy on the host mirrors (one
    # vectorised call per step, no per-environment Python) and is NOT fused into rollout launches.
    integ_rows: np.ndarray | None = None         # [ni, 2 nv]
    integ_gain: np.ndarray | None = None         # [ni, nu]
    integ_limit: np.ndarray | None = None        # [ni]
    capabilities: ControllerCapabilities = ControllerCapabilities(control_space=ControlSpace.TORQUE)
    device_ctrl_mode: str | None = "feedback"
    step_count: int = field(default=0)

    def prepare(self, model: Any, data: Any) -> None:
        if model.nu == 0:
            raise CompatibilityError("LinearFeedbackController requires nu>0.")
        self.K = np.asarray(self.K, dtype=float)
        self.ctrl0 = np.asarray(self.ctrl0, dtype=float)
        self.qpos_goal = np.asarray(self.qpos_goal, dtype=float)
        self.qvel_goal = np.zeros(model.nv) if self.qvel_goal is None else np.asarray(self.qvel_goal, dtype=float)
        if self.K.shape != (model.nu, 2 * model.nv) or self.ctrl0.shape != (model.nu,) or self.qpos_goal.shape != (model.nq,):
            raise ConfigError("LinearFeedbackController: K must be [nu, 2nv], ctrl0 [nu], qpos_goal [nq]")
        if (self.ctrl_noise_std is None) != (self.perturbations is None):
            raise ConfigError("LinearFeedbackController: ctrl_noise_std and perturbations go together")
        self._integ = None
        if self.integ_rows is not None:
            self.integ_rows = np.atleast_2d(np.asarray(self.integ_rows, dtype=float))
            self.integ_gain = np.atleast_2d(np.asarray(self.integ_gain, dtype=float))
            self.integ_limit = np.atleast_1d(np.asarray(self.integ_limit, dtype=float))
            ni = self.integ_rows.shape[0]
            if self.integ_rows.shape != (ni, 2 * model.nv) or self.integ_gain.shape != (ni, model.nu) or self.integ_limit.shape != (ni,):
                raise ConfigError("LinearFeedbackController: integ_rows [ni, 2nv], integ_gain [ni, nu], integ_limit [ni]")
            self._integ = np.zeros((int(getattr(data, "batch", 1)), ni))
            self._dt = float(model.opt.timestep)
            self.device_ctrl_mode = None                 # stateful law: not fused, evaluated batched on the host mirrors
        elif self.device_ctrl_mode is None:
            self.device_ctrl_mode = "feedback"
        self.step_count = 0
        self._uploaded = None

    def upload(self, sim: Any) -> None:
        """Gains (and noise) to the device object, once per (controller, sim) pair."""
        if getattr(self, "_uploaded", None) is not sim:
            sim.set_feedback(self.K, self.ctrl0, self.qpos_goal, self.qvel_goal)
            sim.set_feedback_noise(self.ctrl_noise_std, self.perturbations, self.env_stride)
            self._uploaded = sim

    def __call__(self, model: Any, data: Any, t: float) -> None:
        if getattr(self, "_integ", None) is not None:    # stateful: the whole batch in numpy, integrators advanced once per call
            rows = _ctrl_rows(data)
            rows[...] = self.host_law(model, data, advance_integrators=True).reshape(rows.shape)
            self.step_count += 1
            return
        sim = data.sim
        data.push_host_edits()
        self.upload(sim)
        sim.feedback_ctrl(self.step_count)
        self.step_count += 1
        data.mark_device_newer()                 # ctrl now lives on the device: the mirrors refresh on the next read / step

    @staticmethod
    def fold_yaw_shaping(K: np.ndarray, yaw_direction: np.ndarray, yaw_error_index: int, yaw_rate_index: int, *, proportional_gain: float = 0.0,
                         derivative_gain: float = 0.0, control_scale: float = 1.0) -> np.ndarray:
        """The drone example's yaw shaping (examples/drone2/main.py:430-466) folded into the gain matrix: the P and D terms add
        ``yaw_direction (kp e_yaw + kd e_yawrate)^T`` to ``K``, and the control scale multiplies the component of ``K dx`` along the yaw
        direction by ``control_scale`` - all linear in ``dx``, so ``ctrl0 - K' dx`` IS the shaped law (its integral term: ``integ_*``)."""
        K = np.array(K, dtype=float)
        y = np.asarray(yaw_direction, dtype=float).reshape(-1)
        e = np.zeros(K.shape[1])
        e[yaw_error_index] += proportional_gain
        e[yaw_rate_index] += derivative_gain
        Kp = K + np.outer(y, e)
        return Kp + (control_scale - 1.0) * np.outer(y, y @ Kp) / float(y @ y)

    def host_law(self, model: Any, data: Any, step: int | None = None, advance_integrators: bool = False) -> np.ndarray:
        """The same law with numpy on the host mirrors; returns ctrl [batch, nu] (does not write it).  With integrators their state is
        advanced only when ``advance_integrators`` is set (``__call__`` does); otherwise the update is evaluated on a copy."""
        from . import mj

        qpos, qvel = np.atleast_2d(np.asarray(data.qpos, dtype=float)), np.atleast_2d(np.asarray(data.qvel, dtype=float))
        B = qpos.shape[0]
        lo = np.where(model.actuator_ctrllimited, model.actuator_ctrlrange[:, 0], -np.inf)
        hi = np.where(model.actuator_ctrllimited, model.actuator_ctrlrange[:, 1], np.inf)
        dq = np.zeros((B, model.nv))
        mj.mj_differentiatePos(model, dq, 1.0, np.ascontiguousarray(np.tile(self.qpos_goal, (B, 1))), np.ascontiguousarray(qpos))
        dx = np.concatenate([dq, qvel - self.qvel_goal], axis=1)
        u = self.ctrl0 - dx @ self.K.T
        if getattr(self, "_integ", None) is not None:
            z = np.clip(self._integ + (dx @ self.integ_rows.T) * self._dt, -self.integ_limit, self.integ_limit)
            if advance_integrators:
                self._integ = z
            u = u - z @ self.integ_gain
        if self.perturbations is not None:
            step = self.step_count if step is None else step
            idx = (step + np.arange(B) * self.env_stride) % len(self.perturbations)
            u = u + np.asarray(self.ctrl_noise_std) * np.asarray(self.perturbations)[idx]
        return np.clip(u, lo, hi)


__all__ = ["ZeroController", "PositionTargetDemo", "RandomCtrlController", "LinearFeedbackController", "philox_uniform"]
