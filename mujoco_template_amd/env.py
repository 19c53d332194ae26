"""``Env`` — the step loop over the batched engine.

Signature-compatible with reference ``mujoco_template/env.py:28-260`` (``Env.__init__``,
``from_xml_path``, ``reset``, ``step``, ``linearize``, ``passive``, ``StepResult``); the
ordering contract is the reference's: the controller sees the pre-step state and
``t = data.time``; (A, B) and Jacobians are evaluated after ``ctrl`` is written and before the
step; ``compat_warnings`` appear in ``info`` once; info-key collisions raise ``TemplateError``.

New, keyword-only: ``batch`` / ``dtype`` / ``device`` / ``lanes`` / ``nconmax`` / ``nefcmax`` /
``env0`` on ``from_xml_path``, and :meth:`Env.rollout` — K fused steps in one kernel launch when
the controller runs on the device (``ZeroController``, ``RandomCtrlController``).
"""

from __future__ import annotations

import warnings
from collections.abc import Callable, Iterable, Iterator
from dataclasses import dataclass
from typing import Any

import numpy as np

from ._capi import CTRL_FEEDBACK, CTRL_KEEP, CTRL_RANDOM, CTRL_ZERO
from ._typing import InfoDict, JacobiansDict, Observation
from .compat import check_controller_compat
from .control import Controller
from .exceptions import ConfigError, TemplateError
from .jacobians import compute_requested_jacobians
from .linearization import linearize_discrete
from .model import ModelHandle
from .observations import ObservationExtractor, ObservationSpec


@dataclass
class StepResult:
    obs: Observation | None
    reward: float | None
    done: bool
    info: InfoDict


class Env:
    def __init__(self, handle: ModelHandle, obs_spec: ObservationSpec | None = None, controller: Controller | None = None,
                 reward_fn: Callable[[Any, Any, Observation | None], float] | None = None,
                 done_fn: Callable[[Any, Any, Observation | None], bool] | None = None,
                 info_fn: Callable[[Any, Any, Observation | None], dict] | None = None,
                 enabled_groups: Iterable[int] | None = None, control_decimation: int = 1):
        if control_decimation < 1:
            raise ConfigError("control_decimation must be >= 1")
        self.handle = handle
        self.model = handle.model
        self.data = handle.data
        self._obs_spec = obs_spec
        self.extractor: ObservationExtractor | None = None if obs_spec is None else ObservationExtractor(handle.model, obs_spec)
        self.controller = controller
        self.reward_fn, self.done_fn, self.info_fn = reward_fn, done_fn, info_fn
        self.control_decimation = int(control_decimation)
        self._substep = 0
        self._added_warnings = False
        self._compat_warnings: list[str] = []
        self._device_steps = 0

        caps = controller.capabilities if controller is not None else None
        requested = tuple(int(g) for g in caps.actuator_groups) if caps is not None and caps.actuator_groups is not None else None
        if enabled_groups is not None:
            chosen = tuple(int(g) for g in enabled_groups)
            if requested is not None and set(requested) != set(chosen):
                msg = (f"Controller declares actuator groups {sorted(set(requested))} but user requested "
                       f"{sorted(set(chosen))}; proceeding with the user selection.")
                warnings.warn(msg, RuntimeWarning)
                self._compat_warnings.append(msg)
            self.handle.set_enabled_actuator_groups(chosen)
        elif requested is not None:
            msg = (f"Controller declares actuator groups {sorted(set(requested))} but Env leaves actuator "
                   "availability unchanged by default.")
            warnings.warn(msg, RuntimeWarning)
            self._compat_warnings.append(msg)
        if controller is not None:
            controller.prepare(self.model, self.data)
            report = check_controller_compat(self.model, controller.capabilities, self.handle.enabled_actuator_mask())
            self._compat_warnings = list(report.warnings)
            report.assert_ok()

    @property
    def compat_warnings(self) -> list[str]:
        return list(self._compat_warnings)

    @classmethod
    def from_xml_path(cls, xml_path: str, *, obs_spec: ObservationSpec | None = None, controller: Controller | None = None,
                      reward_fn=None, done_fn=None, info_fn=None, enabled_groups: Iterable[int] | None = None,
                      control_decimation: int = 1, auto_reset: bool = True, keyframe: int | str | None = None,
                      batch: int = 1, dtype: str = "float32", device: int = 0, lanes: int = 0, nconmax: int = 0,
                      nefcmax: int = 0, env0: int = 0, specialize: bool | None = None) -> "Env":
        if obs_spec is None:
            obs_spec = ObservationSpec(include_sensordata=False)
        handle = ModelHandle.from_xml_path(xml_path, batch=batch, dtype=dtype, device=device, lanes=lanes, nconmax=nconmax,
                                           nefcmax=nefcmax, env0=env0, specialize=specialize)
        if controller is not None and hasattr(controller, "env0"):
            controller.env0 = env0
        env = cls(handle, obs_spec=obs_spec, controller=controller, reward_fn=reward_fn, done_fn=done_fn, info_fn=info_fn,
                  enabled_groups=enabled_groups, control_decimation=control_decimation)
        if keyframe is not None and not auto_reset:
            raise ConfigError("auto_reset=False is incompatible with specifying a keyframe")
        if auto_reset:
            env.reset(keyframe)
        return env

    def _ensure_extractor(self) -> ObservationExtractor:
        if self.extractor is None:
            if self._obs_spec is None:
                self._obs_spec = ObservationSpec()
            self.extractor = ObservationExtractor(self.model, self._obs_spec)
        return self.extractor

    def reset(self, keyframe: int | str | None = None) -> Observation:
        if keyframe is None:
            self.handle.reset()
        else:
            self.handle.reset_keyframe(keyframe)
        self.handle.forward()
        self._substep = 0
        self._added_warnings = False
        self._device_steps = 0
        if self.controller is not None:
            self.controller.prepare(self.model, self.data)
        return self._ensure_extractor()(self.data)

    # -- fused device path --------------------------------------------------------------------
    def _device_mode(self) -> int | None:
        if self.controller is None:
            return CTRL_KEEP
        mode = getattr(self.controller, "device_ctrl_mode", None)
        caps = self.controller.capabilities
        if mode is None or caps.needs_linearization or tuple(caps.needs_jacobians) or self.control_decimation != 1:
            return None
        return {"zero": CTRL_ZERO, "random": CTRL_RANDOM, "feedback": CTRL_FEEDBACK}.get(mode)

    def can_fuse(self) -> bool:
        """True when nothing on the host has to observe individual steps (controller on device, no reward/done/info hooks)."""
        extras = bool(self.extractor.extra_items) if self.extractor is not None else False
        return self._device_mode() is not None and not any((self.reward_fn, self.done_fn, self.info_fn)) and not extras

    def rollout(self, nsteps: int, *, obs_every: int = 0, obs_out=None, obs_spec_handle=None):
        """Advance ``nsteps`` in ONE kernel launch (controller evaluated on the device).

        With ``obs_every = k > 0`` the flat observation of every k-th step is written on the GPU
        and returned as a torch tensor ``[nsteps // k, batch, obs_dim]``.  ``obs_spec_handle`` (a device
        ``ObsSpecHandle``) replaces the environment's own observation layout for this call (the CSV recorder's feed).
        """
        mode = self._device_mode()
        if mode is None:
            raise ConfigError("Env.rollout needs a device-side controller (ZeroController / RandomCtrlController) or none")
        if nsteps < 1:
            raise ConfigError("Env.rollout(nsteps): nsteps must be >= 1")
        data, sim = self.data, self.data.sim
        data.push_host_edits()
        seed = int(getattr(self.controller, "seed", 0))
        scale = float(getattr(self.controller, "scale", 1.0))
        if mode == CTRL_FEEDBACK:
            ctl = self.controller
            sim.set_feedback(ctl.K, ctl.ctrl0, ctl.qpos_goal, ctl.qvel_goal)
        spec = ptr = None
        if obs_every > 0:
            import torch

            spec = obs_spec_handle if obs_spec_handle is not None else self._ensure_extractor().device_spec(data)
            if obs_out is None:
                obs_out = torch.empty((nsteps // obs_every, data.batch, spec.dim), device=f"cuda:{sim.device}",
                                      dtype=torch.float32 if sim.dtype == "float32" else torch.float64)
            ptr = obs_out.data_ptr()
        sim.rollout(nsteps, mode, seed=seed, step0=self._device_steps, ctrl_scale=scale, obs_spec=spec, obs_out_ptr=ptr or 0, obs_every=obs_every)
        self._device_steps += nsteps
        self._substep += nsteps
        if hasattr(self.controller, "step_count"):
            self.controller.step_count = self._device_steps
        data.mark_device_newer()
        return obs_out

    # -- reference step loop ----------------------------------------------------------------------
    def step(self, n: int = 1, *, return_obs: bool = True) -> StepResult:
        if n < 1:
            raise ConfigError("Env.step(n): n must be >= 1")
        info: InfoDict = {}
        if not self._added_warnings and self._compat_warnings:
            info["compat_warnings"] = list(self._compat_warnings)
            self._added_warnings = True

        if self._device_mode() is not None and self.controller is not None:
            self.rollout(n)                       # controller + n steps fused on the device
            self.data.sync_host()
        else:
            hist_A: list[np.ndarray] = []
            hist_B: list[np.ndarray] = []
            hist_J: list[JacobiansDict] = []
            for _ in range(n):
                if self.controller is not None and self._substep % self.control_decimation == 0:
                    self.controller(self.model, self.data, _scalar_time(self.data.time))
                    caps = self.controller.capabilities
                    if caps.needs_linearization:
                        A, B = linearize_discrete(self.model, self.data, use_native=True)
                        hist_A.append(A)
                        hist_B.append(B)
                    if caps.needs_jacobians:
                        hist_J.append(compute_requested_jacobians(self.model, self.data, caps.needs_jacobians))
                self.handle.step()
                self._substep += 1
                self._device_steps += 1
            if hist_A:
                info["A"] = hist_A[0] if len(hist_A) == 1 else hist_A
                info["B"] = hist_B[0] if len(hist_B) == 1 else hist_B
            if hist_J:
                info["jacobians"] = hist_J[0] if len(hist_J) == 1 else hist_J

        obs: Observation | None = self._ensure_extractor()(self.data) if return_obs else None
        reward: float | None = None
        done = False
        if self.reward_fn:
            reward = self.reward_fn(self.model, self.data, obs)
        if self.done_fn:
            done = bool(self.done_fn(self.model, self.data, obs))
        if self.info_fn:
            for key, value in self.info_fn(self.model, self.data, obs).items():
                if key in info:
                    raise TemplateError(f"info key collision: {key}")
                info[key] = value
        return StepResult(obs=obs, reward=reward, done=done, info=info)

    def linearize(self, eps: float = 1e-6, horizon_steps: int = 1) -> tuple[np.ndarray, np.ndarray]:
        return linearize_discrete(self.model, self.data, use_native=True, eps=eps, horizon_steps=horizon_steps)

    def passive(self, *, duration: float | None = None, max_steps: int | None = None,
                hooks: Callable[[StepResult], None] | Iterable[Callable[[StepResult], None]] | None = None,
                return_obs: bool = True) -> Iterator[StepResult]:
        from .runtime import iterate_passive

        yield from iterate_passive(self, duration=duration, max_steps=max_steps, hooks=hooks, return_obs=return_obs)


def _scalar_time(t: Any) -> float:
    return float(t) if np.ndim(t) == 0 else float(np.asarray(t).flat[0])


__all__ = ["Env", "StepResult"]
