"""``Env`` - controller, engine and observation extractor behind one ``step()``.

API of the reference (``mujoco_template/env.py:20-260``: ``StepResult``, ``Env.__init__``, ``from_xml_path``, ``reset``,
``step``, ``linearize``, ``passive``) with its ordering guarantees: the controller is called on the PRE-step state with
``t = data.time``; the discrete linearisation and the requested Jacobians are evaluated after ``ctrl`` was written and
before the physics advances; compatibility warnings travel in ``info`` exactly once; a user ``info_fn`` may not overwrite
a key the environment produced (``TemplateError``).

What the batched engine adds:

* ``from_xml_path(..., batch, dtype, device, lanes, nconmax, nefcmax, env0, specialize)``;
* ``from_xml_path(..., batch=GLOBAL, shard=True)``: one process per GPU (torchrun environment) - this process takes its contiguous
  block of the global batch (``env.shard``: a ``distributed.ShardPlan``), keys its random streams by the GLOBAL environment index
  and steps it on ``cuda:LOCAL_RANK``; ``rollout(..., gather=True)`` / ``gather_observations`` return the observation block of the
  WHOLE batch on every rank (the path's one collective: RCCL all-gather, through the library's ``mjb_allgather_obs`` when this
  process can own an ``ncclComm_t``);
* two ways through a step.  *Host-driven*: any Python controller, one engine call per sub-step (``_advance_on_host``).
  *Fused*: a controller whose law the step kernel can evaluate itself (``control.device_ctrl_mode_of``) runs inside ONE
  kernel launch for all sub-steps (``rollout``); ``step`` picks it automatically;
* :meth:`Env.rollout` - the fused path exposed directly, optionally filling a device-side observation ring.
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
from .control import Controller, device_ctrl_mode_of
from .exceptions import ConfigError, TemplateError
from .jacobians import compute_requested_jacobians
from .linearization import linearize_discrete
from .model import ModelHandle
from .observations import ObservationExtractor, ObservationSpec

_KERNEL_MODE = {"zero": CTRL_ZERO, "random": CTRL_RANDOM, "feedback": CTRL_FEEDBACK}


@dataclass
class StepResult:
    obs: Observation | None
    reward: float | None
    done: bool
    info: InfoDict


def _one_or_all(history: list) -> Any:
    """A single sub-step reports its value bare, several sub-steps report the list (reference env.py:203-215)."""
    return history[0] if len(history) == 1 else history


class Env:
    def __init__(self, handle: ModelHandle, obs_spec: ObservationSpec | None = None, controller: Controller | None = None,
                 reward_fn: Callable[[Any, Any, Observation | None], float] | None = None,
                 done_fn: Callable[[Any, Any, Observation | None], bool] | None = None,
                 info_fn: Callable[[Any, Any, Observation | None], dict] | None = None,
                 enabled_groups: Iterable[int] | None = None, control_decimation: int = 1):
        if control_decimation < 1:
            raise ConfigError("control_decimation must be >= 1")
        self.handle, self.model, self.data = handle, handle.model, handle.data
        self.controller = controller
        self.reward_fn, self.done_fn, self.info_fn = reward_fn, done_fn, info_fn
        self.control_decimation = int(control_decimation)
        self._obs_spec = obs_spec
        self.extractor: ObservationExtractor | None = ObservationExtractor(handle.model, obs_spec) if obs_spec is not None else None
        self._substep = 0                 # physics steps since reset (decimation phase)
        self._device_steps = 0            # same count, as the step index of the device-side RNG stream
        self._warnings_reported = False
        self._engine_warnings_reported = 0
        self._compat_warnings: list[str] = []
        self.shard = None                 # distributed.ShardPlan when created with shard=True
        self._collective = "auto"
        self._comm = None                 # RcclCommunicator (lazily, first gather)
        self.gather_collective = None     # which collective the last gather used (reported by bench.py)
        self._select_actuator_groups(enabled_groups)
        if controller is not None:
            controller.prepare(self.model, self.data)
            report = check_controller_compat(self.model, controller.capabilities, self.handle.enabled_actuator_mask())
            self._compat_warnings = list(report.warnings)     # the report's list replaces the group notes (reference env.py:92)
            report.assert_ok()

    def _select_actuator_groups(self, enabled_groups: Iterable[int] | None) -> None:
        """The user's ``enabled_groups`` win over the controller's declaration; a mismatch or an ignored declaration warns."""
        caps = getattr(self.controller, "capabilities", None)
        declared = None if caps is None or caps.actuator_groups is None else sorted({int(g) for g in caps.actuator_groups})
        note = None
        if enabled_groups is not None:
            chosen = sorted({int(g) for g in enabled_groups})
            if declared is not None and declared != chosen:
                note = f"Controller declares actuator groups {declared} but user requested {chosen}; proceeding with the user selection."
            self.handle.set_enabled_actuator_groups(chosen)
        elif declared is not None:
            note = f"Controller declares actuator groups {declared} but Env leaves actuator availability unchanged by default."
        if note is not None:
            warnings.warn(note, RuntimeWarning)
            self._compat_warnings.append(note)

    @property
    def compat_warnings(self) -> list[str]:
        return list(self._compat_warnings)

    @classmethod
    def from_xml_path(cls, xml_path: str, *, obs_spec: ObservationSpec | None = None, controller: Controller | None = None,
                      reward_fn=None, done_fn=None, info_fn=None, enabled_groups: Iterable[int] | None = None,
                      control_decimation: int = 1, auto_reset: bool = True, keyframe: int | str | None = None,
                      batch: int = 1, dtype: str = "float32", device: int | None = None, lanes: int = 0, nconmax: int = 0,
                      nefcmax: int = 0, env0: int = 0, specialize: bool | None = None, shard: bool = False,
                      collective: str = "auto") -> "Env":
        """Reference signature (``mujoco_template/env.py:100-143``) + the engine's keywords.  ``shard=True``: ``batch`` is the GLOBAL batch
        of a one-process-per-GPU job (RANK / WORLD_SIZE / LOCAL_RANK from the torchrun environment); this process creates its
        contiguous block on ``cuda:LOCAL_RANK`` (``device`` overrides) with ``env0`` = the block's first global index.
        ``collective``: ``"auto"`` (the library's ``mjb_allgather_obs`` on an own RCCL communicator when the process group's backend is
        ``nccl`` and the shards are equal, else ``torch.distributed``), ``"rccl"`` (require the former), ``"torch"``."""
        if keyframe is not None and not auto_reset:
            raise ConfigError("auto_reset=False is incompatible with specifying a keyframe")
        if collective not in ("auto", "rccl", "torch"):
            raise ConfigError("collective must be 'auto', 'rccl' or 'torch'")
        plan = None
        if shard:
            from .distributed import ShardPlan

            try:
                plan = ShardPlan.from_environment(batch)
            except ValueError as exc:
                raise ConfigError(str(exc)) from None
            batch, env0 = plan.count, int(env0) + plan.env0
            if device is None:
                device = plan.local_rank if plan.world_size > 1 else 0
        handle = ModelHandle.from_xml_path(xml_path, batch=batch, dtype=dtype, device=0 if device is None else device, lanes=lanes, nconmax=nconmax,
                                           nefcmax=nefcmax, env0=env0, specialize=specialize)
        if controller is not None and hasattr(controller, "env0"):
            controller.env0 = env0                  # device-side RNG streams are keyed by the GLOBAL environment index
        env = cls(handle, obs_spec=obs_spec if obs_spec is not None else ObservationSpec(include_sensordata=False),
                  controller=controller, reward_fn=reward_fn, done_fn=done_fn, info_fn=info_fn,
                  enabled_groups=enabled_groups, control_decimation=control_decimation)
        env.shard, env._collective = plan, collective
        if auto_reset:
            env.reset(keyframe)
        return env

    def _ensure_extractor(self) -> ObservationExtractor:
        if self.extractor is None:
            self._obs_spec = self._obs_spec or ObservationSpec()
            self.extractor = ObservationExtractor(self.model, self._obs_spec)
        return self.extractor

    def reset(self, keyframe: int | str | None = None) -> Observation:
        if keyframe is None:
            self.handle.reset()
        else:
            self.handle.reset_keyframe(keyframe)
        self.handle.forward()
        self._substep = self._device_steps = 0
        self._warnings_reported = False
        if self.controller is not None:
            self.controller.prepare(self.model, self.data)     # several example controllers write qpos/qvel/ctrl here
        return self._ensure_extractor()(self.data)

    # -- fused path ---------------------------------------------------------------------------------
    def _device_mode(self) -> int | None:
        """Kernel ctrl mode when the whole step can stay on the GPU; ``None`` when the host has to drive it."""
        if self.controller is None:
            return CTRL_KEEP
        if self.control_decimation != 1:
            return None
        name = device_ctrl_mode_of(self.controller)
        return None if name is None else _KERNEL_MODE[name]

    def can_fuse(self) -> bool:
        """Nothing on the host needs to see individual steps: device controller, no reward / done / info hooks, no extras."""
        if self._device_mode() is None or self.reward_fn or self.done_fn or self.info_fn:
            return False
        return not (self.extractor is not None and self.extractor.extra_items)

    # -- multi-GPU: the one collective of the path --------------------------------------------------
    def _communicator(self):
        """The process' own RCCL communicator for ``mjb_allgather_obs`` (created at the first gather), or ``None`` = use torch.distributed."""
        if self._collective == "torch" or self.shard is None:
            return None
        if self._comm is not None:
            return self._comm or None
        import torch.distributed as dist

        usable = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" and self.shard.equal_shards
        if not usable:
            if self._collective == "rccl":
                raise ConfigError("collective='rccl' needs an initialised torch.distributed group with backend 'nccl' and equal shards")
            self._comm = False
            return None
        from .distributed import RcclCommunicator

        try:
            self._comm = RcclCommunicator(self.shard.rank, self.shard.world_size, self.data.sim.device)
        except Exception as exc:                                   # no loadable RCCL / init refused: the torch collective still works
            if self._collective == "rccl":
                raise TemplateError(f"could not create an RCCL communicator: {exc}") from exc
            warnings.warn(f"mjb_allgather_obs unavailable ({exc}); the observation all-gather uses torch.distributed", RuntimeWarning)
            self._comm = False
            return None
        return self._comm

    def gather_observations(self, local_obs):
        """All-gather a device observation block ``[..., local batch, dim]`` of a sharded environment into ``[..., GLOBAL batch, dim]`` on
        every rank (environment order).  Without ``shard=True`` (or in a one-process job without a process group) it is the identity."""
        if self.shard is None:
            return local_obs
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            if self.shard.world_size > 1:
                raise ConfigError("a sharded Env needs torch.distributed initialised before gathering (distributed.init_process_group())")
            self.gather_collective = "identity (one process, no process group)"
            return local_obs
        comm = self._communicator()
        if comm is not None:
            self.gather_collective = "mjb_allgather_obs (ncclAllGather on the library's own RCCL communicator)"
            return self.shard.gather(local_obs, comm=comm, stream=None)
        self.gather_collective = f"torch.distributed ({dist.get_backend()})"
        from .distributed import all_gather_obs

        return all_gather_obs(local_obs, counts=list(self.shard.counts), single_rank=True)

    def observe_device(self, gather: bool = False):
        """Flat observation ``[batch, obs_dim]`` of the current state as a torch tensor on the GPU (``ObservationExtractor.gather_device``);
        ``gather=True`` on a sharded environment: the all-gathered ``[GLOBAL batch, obs_dim]``."""
        obs = self._ensure_extractor().gather_device(self.data)
        return self.gather_observations(obs) if gather else obs

    def rollout(self, nsteps: int, *, obs_every: int = 0, obs_out=None, obs_spec_handle=None, gather: bool = False):
        """Advance ``nsteps`` in ONE kernel launch (controller evaluated on the device).

        With ``obs_every = k > 0`` the flat observation of every k-th step is written on the GPU
        and returned as a torch tensor ``[nsteps // k, batch, obs_dim]``.  ``obs_spec_handle`` (a device
        ``ObsSpecHandle``) replaces the environment's own observation layout for this call (the CSV recorder's feed).
        ``gather=True`` (sharded environments): the returned block is the all-gathered ``[nsteps // k, GLOBAL batch, obs_dim]``.
        """
        mode = self._device_mode()
        if mode is None:
            raise ConfigError("Env.rollout needs a device-side controller (ZeroController / RandomCtrlController) or none")
        if nsteps < 1:
            raise ConfigError("Env.rollout(nsteps): nsteps must be >= 1")
        data, sim, ctl = self.data, self.data.sim, self.controller
        data.push_host_edits()
        if mode == CTRL_FEEDBACK:
            ctl.upload(sim)                                      # gains + optional ctrl noise, once per (controller, sim)
        spec, ring_ptr = None, 0
        if obs_every > 0:
            import torch

            spec = obs_spec_handle if obs_spec_handle is not None else self._ensure_extractor().device_spec(data)
            if obs_out is None:
                obs_out = torch.empty((nsteps // obs_every, data.batch, spec.dim), device=f"cuda:{sim.device}",
                                      dtype=torch.float32 if sim.dtype == "float32" else torch.float64)
            ring_ptr = obs_out.data_ptr()
        sim.rollout(nsteps, mode, seed=int(getattr(ctl, "seed", 0)), step0=self._device_steps,
                    ctrl_scale=float(getattr(ctl, "scale", 1.0)), obs_spec=spec, obs_out_ptr=ring_ptr, obs_every=obs_every)
        self._device_steps += nsteps
        self._substep += nsteps
        if hasattr(ctl, "step_count"):
            ctl.step_count = self._device_steps
        data.mark_device_newer()
        if gather and obs_out is not None:
            return self.gather_observations(obs_out)
        return obs_out

    # -- host-driven path --------------------------------------------------------------------------
    def _advance_on_host(self, n: int, info: InfoDict) -> None:
        """``n`` physics steps with the Python controller in the loop; per-sub-step (A, B) / Jacobians go into ``info``."""
        ctl = self.controller
        caps = ctl.capabilities if ctl is not None else None
        lin_A: list[np.ndarray] = []
        lin_B: list[np.ndarray] = []
        jacs: list[JacobiansDict] = []
        for _ in range(n):
            if ctl is not None and self._substep % self.control_decimation == 0:
                ctl(self.model, self.data, _scalar_time(self.data.time))
                if caps.needs_linearization:
                    A, B = linearize_discrete(self.model, self.data, use_native=True)
                    lin_A.append(A); lin_B.append(B)
                if caps.needs_jacobians:
                    jacs.append(compute_requested_jacobians(self.model, self.data, caps.needs_jacobians))
            self.handle.step()
            self._substep += 1
            self._device_steps += 1
        if lin_A:
            info["A"], info["B"] = _one_or_all(lin_A), _one_or_all(lin_B)
        if jacs:
            info["jacobians"] = _one_or_all(jacs)

    def step(self, n: int = 1, *, return_obs: bool = True) -> StepResult:
        if n < 1:
            raise ConfigError("Env.step(n): n must be >= 1")
        info: InfoDict = {}
        if self._compat_warnings and not self._warnings_reported:
            info["compat_warnings"] = list(self._compat_warnings)
            self._warnings_reported = True
        if self.controller is not None and self._device_mode() is not None:
            self.rollout(n)                        # controller + n steps in one launch
            self.data.sync_host()
        else:
            self._advance_on_host(n, info)
        ew = getattr(self.data, "engine_warnings", None)
        if ew and len(ew) > self._engine_warnings_reported:      # truncated physics (LDS caps) / bad-state resets, reported once each
            info["engine_warnings"] = list(ew[self._engine_warnings_reported:])
            self._engine_warnings_reported = len(ew)
        obs: Observation | None = self._ensure_extractor()(self.data) if return_obs else None
        # the user hooks run even without an observation (they then receive obs=None), like the reference
        reward = self.reward_fn(self.model, self.data, obs) if self.reward_fn else None
        done = bool(self.done_fn(self.model, self.data, obs)) if self.done_fn else False
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
    """The controller's ``t``: the clock of environment 0 (all environments share the timestep)."""
    return float(t) if np.ndim(t) == 0 else float(np.asarray(t).flat[0])


__all__ = ["Env", "StepResult"]
