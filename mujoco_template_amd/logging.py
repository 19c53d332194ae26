"""State / control CSV recorder fed from the batched engine (reference ``mujoco_template/logging.py:18-247``).

Same on-disk schema as the reference: ``time_s``, then per joint its ``qpos[...]`` and ``qvel[...]`` columns
(free joints expand to ``.pos_x .. .quat_z`` / ``.lin_x .. .ang_z``, hinge / slide joints are one bare column),
then ``ctrl[<actuator>]`` (``ctrl[none]`` for actuator-less models), then one column per ``DataProbe``.

Row order: the reference writes ``time``, then ALL qpos values (in the header's joint order), then ALL qvel values,
then ctrl (``logging.py:207-224``) although its header interleaves ``qpos[j]`` / ``qvel[j]`` per joint - so with more
than one joint the values sit under shifted labels.  Files are reproduced cell for cell by default;
``align_columns=True`` writes every value under its own header instead.

Two feeds:

* ``recorder(result)`` - the reference's ``StepHook`` protocol, one row per ``Env.step`` (host mirrors of
  environment ``env_index``; ``env_index=None`` logs every environment with a leading ``env`` column);
* ``recorder.record_rollout(nsteps)`` - the batched fast path: fused K-step launches write
  ``ctrl | qpos | qvel | time`` of every step into a device ring (``mjb_rollout``'s observation ring), one
  device-to-host copy per chunk, rows formatted in bulk.  Probes need a ``StepResult`` and are therefore
  only available on the hook path.
"""

from __future__ import annotations

from collections.abc import Callable, Iterator, Sequence
from dataclasses import dataclass
from pathlib import Path
from typing import Any

import numpy as np

from . import mj
from .env import Env, StepResult
from .exceptions import ConfigError
from .runtime import TrajectoryLogger

_QPOS_PARTS = {mj.mjtJoint.mjJNT_FREE: ("pos_x", "pos_y", "pos_z", "quat_w", "quat_x", "quat_y", "quat_z"),
               mj.mjtJoint.mjJNT_BALL: ("quat_w", "quat_x", "quat_y", "quat_z"),
               mj.mjtJoint.mjJNT_SLIDE: ("",), mj.mjtJoint.mjJNT_HINGE: ("",)}
_QVEL_PARTS = {mj.mjtJoint.mjJNT_FREE: ("lin_x", "lin_y", "lin_z", "ang_x", "ang_y", "ang_z"),
               mj.mjtJoint.mjJNT_BALL: ("ang_x", "ang_y", "ang_z"),
               mj.mjtJoint.mjJNT_SLIDE: ("",), mj.mjtJoint.mjJNT_HINGE: ("",)}


@dataclass(frozen=True)
class DataProbe:
    """An extra logged column: ``extractor(env, result)`` must return one scalar-compatible value (or None)."""

    name: str
    extractor: Callable[[Env, StepResult], Any]


def _span(adr: np.ndarray, j: int, total: int) -> range:
    return range(int(adr[j]), int(adr[j + 1]) if j + 1 < len(adr) else int(total))


def _labels(kind: str, joint: str, parts: tuple[str, ...] | None, n: int) -> list[str]:
    if parts is None or len(parts) != n:
        parts = tuple(f"c{i}" if n > 1 else "" for i in range(n))
    return [f"{kind}[{joint}]" + (f".{p}" if p else "") for p in parts]


def build_schema(model: "mj.MjModel", probe_names: Sequence[str] = ()) -> tuple[tuple[str, ...], list[int], list[int]]:
    """Column names and the qpos / qvel index of every state column, in the reference's joint-major order."""
    cols, qi, vi = ["time_s"], [], []
    for j in range(model.njnt):
        name = mj.mj_id2name(model, mj.mjtObj.mjOBJ_JOINT, j) or f"joint_{j}"
        jt = int(model.jnt_type[j])
        qs, vs = _span(model.jnt_qposadr, j, model.nq), _span(model.jnt_dofadr, j, model.nv)
        cols += _labels("qpos", name, _QPOS_PARTS.get(jt), len(qs)); qi += list(qs)
        cols += _labels("qvel", name, _QVEL_PARTS.get(jt), len(vs)); vi += list(vs)
    if model.nu > 0:
        cols += [f"ctrl[{mj.mj_id2name(model, mj.mjtObj.mjOBJ_ACTUATOR, a) or f'actuator_{a}'}]" for a in range(model.nu)]
    else:
        cols.append("ctrl[none]")
    cols += list(probe_names)
    return tuple(cols), qi, vi


class StateControlRecorder:
    """Clock, generalized coordinates, velocities and controls per simulation step, as CSV and / or in memory."""

    def __init__(self, env: Env, *, log_path: str | Path | None = None, store_rows: bool = True,
                 probes: Sequence[DataProbe] = (), env_index: int | None = 0, align_columns: bool = False) -> None:
        self._env, self._model = env, env.model
        self._store_rows, self._align = bool(store_rows), bool(align_columns)
        self._rows: list[tuple[object, ...]] = []
        self._probes = self._normalize_probes(probes)
        batch = int(getattr(env.data, "batch", 1))
        if env_index is not None and not 0 <= int(env_index) < batch:
            raise ConfigError(f"env_index {env_index} out of range for a batch of {batch}")
        self._env_index, self._batch = (None if env_index is None else int(env_index)), batch
        base, self._qpos_indices, self._qvel_indices = build_schema(self._model, [p.name for p in self._probes])
        if len(self._qpos_indices) != self._model.nq or len(self._qvel_indices) != self._model.nv:
            raise ConfigError("Internal recorder error: state index coverage mismatch.")
        self._columns = (("env",) if env_index is None else ()) + base
        self._column_index_map = {name: i for i, name in enumerate(self._columns)}
        self._logger = TrajectoryLogger(log_path, self._columns, self._format_row)
        self._dev_spec = None

    @staticmethod
    def _normalize_probes(probes: Sequence[DataProbe]) -> tuple[DataProbe, ...]:
        seen: set[str] = set()
        for p in probes:
            if not isinstance(p, DataProbe):
                raise ConfigError("All probes must be instances of DataProbe.")
            if not p.name:
                raise ConfigError("Probe names must be non-empty strings.")
            if p.name in seen:
                raise ConfigError(f"Duplicate probe name detected: {p.name}")
            if not callable(p.extractor):
                raise ConfigError(f"Probe '{p.name}' extractor must be callable.")
            seen.add(p.name)
        return tuple(probes)

    # -- introspection ---------------------------------------------------------------------------
    @property
    def columns(self) -> tuple[str, ...]:
        return self._columns

    @property
    def column_index(self) -> dict[str, int]:
        return dict(self._column_index_map)

    @property
    def rows(self) -> list[tuple[object, ...]]:
        return self._rows

    def as_dicts(self) -> Iterator[dict[str, object]]:
        for row in self._rows:
            yield {name: row[i] for name, i in self._column_index_map.items()}

    def __enter__(self) -> "StateControlRecorder":
        self._logger.__enter__()
        return self

    def __exit__(self, exc_type, exc, exc_tb) -> None:
        self._logger.__exit__(exc_type, exc, exc_tb)

    def close(self) -> None:
        self._logger.close()

    # -- row formatting --------------------------------------------------------------------------
    def _state_cells(self, t: float, qpos: np.ndarray, qvel: np.ndarray, ctrl: np.ndarray) -> list[object]:
        row: list[object] = [float(t)]
        if self._align:                                             # value under its own header (joint-major, interleaved)
            for j in range(self._model.njnt):
                row += [float(qpos[i]) for i in _span(self._model.jnt_qposadr, j, self._model.nq)]
                row += [float(qvel[i]) for i in _span(self._model.jnt_dofadr, j, self._model.nv)]
        else:                                                       # the reference's order: all qpos, then all qvel
            row += [float(qpos[i]) for i in self._qpos_indices]
            row += [float(qvel[i]) for i in self._qvel_indices]
        row += [float(c) for c in ctrl[: self._model.nu]] if self._model.nu > 0 else [""]
        return row

    def _probe_cells(self, result: StepResult) -> list[object]:
        cells: list[object] = []
        for p in self._probes:
            v = p.extractor(self._env, result)
            if isinstance(v, np.ndarray):
                if v.size != 1:
                    raise ConfigError(f"Probe '{p.name}' returned array with {v.size} elements; expected scalar.")
                v = float(v.item())
            elif isinstance(v, (list, tuple)):
                raise ConfigError(f"Probe '{p.name}' returned a non-scalar sequence; expected scalar-compatible value.")
            elif v is None:
                v = ""
            cells.append(v)
        return cells

    def _host_state(self, e: int):
        d = self._env.data
        if self._batch == 1:
            return float(d.time), np.asarray(d.qpos), np.asarray(d.qvel), np.asarray(d.ctrl)
        return float(np.asarray(d.time)[e]), np.asarray(d.qpos)[e], np.asarray(d.qvel)[e], np.asarray(d.ctrl)[e]

    def _format_row(self, result: StepResult) -> tuple[object, ...]:
        e = 0 if self._env_index is None else self._env_index
        return tuple(self._state_cells(*self._host_state(e)) + self._probe_cells(result))

    def __call__(self, result: StepResult) -> None:
        """``StepHook``: append the row(s) of the step that just finished."""
        if self._env_index is not None:
            row = self._logger.log(result)
            if self._store_rows:
                self._rows.append(row)
            return
        probes = self._probe_cells(result)
        for e in range(self._batch):
            row = self._logger.write_row([e] + self._state_cells(*self._host_state(e)) + probes)
            if self._store_rows:
                self._rows.append(row)

    # -- batched feed ----------------------------------------------------------------------------
    def record_rollout(self, nsteps: int, *, chunk: int = 256) -> int:
        """Advance ``nsteps`` in fused launches and log every step (device-side controller required, no probes)."""
        if self._probes:
            raise ConfigError("record_rollout() cannot evaluate probes (no per-step StepResult); use the hook path.")
        if nsteps < 1:
            raise ConfigError("record_rollout(nsteps): nsteps must be >= 1")
        env, m = self._env, self._model
        sim = env.data.sim
        if self._dev_spec is None or self._dev_spec.sim is not sim:
            self._dev_spec = sim.make_obs_spec(1 | 2 | 4 | 16)      # flat layout: ctrl | qpos | qvel | time
        nu, nq, nv = m.nu, m.nq, m.nv
        envs = range(self._batch) if self._env_index is None else (self._env_index,)
        done = 0
        while done < nsteps:
            n = min(int(chunk), nsteps - done)
            ring = env.rollout(n, obs_every=1, obs_spec_handle=self._dev_spec).cpu().numpy().astype(np.float64)   # [n, B, dim]
            for s in range(n):
                for e in envs:
                    r = ring[s, e]
                    cells = self._state_cells(r[nu + nq + nv], r[nu:nu + nq], r[nu + nq:nu + nq + nv], r[:nu])
                    row = self._logger.write_row(([e] if self._env_index is None else []) + cells)
                    if self._store_rows:
                        self._rows.append(row)
            done += n
        return done


__all__ = ["DataProbe", "StateControlRecorder", "build_schema"]
