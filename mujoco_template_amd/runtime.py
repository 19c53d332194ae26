"""Passive rollout loop (reference ``mujoco_template/runtime.py:620-685``):
``iterate_passive`` / ``run_passive_headless`` keep their signatures.  When the environment's
controller runs on the device and no per-step Python hook is attached, ``run_passive_headless``
executes the whole loop as fused K-step kernel launches (one host crossing per chunk for the
whole batch instead of one per step per environment).
"""

from __future__ import annotations

import csv
from collections.abc import Callable, Iterable, Iterator, Sequence
from pathlib import Path
from typing import IO, TYPE_CHECKING, Any

import numpy as np

from .exceptions import ConfigError

if TYPE_CHECKING:  # pragma: no cover
    from .env import Env, StepResult

StepHook = Callable[["StepResult"], None]


def _normalize_hooks(hooks: StepHook | Iterable[StepHook] | None) -> tuple[StepHook, ...]:
    if hooks is None:
        return ()
    if callable(hooks):
        return (hooks,)
    normalized = tuple(hooks)
    if not all(callable(h) for h in normalized):
        raise ConfigError("All hooks must be callables accepting StepResult.")
    return normalized


def _validate(duration: float | None, max_steps: int | None) -> None:
    if duration is not None and duration < 0:
        raise ConfigError("duration must be >= 0 when provided.")
    if max_steps is not None and max_steps < 1:
        raise ConfigError("max_steps must be >= 1 when provided.")


def _time_reached(env: "Env", duration: float) -> bool:
    return bool(np.all(np.asarray(env.data.time) >= duration))


def iterate_passive(env: "Env", *, duration: float | None = None, max_steps: int | None = None,
                    hooks: StepHook | Iterable[StepHook] | None = None, return_obs: bool = True) -> Iterator["StepResult"]:
    """Yield one ``StepResult`` per step; the stop condition is checked after the step (>= 1 step always)."""
    _validate(duration, max_steps)
    active_hooks = _normalize_hooks(hooks)
    steps = 0
    while True:
        result = env.step(return_obs=return_obs)
        steps += 1
        for hook in active_hooks:
            hook(result)
        yield result
        if max_steps is not None and steps >= max_steps:
            break
        if duration is not None and _time_reached(env, duration):
            break


def run_passive_headless(env: "Env", *, duration: float | None = None, max_steps: int | None = None,
                         hooks: StepHook | Iterable[StepHook] | None = None, return_obs: bool = True,
                         chunk: int = 256) -> int:
    """Drive the environment headlessly and return the executed step count."""
    _validate(duration, max_steps)
    active_hooks = _normalize_hooks(hooks)
    if not active_hooks and env.can_fuse():   # nothing observes the intermediate StepResults
        # fused path: chunks of steps per launch.  With a duration stop the clocks are re-read after every chunk: normally time
        # advances by exactly one timestep per step, but an in-kernel bad-state reset restarts that environment's clock
        # (mj_checkPos/Vel/Acc -> mj_resetData), and the per-step loop would keep stepping until EVERY clock has reached the duration.
        dt = float(env.model.opt.timestep)
        done = 0
        while max_steps is None or done < max_steps:
            n = chunk if max_steps is None else min(chunk, max_steps - done)
            if duration is not None:
                t_min = float(np.min(np.asarray(env.data.time)))
                if done > 0 and t_min >= duration:
                    break
                n = min(n, max(1, int(np.ceil((duration - t_min) / dt - 1e-9))))     # never step past the slowest clock's target
            elif max_steps is None:
                break                                  # no stop condition at all: like the generator, fall through to one plain step
            env.rollout(n)
            done += n
        if done > 0:
            return done
    steps = 0
    for _ in iterate_passive(env, duration=duration, max_steps=max_steps, hooks=active_hooks, return_obs=return_obs):
        steps += 1
    return steps


__all__ = ["StepHook", "iterate_passive", "run_passive_headless", "_normalize_hooks"]


class TrajectoryLogger:
    """Row sink with a fixed header, optionally backed by a CSV file (reference ``runtime.py:559-617``).

    ``path=None`` keeps the formatting / length check but writes nothing.  Usable as a context manager; the file
    is opened (parents created) on ``__enter__`` and the header row is written first.
    """

    def __init__(self, path: str | Path | None, columns: Sequence[str], formatter: Callable[["StepResult"], Sequence[Any]]) -> None:
        if not columns:
            raise ConfigError("TrajectoryLogger requires at least one column name.")
        if formatter is None:
            raise ConfigError("TrajectoryLogger requires a formatter callable.")
        self._path = None if path is None else Path(path)
        self._columns = tuple(columns)
        self._formatter = formatter
        self._file: IO[str] | None = None
        self._writer = None

    @property
    def columns(self) -> tuple[str, ...]:
        return self._columns

    @property
    def enabled(self) -> bool:
        return self._path is not None

    def __enter__(self) -> "TrajectoryLogger":
        if self._path is not None and self._file is None:
            self._path.parent.mkdir(parents=True, exist_ok=True)
            self._file = self._path.open("w", newline="", encoding="utf-8")
            self._writer = csv.writer(self._file)
            self._writer.writerow(self._columns)
        return self

    def __exit__(self, exc_type, exc, exc_tb) -> None:
        self.close()

    def close(self) -> None:
        if self._file is not None:
            self._file.close()
        self._file = self._writer = None

    def write_row(self, row: Sequence[Any]) -> tuple[Any, ...]:
        row = tuple(row)
        if len(row) != len(self._columns):
            raise ConfigError(f"Formatter returned a row of unexpected length ({len(row)} received, expected {len(self._columns)}).")
        if self._writer is not None:
            self._writer.writerow(row)
        return row

    def log(self, result: "StepResult") -> tuple[Any, ...]:
        return self.write_row(self._formatter(result))
