"""Controller protocol — the hook through which ``ctrl`` enters the step
(reference ``mujoco_template/control.py:9-32``).  A controller writes ``data.ctrl`` only.

Controllers that also expose ``device_ctrl_mode`` (``"zero"`` / ``"random"``) can be evaluated
inside the fused rollout kernel, so ``Env`` never leaves the GPU between steps.
"""

from __future__ import annotations

from collections.abc import Iterable
from dataclasses import dataclass, field
from typing import Any, Protocol


class ControlSpace:
    TORQUE = "torque"
    POSITION = "position"
    VELOCITY = "velocity"
    INTVELOCITY = "intvelocity"


@dataclass(frozen=True)
class ControllerCapabilities:
    control_space: str = ControlSpace.TORQUE
    needs_linearization: bool = False
    needs_jacobians: Iterable[str] = field(default_factory=tuple)
    actuator_groups: Iterable[int] | None = None


class Controller(Protocol):
    capabilities: ControllerCapabilities

    def prepare(self, model: Any, data: Any) -> None: ...

    def __call__(self, model: Any, data: Any, t: float) -> None: ...


__all__ = ["ControlSpace", "ControllerCapabilities", "Controller"]
