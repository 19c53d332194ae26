"""How ``ctrl`` enters a step.

The contract is the reference's (``mujoco_template/control.py:9-32``): a controller is any object with a
``capabilities`` attribute, an optional-in-practice ``prepare(model, data)`` and a ``__call__(model, data, t)`` that
writes ``data.ctrl`` and nothing else.  ``ControllerCapabilities`` tells ``Env`` what to compute before the call
(the discrete linearisation, named Jacobians) and which actuator groups to enable.

Batched addition: a controller whose law can be evaluated inside the fused step kernel announces it through a
``device_ctrl_mode`` attribute (one of :data:`DEVICE_CTRL_MODES`); ``Env.rollout`` / ``run_passive_headless`` then
keep the whole loop on the GPU.  :func:`device_ctrl_mode_of` is the single place that decides whether a controller
qualifies.
"""

from __future__ import annotations

from collections.abc import Iterable
from dataclasses import dataclass, field
from typing import Any, Protocol, runtime_checkable

# laws the step kernel can evaluate itself (include/mjbatch.h: MJB_CTRL_ZERO / RANDOM / FEEDBACK)
DEVICE_CTRL_MODES = ("zero", "random", "feedback")


class ControlSpace:
    """String constants naming what an actuator's ``ctrl`` means; checked against the model by ``compat``."""

    TORQUE, POSITION, VELOCITY, INTVELOCITY = "torque", "position", "velocity", "intvelocity"


@dataclass(frozen=True)
class ControllerCapabilities:
    """What a controller expects from the environment before it is called."""

    control_space: str = ControlSpace.TORQUE
    needs_linearization: bool = False                      # Env puts (A, B) into info before the call
    needs_jacobians: Iterable[str] = field(default_factory=tuple)   # request tokens, see jacobians.py
    actuator_groups: Iterable[int] | None = None           # None = leave the model's groups alone


@runtime_checkable
class Controller(Protocol):
    capabilities: ControllerCapabilities

    def prepare(self, model: Any, data: Any) -> None: ...

    def __call__(self, model: Any, data: Any, t: float) -> None: ...


def device_ctrl_mode_of(controller: Any) -> str | None:
    """``device_ctrl_mode`` of a controller if the step kernel can run its law unaided, else ``None``.

    A controller that needs the linearisation or Jacobians recomputed every step is host-driven by definition.
    """
    if controller is None:
        return None
    mode = getattr(controller, "device_ctrl_mode", None)
    if mode not in DEVICE_CTRL_MODES:
        return None
    caps = controller.capabilities
    if caps.needs_linearization or tuple(caps.needs_jacobians):
        return None
    return mode


__all__ = ["ControlSpace", "ControllerCapabilities", "Controller", "DEVICE_CTRL_MODES", "device_ctrl_mode_of"]
