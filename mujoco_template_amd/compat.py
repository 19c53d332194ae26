"""Setup-time controller/model diagnostics (reference ``mujoco_template/compat.py:31-127``).
Not on the per-step path; kept so ``Env`` surfaces the same warnings/failures."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np

from .control import ControlSpace, ControllerCapabilities
from .exceptions import CompatibilityError, ConfigError


@dataclass
class CompatibilityReport:
    ok: bool
    reasons: list[str] = field(default_factory=list)
    warnings: list[str] = field(default_factory=list)

    def assert_ok(self) -> None:
        if not self.ok:
            raise CompatibilityError("\n".join(["Incompatible controller/model:"] + [f"- {r}" for r in self.reasons]))


def _bad_range(lo: float, hi: float) -> bool:
    return not (np.isfinite(lo) and np.isfinite(hi) and hi > lo)


def check_controller_compat(model: Any, ctrl_cap: ControllerCapabilities, enabled_mask: np.ndarray | None) -> CompatibilityReport:
    reasons: list[str] = []
    notes: list[str] = []
    nu = int(model.nu)
    if nu == 0:
        reasons.append("Model has no actuators (nu=0).")
    mask = np.ones(nu, dtype=bool) if enabled_mask is None else np.asarray(enabled_mask, dtype=bool)
    if mask.shape[0] != nu:
        raise ConfigError("enabled_mask must have length model.nu")
    if not mask.any():
        reasons.append("All actuators are disabled by group selection.")
    enabled = np.flatnonzero(mask)

    if ctrl_cap.actuator_groups is not None:
        wanted = {int(g) for g in ctrl_cap.actuator_groups}
        have = {int(g) for g in np.asarray(model.actuator_group)[enabled]}
        if not have:
            notes.append("Controller declared actuator groups but none are currently enabled; continuing without additional group gating.")
        if sorted(wanted - have):
            notes.append(f"Controller requested actuator groups {sorted(wanted - have)} but they are not enabled; controller will still run with the available groups.")
        if sorted(have - wanted):
            notes.append(f"Enabled actuators include groups {sorted(have - wanted)} beyond the controller request; behaviour matches MuJoCo but may require controller-side masking.")

    space = ctrl_cap.control_space
    if space in (ControlSpace.POSITION, ControlSpace.VELOCITY, ControlSpace.INTVELOCITY):
        limited = np.asarray(model.actuator_ctrllimited, dtype=bool)
        rng = np.asarray(model.actuator_ctrlrange).reshape(-1, 2)
        for a in enabled:
            if not limited[a]:
                notes.append(f"Enabled actuator {a} lacks ctrlrange limits required for servo control.")
            elif _bad_range(*rng[a]):
                notes.append(f"Invalid ctrlrange for enabled actuator {a}: [{rng[a][0]}, {rng[a][1]}]")
    if space == ControlSpace.INTVELOCITY:
        actlim = np.asarray(model.actuator_actlimited, dtype=bool)
        arng = np.asarray(model.actuator_actrange).reshape(-1, 2)
        for a in enabled:
            if not actlim[a]:
                notes.append(f"Enabled actuator {a} has no activation limits (actlimited=0) under intvelocity control.")
            elif _bad_range(*arng[a]):
                notes.append(f"Invalid actrange for enabled actuator {a}: [{arng[a][0]}, {arng[a][1]}]")
    if space == ControlSpace.TORQUE:
        flim = np.asarray(model.actuator_forcelimited, dtype=bool)
        frng = np.asarray(model.actuator_forcerange).reshape(-1, 2)
        for a in enabled:
            if flim[a] and _bad_range(*frng[a]):
                notes.append(f"Invalid forcerange for enabled actuator {a}: [{frng[a][0]}, {frng[a][1]}]")

    notes.append("Note: joint/tendon constraints or other clamps may still limit motion/force beyond actuator-level checks.")
    return CompatibilityReport(ok=not reasons, reasons=reasons, warnings=notes)


__all__ = ["CompatibilityReport", "check_controller_compat"]
