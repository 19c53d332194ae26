"""Does this controller fit this model?  Setup-time diagnostics, never on the per-step path.

``check_controller_compat(model, capabilities, enabled_mask)`` returns the reference's ``CompatibilityReport``
(``mujoco_template/compat.py:11-127``): hard *reasons* make ``assert_ok()`` raise ``CompatibilityError``; everything else
is a *warning* that ``Env`` relays once through ``info["compat_warnings"]``.

The checks are table-driven: every control space names the per-actuator (flag, range) pair that has to be sane for
it - servo-like spaces need a bounded ``ctrlrange``, ``intvelocity`` additionally a bounded ``actrange``, torque control only
looks at force limits that are switched on - and one routine walks the enabled actuators of the batch's (shared) model.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np

from .control import ControlSpace, ControllerCapabilities
from .exceptions import CompatibilityError, ConfigError

_SERVO_SPACES = (ControlSpace.POSITION, ControlSpace.VELOCITY, ControlSpace.INTVELOCITY)
_CLOSING_NOTE = "Note: joint/tendon constraints or other clamps may still limit motion/force beyond actuator-level checks."


@dataclass
class CompatibilityReport:
    ok: bool
    reasons: list[str] = field(default_factory=list)
    warnings: list[str] = field(default_factory=list)

    def assert_ok(self) -> None:
        if self.ok:
            return
        bullet_list = "".join(f"\n- {why}" for why in self.reasons)
        raise CompatibilityError("Incompatible controller/model:" + bullet_list)


def _range_is_usable(pair: np.ndarray) -> bool:
    lo, hi = float(pair[0]), float(pair[1])
    return bool(np.isfinite(lo) and np.isfinite(hi) and hi > lo)


def _range_findings(model: Any, actuators: np.ndarray, flag_field: str, range_field: str, label: str,
                    unflagged: str | None) -> list[str]:
    """Walk the enabled actuators: ``unflagged`` (if given) is reported where the limit flag is off, an unusable range where it is on."""
    flags = np.asarray(getattr(model, flag_field), dtype=bool)
    ranges = np.asarray(getattr(model, range_field), dtype=float).reshape(-1, 2)
    found: list[str] = []
    for a in actuators:
        if not flags[a]:
            if unflagged is not None:
                found.append(unflagged.format(a=int(a)))
        elif not _range_is_usable(ranges[a]):
            found.append(f"Invalid {label} for enabled actuator {int(a)}: [{ranges[a][0]}, {ranges[a][1]}]")
    return found


def _group_findings(model: Any, capabilities: ControllerCapabilities, actuators: np.ndarray) -> list[str]:
    if capabilities.actuator_groups is None:
        return []
    asked = {int(g) for g in capabilities.actuator_groups}
    live = {int(g) for g in np.asarray(model.actuator_group)[actuators]}
    found: list[str] = []
    if not live:
        found.append("Controller declared actuator groups but none are currently enabled; continuing without additional group gating.")
    if asked - live:
        found.append(f"Controller requested actuator groups {sorted(asked - live)} but they are not enabled; "
                     "controller will still run with the available groups.")
    if live - asked:
        found.append(f"Enabled actuators include groups {sorted(live - asked)} beyond the controller request; "
                     "behaviour matches MuJoCo but may require controller-side masking.")
    return found


def check_controller_compat(model: Any, ctrl_cap: ControllerCapabilities, enabled_mask: np.ndarray | None) -> CompatibilityReport:
    nu = int(model.nu)
    mask = np.ones(nu, dtype=bool) if enabled_mask is None else np.asarray(enabled_mask, dtype=bool)
    if mask.shape[0] != nu:
        raise ConfigError("enabled_mask must have length model.nu")
    reasons: list[str] = []
    if nu == 0:
        reasons.append("Model has no actuators (nu=0).")
    if not mask.any():
        reasons.append("All actuators are disabled by group selection.")
    live = np.flatnonzero(mask)

    notes = _group_findings(model, ctrl_cap, live)
    space = ctrl_cap.control_space
    if space in _SERVO_SPACES:
        notes += _range_findings(model, live, "actuator_ctrllimited", "actuator_ctrlrange", "ctrlrange",
                                 "Enabled actuator {a} lacks ctrlrange limits required for servo control.")
    if space == ControlSpace.INTVELOCITY:
        notes += _range_findings(model, live, "actuator_actlimited", "actuator_actrange", "actrange",
                                 "Enabled actuator {a} has no activation limits (actlimited=0) under intvelocity control.")
    if space == ControlSpace.TORQUE:
        notes += _range_findings(model, live, "actuator_forcelimited", "actuator_forcerange", "forcerange", None)
    notes.append(_CLOSING_NOTE)
    return CompatibilityReport(ok=not reasons, reasons=reasons, warnings=notes)


__all__ = ["CompatibilityReport", "check_controller_compat"]
