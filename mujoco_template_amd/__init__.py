"""mujoco_template_amd — MI355X-native batched step/rollout engine behind the
``Env`` / ``Controller`` / ``ObservationSpec`` API of ChenDavidTimothy/mujoco-template.

Only the per-step physics path is rebuilt: ``Env.step`` / ``Env.passive`` /
``ModelHandle`` / ``ObservationExtractor`` / ``linearize_discrete`` /
``compute_requested_jacobians`` / ``iterate_passive`` dispatch through the C ABI of
``include/mjbatch.h`` into hand-written HIP kernels (``csrc/``).  ``mj`` is the slice of the
``mujoco`` surface the path uses, backed by the same engine.
"""

from __future__ import annotations

from . import mj
from ._typing import InfoDict, JacobianDict, JacobiansDict, Observation, ObservationArray, ObservationDict, StateSnapshot
from .compat import CompatibilityReport, check_controller_compat
from .control import ControlSpace, Controller, ControllerCapabilities
from .controllers import LinearFeedbackController, PositionTargetDemo, RandomCtrlController, ZeroController
from .env import Env, StepResult
from .exceptions import CompatibilityError, ConfigError, LinearizationError, NameLookupError, TemplateError
from .jacobians import compute_requested_jacobians
from .linearization import linearize_discrete
from .model import ModelHandle
from .observations import ObservationExtractor, ObservationProducer, ObservationSpec
from .logging import DataProbe, StateControlRecorder
from .runtime import StepHook, TrajectoryLogger, iterate_passive, run_passive_headless
from .setpoints import steady_ctrl0

__version__ = "0.1.0"

__all__ = [
    "mj", "TemplateError", "NameLookupError", "CompatibilityError", "LinearizationError", "ConfigError",
    "ControlSpace", "Controller", "ControllerCapabilities", "ObservationSpec", "ObservationExtractor",
    "ObservationProducer", "ModelHandle", "CompatibilityReport", "StepResult", "Env", "ZeroController",
    "PositionTargetDemo", "RandomCtrlController", "LinearFeedbackController", "check_controller_compat", "linearize_discrete",
    "compute_requested_jacobians", "StepHook", "iterate_passive", "run_passive_headless", "steady_ctrl0", "DataProbe",
    "StateControlRecorder", "TrajectoryLogger", "ObservationDict",
    "ObservationArray", "Observation", "JacobianDict", "JacobiansDict", "InfoDict", "StateSnapshot", "__version__",
]
