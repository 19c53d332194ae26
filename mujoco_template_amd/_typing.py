"""Shapes and containers that cross the public API.

The names are the reference's (``mujoco_template/_typing.py:5-23``) so annotations in user code stay valid; the
meaning is batch-aware:

* an *observation* is either a mapping ``field -> array`` or one flat array; with ``batch == 1`` arrays have the
  reference's shapes (``[nq]``, ``[k, 3]`` ...), with ``batch > 1`` they gain a leading ``[batch]`` axis;
* a *Jacobian block* maps ``"jacp"`` / ``"jacr"`` to ``[3, nv]`` (``[batch, 3, nv]``) arrays, and the Jacobians of one
  step map request tokens (``"site:tip"``, ``"bodycom:torso"`` ...) to such blocks;
* ``info`` values are whatever a step can report: scalars, strings, arrays, the (A, B) matrices of one or several
  sub-steps, Jacobian dictionaries of one or several sub-steps;
* a *state snapshot* is the dictionary ``state_utils`` saves and restores (``qpos qvel act ctrl time``).
"""

from __future__ import annotations

from typing import Union

from numpy import ndarray

# observations
ObservationArray = ndarray
ObservationDict = dict[str, ndarray]
Observation = Union[ObservationDict, ObservationArray]

# Jacobians: token -> {"jacp": ..., "jacr": ...}
JacobianDict = dict[str, ndarray]
JacobiansDict = dict[str, JacobianDict]

# StepResult.info
_Scalar = Union[str, float, int]
_PerSubstep = Union[list[ndarray], tuple[ndarray, ...], list[JacobiansDict], tuple[JacobiansDict, ...]]
InfoValue = Union[_Scalar, ndarray, list[str], JacobiansDict, _PerSubstep]
InfoDict = dict[str, InfoValue]

# state_utils
StateSnapshot = dict[str, Union[ndarray, float, None]]

__all__ = ["ObservationDict", "ObservationArray", "Observation", "JacobianDict", "JacobiansDict", "InfoValue", "InfoDict", "StateSnapshot"]
