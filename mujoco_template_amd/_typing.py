"""Type aliases shared by the front-end modules (batched variants of the reference's aliases)."""

from __future__ import annotations

import numpy as np

ObservationDict = dict[str, np.ndarray]
ObservationArray = np.ndarray
Observation = ObservationDict | ObservationArray
JacobianDict = dict[str, np.ndarray]
JacobiansDict = dict[str, JacobianDict]
InfoValue = (
    str | float | int | np.ndarray | list[str] | JacobiansDict | list[np.ndarray] | tuple[np.ndarray, ...]
    | list[JacobiansDict] | tuple[JacobiansDict, ...]
)
InfoDict = dict[str, InfoValue]
StateSnapshot = dict[str, np.ndarray | float | None]

__all__ = ["ObservationDict", "ObservationArray", "Observation", "JacobianDict", "JacobiansDict", "InfoDict", "StateSnapshot"]
