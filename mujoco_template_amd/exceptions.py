"""Error vocabulary of the package and its mapping onto the C ABI.

Callers of the reference catch ``TemplateError`` and its four subclasses (reference
``mujoco_template/exceptions.py:4-21``); the same names exist here so that ``except`` clauses keep working.
What is specific to this engine lives here too: every entry point of ``include/mjbatch.h`` returns a negative
status and leaves a message for ``mjb_last_error()``; :func:`raise_for_status` turns that pair into the
exception class a user of the reference would expect for the same mistake.

====================  =====  ==================================================================
exception             code   raised for
====================  =====  ==================================================================
``ConfigError``        -1    bad argument, unknown array / field name, bad shape
``ValueError``         -2    the compiled-model table was rejected (like MuJoCo's XML compiler)
``TemplateError``      -3    HIP error, no device, anything else
``NameLookupError``    -4    index / name out of range (keyframes, ids)
====================  =====  ==================================================================
"""

from __future__ import annotations

MJB_OK, MJB_ERR_ARG, MJB_ERR_MODEL, MJB_ERR_DEVICE, MJB_ERR_LOOKUP = 0, -1, -2, -3, -4


class TemplateError(RuntimeError):
    """Root of the hierarchy: anything the engine or its Python front cannot carry out."""


class ConfigError(TemplateError):
    """The request itself is malformed (arguments, shapes, names of fields, option values)."""


class NameLookupError(TemplateError):
    """A site / body / geom / keyframe / joint name or index does not exist in the model."""


class CompatibilityError(TemplateError):
    """Controller and model do not fit together (control space, actuator groups, missing actuators)."""


class LinearizationError(TemplateError):
    """The finite-difference transition matrices cannot be produced for this state / model."""


_STATUS_TO_EXCEPTION: dict[int, type[Exception]] = {
    MJB_ERR_ARG: ConfigError,
    MJB_ERR_MODEL: ValueError,
    MJB_ERR_DEVICE: TemplateError,
    MJB_ERR_LOOKUP: NameLookupError,
}


def raise_for_status(status: int, message: str) -> None:
    """Raise the exception that belongs to a non-zero C-ABI ``status`` (no-op for ``MJB_OK``)."""
    if status != MJB_OK:
        raise _STATUS_TO_EXCEPTION.get(int(status), TemplateError)(message)


__all__ = ["TemplateError", "NameLookupError", "CompatibilityError", "LinearizationError", "ConfigError", "raise_for_status"]
