"""Error convention of the package — mirrors reference ``mujoco_template/exceptions.py:4-21``
so callers can catch the same classes."""

from __future__ import annotations


class TemplateError(RuntimeError):
    """Base exception for the template."""


class NameLookupError(TemplateError):
    """A named entity cannot be resolved inside a model."""


class CompatibilityError(TemplateError):
    """Controller/model compatibility checks failed."""


class LinearizationError(TemplateError):
    """Linearization cannot be performed."""


class ConfigError(TemplateError):
    """Template configuration is invalid."""


__all__ = ["TemplateError", "NameLookupError", "CompatibilityError", "LinearizationError", "ConfigError"]
