"""Requested Jacobians (reference ``mujoco_template/jacobians.py:12-83``): token grammar
``site:<n>`` / ``body:<n>`` / ``bodycom:<n>`` / ``subtreecom:<n>``; all tokens of one call are
evaluated in ONE batched kernel launch (``mjb_jac``).  Blocks are ``[3, nv]`` for batch 1 and
``[batch, 3, nv]`` otherwise.
"""

from __future__ import annotations

from collections.abc import Iterable
from typing import Any

import numpy as np

from . import mj
from ._typing import JacobiansDict
from .exceptions import ConfigError, NameLookupError

_KINDS = {"site": (0, mj.mjtObj.mjOBJ_SITE, "Site"), "body": (1, mj.mjtObj.mjOBJ_BODY, "Body"),
          "bodycom": (2, mj.mjtObj.mjOBJ_BODY, "Body"), "subtreecom": (3, mj.mjtObj.mjOBJ_BODY, "Body")}


def _parse_jacobian_token(token: str) -> tuple[str, str | None]:
    if token == "com":
        return ("com", None)
    head, sep, name = token.partition(":")
    if sep and head in _KINDS:
        return (head, name)
    raise ConfigError(f"Unknown jacobian token: {token}")


def compute_requested_jacobians(model: Any, data: Any, tokens: Iterable[str]) -> JacobiansDict:
    tokens = list(tokens)
    kinds: list[int] = []
    ids: list[int] = []
    for token in tokens:
        kind, name = _parse_jacobian_token(token)
        if kind == "com":
            raise ConfigError("'com' jacobian is ambiguous; request 'bodycom:<name>' or 'subtreecom:<name>'.")
        code, objtype, label = _KINDS[kind]
        idx = mj.mj_name2id(model, objtype, name or "")
        if idx < 0:
            raise NameLookupError(f"{label} not found: {name}")
        kinds.append(code)
        ids.append(idx)
    out: JacobiansDict = {}
    if not tokens:
        return out
    data.push_host_edits()
    jp, jr = data.sim.jac(kinds, ids)
    squeeze = data.batch == 1
    for r, token in enumerate(tokens):
        block = {"jacp": np.array(jp[0, r] if squeeze else jp[:, r])}
        if kinds[r] in (0, 1):
            block["jacr"] = np.array(jr[0, r] if squeeze else jr[:, r])
        out[token] = block
    return out


__all__ = ["compute_requested_jacobians"]
