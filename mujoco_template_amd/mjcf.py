"""MJCF-subset model compiler: Python face of ``mjb_model_load_xml`` (``csrc/mjb_mjcf.cpp``, host C++ inside the library).

The reference never parses XML itself: ``ModelHandle.from_xml_path`` hands the
file to ``mj.MjModel.from_xml_path`` (reference ``mujoco_template/model.py:22-37``),
i.e. to the compiler inside the third-party ``mujoco`` C library, which is not available on either
box.  The replacement for that call lives in the library, behind the C ABI (``include/mjbatch.h``:
``mjb_model_load_xml`` / ``mjb_model_load_xml_string``): it compiles the MJCF subset used by the reference's example models
and its test model (``examples/*/**.xml``, ``tests/test_mujoco_template.py:40-61``) into the flat table of named arrays
whose field names follow MuJoCo's ``mjModel``; :func:`compile_xml_path` / :func:`compile_xml_string` call it and present the
table as a :class:`CompiledModel`, so the batched engine, the CPU oracle and the Python front (``compat``-style readers of
``actuator_*`` / ``jnt_*``) share one schema.  Anything outside the subset raises :class:`MjcfError` naming it.

This module also keeps plain-numpy kinematics / Jacobian / mass-matrix helpers (Jacobian form, no recursion): compile-time
cross-checks and test anchors.  ``tests/pymjcf.py`` holds an independent pure-Python restatement of the compiler that the test
suite compares with the C++ one field by field.
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Any

import numpy as np

# ----------------------------------------------------------------------------
# enums (values follow MuJoCo's mjtJoint / mjtGeom / mjtObj ordering)
# ----------------------------------------------------------------------------

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
_GEOM_NAMES = {
    "plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
    "ellipsoid": GEOM_ELLIPSOID, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH,
}
_JNT_NAMES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}

TRN_JOINT, TRN_SITE = 0, 4  # mjTRN_JOINT, mjTRN_SITE
GAIN_FIXED = 0
BIAS_NONE, BIAS_AFFINE = 0, 1
INT_EULER, INT_RK4 = 0, 1

SENS_JOINTPOS, SENS_GYRO, SENS_ACCELEROMETER, SENS_FRAMEQUAT = 0, 1, 2, 3
_SENSOR_DIM = {SENS_JOINTPOS: 1, SENS_GYRO: 3, SENS_ACCELEROMETER: 3, SENS_FRAMEQUAT: 4}

OBJ_BODY, OBJ_XBODY, OBJ_JOINT, OBJ_DOF, OBJ_GEOM, OBJ_SITE = 1, 2, 3, 4, 5, 6
OBJ_TENDON, OBJ_ACTUATOR, OBJ_SENSOR, OBJ_KEY = 18, 19, 20, 24

MINVAL = 1e-15
DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


class MjcfError(ValueError):
    """Raised for MJCF the subset compiler cannot handle (maps to ValueError like mujoco's)."""


# ----------------------------------------------------------------------------
# small math helpers
# ----------------------------------------------------------------------------

def _floats(text: str | None, n: int | None = None) -> np.ndarray | None:
    if text is None:
        return None
    vals = np.array([float(t) for t in text.split()], dtype=np.float64)
    if n is not None and vals.size != n:
        raise MjcfError(f"expected {n} numbers, got {vals.size}: {text!r}")
    return vals


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def quat_to_mat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat_to_quat(m: np.ndarray) -> np.ndarray:
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def z_to_quat(vec: np.ndarray) -> np.ndarray:
    """Quaternion rotating (0,0,1) onto ``vec`` (shortest arc)."""
    v = vec / np.linalg.norm(vec)
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, v)
    s = np.linalg.norm(axis)
    c = float(np.dot(z, v))
    if s < 1e-10:
        return np.array([1.0, 0.0, 0.0, 0.0]) if c > 0 else np.array([0.0, 1.0, 0.0, 0.0])
    axis = axis / s
    ang = math.atan2(s, c)
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


# ----------------------------------------------------------------------------
# compiled model container
# ----------------------------------------------------------------------------

@dataclass
class CompiledModel:
    """Flat mjModel-like table.  All float arrays are float64, ints int32."""

    name: str = ""
    # sizes
    nq: int = 0
    nv: int = 0
    nu: int = 0
    na: int = 0
    nbody: int = 0
    njnt: int = 0
    ngeom: int = 0
    nsite: int = 0
    ntendon: int = 0
    nwrap: int = 0
    nsensor: int = 0
    nsensordata: int = 0
    nkey: int = 0
    npair: int = 0
    nexclude: int = 0
    # options
    timestep: float = 0.002
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    integrator: int = INT_EULER
    density: float = 0.0
    viscosity: float = 0.0
    impratio: float = 1.0
    tolerance: float = 1e-8
    iterations: int = 100
    ls_iterations: int = 50
    disableactuator: int = 0
    meaninertia: float = 1.0
    arrays: dict[str, np.ndarray] = field(default_factory=dict)
    names: dict[int, list[str]] = field(default_factory=dict)

    def __getattr__(self, item: str) -> Any:  # array access by mjModel field name
        arrays = self.__dict__.get("arrays", {})
        if item in arrays:
            return arrays[item]
        raise AttributeError(item)

    def name2id(self, objtype: int, name: str) -> int:
        if objtype == OBJ_XBODY:
            objtype = OBJ_BODY
        try:
            return self.names.get(objtype, []).index(name)
        except ValueError:
            return -1

    def id2name(self, objtype: int, idx: int) -> str | None:
        if objtype == OBJ_XBODY:
            objtype = OBJ_BODY
        lst = self.names.get(objtype, [])
        if 0 <= idx < len(lst) and lst[idx]:
            return lst[idx]
        return None




# ----------------------------------------------------------------------------
# numpy kinematics / mass matrix at a configuration (compile-time only)
# ----------------------------------------------------------------------------

def kinematics_numpy(m: CompiledModel, qpos: np.ndarray) -> dict[str, np.ndarray]:
    A = m.arrays
    nb = m.nbody
    xpos = np.zeros((nb, 3)); xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1)); xmat = np.tile(np.eye(3), (nb, 1, 1))
    xanchor = np.zeros((m.njnt, 3)); xaxis = np.zeros((m.njnt, 3))
    for b in range(1, nb):
        p = A["body_parentid"][b]
        jadr, jnum = A["body_jntadr"][b], A["body_jntnum"][b]
        if jnum == 1 and A["jnt_type"][jadr] == JNT_FREE:
            qa = A["jnt_qposadr"][jadr]
            pos = qpos[qa:qa + 3].copy()
            quat = qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
            xanchor[jadr] = pos
            xaxis[jadr] = np.array([0.0, 0, 1])
        else:
            pos = xpos[p] + xmat[p] @ A["body_pos"][b]
            quat = quat_mul(xquat[p], A["body_quat"][b])
            for j in range(jadr, jadr + jnum):
                qa = A["jnt_qposadr"][j]
                R = quat_to_mat(quat)
                anchor = R @ A["jnt_pos"][j] + pos
                axis = R @ A["jnt_axis"][j]
                xanchor[j], xaxis[j] = anchor, axis
                val = qpos[qa] - A["qpos0"][qa]
                if A["jnt_type"][j] == JNT_SLIDE:
                    pos = pos + axis * val
                else:
                    qloc = np.concatenate([[math.cos(val / 2)], A["jnt_axis"][j] * math.sin(val / 2)])
                    quat = quat_mul(quat, qloc)
                    pos = anchor - quat_to_mat(quat) @ A["jnt_pos"][j]
        quat = quat / np.linalg.norm(quat)
        xpos[b], xquat[b], xmat[b] = pos, quat, quat_to_mat(quat)
    xipos = np.array([xpos[b] + xmat[b] @ A["body_ipos"][b] for b in range(nb)])
    ximat = np.array([xmat[b] @ quat_to_mat(A["body_iquat"][b]) for b in range(nb)])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, xanchor=xanchor, xaxis=xaxis)


def jac_point_numpy(m: CompiledModel, kin: dict[str, np.ndarray], body: int, point: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Translational/rotational Jacobian (3×nv each) of ``point`` fixed to ``body``."""
    A = m.arrays
    jp = np.zeros((3, m.nv)); jr = np.zeros((3, m.nv))
    b = body
    while b > 0:
        for j in range(A["body_jntadr"][b], A["body_jntadr"][b] + A["body_jntnum"][b]):
            d = A["jnt_dofadr"][j]
            t = A["jnt_type"][j]
            if t == JNT_FREE:
                jp[:, d:d + 3] = np.eye(3)
                R = kin["xmat"][b]
                for k in range(3):
                    ax = R[:, k]
                    jr[:, d + 3 + k] = ax
                    jp[:, d + 3 + k] = np.cross(ax, point - kin["xpos"][b])
            elif t == JNT_SLIDE:
                jp[:, d] = kin["xaxis"][j]
            else:
                jr[:, d] = kin["xaxis"][j]
                jp[:, d] = np.cross(kin["xaxis"][j], point - kin["xanchor"][j])
        b = A["body_parentid"][b]
    return jp, jr


def mass_matrix_numpy(m: CompiledModel, kin: dict[str, np.ndarray]) -> np.ndarray:
    """M = Σ_b m Jpᵀ Jp + Jrᵀ I Jr + diag(armature): Jacobian form, independent of the CRB recursion."""
    A = m.arrays
    M = np.diag(A["dof_armature"]).astype(np.float64) if m.nv else np.zeros((0, 0))
    for b in range(1, m.nbody):
        if A["body_mass"][b] <= 0:
            continue
        jp, jr = jac_point_numpy(m, kin, b, kin["xipos"][b])
        Iw = kin["ximat"][b] @ np.diag(A["body_inertia"][b]) @ kin["ximat"][b].T
        M += A["body_mass"][b] * jp.T @ jp + jr.T @ Iw @ jr
    return M


# ----------------------------------------------------------------------------
# public entry points: the compiler is native code behind the C ABI
# ----------------------------------------------------------------------------

def _compiled_from_handle(dm) -> CompiledModel:
    cm = dm.compiled
    object.__setattr__(cm, "_device_model", dm)          # MjModel adopts this handle instead of packing the table again
    return cm


def compile_xml_string(xml_text: str, base_dir: str = ".") -> CompiledModel:
    from ._capi import DeviceModel

    try:
        return _compiled_from_handle(DeviceModel.load_xml_string(xml_text, base_dir))
    except MjcfError:
        raise
    except ValueError as exc:                              # MJB_ERR_MODEL: the compiler's message names what it rejected
        raise MjcfError(str(exc)) from None


def compile_xml_path(xml_path: str) -> CompiledModel:
    from ._capi import DeviceModel

    if not os.path.exists(xml_path):
        raise MjcfError(f"XML file not found: {xml_path}")
    try:
        return _compiled_from_handle(DeviceModel.load_xml(xml_path))
    except MjcfError:
        raise
    except ValueError as exc:
        raise MjcfError(str(exc)) from None


__all__ = ["CompiledModel", "MjcfError", "compile_xml_path", "compile_xml_string"]
