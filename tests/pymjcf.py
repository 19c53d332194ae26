"""Independent pure-Python restatement of the MJCF-subset compiler — TEST INFRASTRUCTURE.

The product compiler is native (``mujoco_template_amd/csrc/mjb_mjcf.cpp`` behind ``mjb_model_load_xml``).  This module is the
round-1 Python compiler, kept so that the test suite can compare the two implementations field by field on every model
(``tests/test_mjcf.py::test_native_compiler_matches_the_python_restatement``): the oracle and the HIP path both consume the
compiler's output, so a second implementation of it is the only check that does not pass through both sides.
MuJoCo's compiler semantics (defaults classes, fromto, inertia from geoms, autolimits, invweight0, contact parameter mixing) are
restated from its documented behaviour [MJ-KNOWLEDGE, SURVEY.md §8(c)].
"""

from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from typing import Any

import numpy as np

from mujoco_template_amd.mjcf import *  # noqa: F401,F403  (enums)
from mujoco_template_amd.mjcf import (_floats, _GEOM_NAMES, _JNT_NAMES, _SENSOR_DIM, BIAS_AFFINE, BIAS_NONE, DEFAULT_SOLIMP, DEFAULT_SOLREF, GEOM_BOX,
                                      GEOM_CAPSULE, GEOM_CYLINDER, GEOM_ELLIPSOID, GEOM_MESH, GEOM_PLANE, GEOM_SPHERE, INT_EULER, INT_RK4, JNT_BALL,
                                      JNT_FREE, JNT_HINGE, JNT_SLIDE, MINVAL, OBJ_ACTUATOR, OBJ_BODY, OBJ_GEOM, OBJ_JOINT, OBJ_KEY, OBJ_SENSOR, OBJ_SITE,
                                      OBJ_TENDON, SENS_ACCELEROMETER, SENS_FRAMEQUAT, SENS_GYRO, SENS_JOINTPOS, TRN_JOINT, TRN_SITE, CompiledModel,
                                      MjcfError, jac_point_numpy, kinematics_numpy, mass_matrix_numpy, mat_to_quat, quat_mul, quat_to_mat, z_to_quat)

# ----------------------------------------------------------------------------
# defaults handling
# ----------------------------------------------------------------------------

_ACT_TAGS = ("general", "motor", "position", "velocity")


class _Defaults:
    def __init__(self) -> None:
        self.classes: dict[str, dict[str, dict[str, str]]] = {"main": {}}

    def add(self, elem: ET.Element, parent: str | None) -> None:
        cname = elem.get("class", "main" if parent is None else None)
        if cname is None:
            raise MjcfError("nested <default> needs a class attribute")
        base = {} if parent is None else {k: dict(v) for k, v in self.classes[parent].items()}
        if cname == "main" and parent is None:
            base = {k: dict(v) for k, v in self.classes["main"].items()}
        for child in elem:
            if child.tag == "default":
                continue
            tag = "actuator" if child.tag in _ACT_TAGS else child.tag
            base.setdefault(tag, {}).update(child.attrib)
        self.classes[cname] = base
        for child in elem:
            if child.tag == "default":
                self.add(child, cname)

    def resolve(self, tag: str, elem: ET.Element, childclass: str | None) -> dict[str, str]:
        cname = elem.get("class") or childclass or "main"
        if cname not in self.classes:
            raise MjcfError(f"unknown default class {cname!r}")
        key = "actuator" if tag in _ACT_TAGS else tag
        out = dict(self.classes[cname].get(key, {}))
        out.update(elem.attrib)
        return out



# ----------------------------------------------------------------------------
# schema: what the compiler honours, what it may ignore, and nothing else
# ----------------------------------------------------------------------------
# Every element / attribute is either HONOURED (read by the compiler below), IGNORABLE (rendering, naming and bookkeeping that
# cannot change the physics of the supported subset) or REJECTED with an MjcfError naming it — a model outside the subset must
# never simulate silently with different physics (<equality>, <option><flag>, frictionloss, ... were dropped unseen).
_ORIENT = {"quat", "axisangle", "euler", "xyaxes", "zaxis"}
_VISUAL_ATTRS = {"rgba", "material", "group"}                  # "group" of geoms / sites is a rendering group
_SCHEMA_TOP = {"compiler", "option", "default", "worldbody", "tendon", "actuator", "sensor", "contact", "keyframe", "include",
               # no physics in the supported subset:
               "asset", "visual", "statistic", "size", "custom"}
_SCHEMA_ATTRS: dict[str, tuple[set[str], set[str]]] = {
    # tag: (honoured, ignorable)
    "mujoco": ({"model"}, set()),
    "compiler": ({"angle", "autolimits"}, {"meshdir", "texturedir", "assetdir", "strippath", "discardvisual", "balanceinertia", "boundmass",
                                            "boundinertia", "fusestatic", "usethread", "alignfree"}),
    "option": ({"timestep", "gravity", "integrator", "density", "viscosity", "impratio", "tolerance", "iterations", "cone", "solver",
                "jacobian"}, {"ls_iterations", "ls_tolerance", "noslip_tolerance", "ccd_tolerance", "mpr_tolerance", "apirate"}),
    "body": ({"name", "pos", "childclass"} | _ORIENT, {"user"}),
    "joint": ({"name", "class", "type", "pos", "axis", "range", "limited", "damping", "stiffness", "armature", "margin", "ref", "springref",
               "solreflimit", "solimplimit"}, {"group", "user"}),
    "freejoint": ({"name"}, {"group"}),
    "geom": ({"name", "class", "type", "size", "pos", "fromto", "contype", "conaffinity", "condim", "friction", "solref", "solimp", "solmix",
              "margin", "gap", "priority", "density", "mass"} | _ORIENT, _VISUAL_ATTRS | {"user", "mesh", "fitscale"}),
    "site": ({"name", "class", "pos"} | _ORIENT, _VISUAL_ATTRS | {"size", "type", "fromto", "user"}),
    "fixed": ({"name", "class", "limited", "range", "margin", "solreflimit", "solimplimit"}, _VISUAL_ATTRS | {"user", "width"}),
    "tendon/joint": ({"joint", "coef"}, set()),
    "motor": ({"name", "class", "joint", "site", "gear", "ctrllimited", "ctrlrange", "forcelimited", "forcerange", "group"}, {"user"}),
    "position": ({"name", "class", "joint", "site", "gear", "ctrllimited", "ctrlrange", "forcelimited", "forcerange", "group", "kp", "kv"}, {"user"}),
    "general": ({"name", "class", "joint", "site", "gear", "ctrllimited", "ctrlrange", "forcelimited", "forcerange", "group", "dyntype",
                 "gaintype", "biastype", "gainprm", "biasprm"}, {"user"}),
    "jointpos": ({"name", "joint"}, {"noise", "cutoff", "user"}),
    "gyro": ({"name", "site"}, {"noise", "cutoff", "user"}),
    "accelerometer": ({"name", "site"}, {"noise", "cutoff", "user"}),
    "framequat": ({"name", "objtype", "objname"}, {"noise", "cutoff", "user"}),
    "exclude": ({"name", "body1", "body2"}, set()),
    "key": ({"name", "qpos", "qvel", "ctrl", "time"}, set()),
    "include": ({"file"}, set()),
    "camera": (set(), None), "light": (set(), None),               # None: every attribute ignorable (rendering only)
}
# attributes whose NON-DEFAULT presence changes the physics and that the engine does not implement: named rejections
_REJECT_ATTRS = {
    "joint": {"frictionloss": "joint frictionloss", "actuatorfrcrange": "actuatorfrcrange", "actuatorfrclimited": "actuatorfrclimited",
              "solreffriction": "joint friction constraints", "solimpfriction": "joint friction constraints", "springdamper": "springdamper"},
    "geom": {"fluidshape": "ellipsoid fluid model", "fluidcoef": "ellipsoid fluid model"},
    "body": {"mocap": "mocap bodies", "gravcomp": "gravity compensation"},
    "fixed": {"frictionloss": "tendon frictionloss", "stiffness": "tendon springs", "damping": "tendon damping", "springlength": "tendon springs"},
    "option": {"wind": "wind", "magnetic": None, "o_margin": "contact overrides", "o_solref": "contact overrides", "o_solimp": "contact overrides",
               "o_friction": "contact overrides", "noslip_iterations": "the noslip solver", "actuatorgroupdisable": "actuatorgroupdisable (use opt.disableactuator)"},
    "compiler": {"coordinate": None, "eulerseq": None, "settotalmass": "settotalmass", "inertiafromgeom": None, "inertiagrouprange": "inertiagrouprange"},
}
# rejected attributes that are harmless at these values (MuJoCo's defaults, or what the shipped models state explicitly)
_REJECT_OK_VALUES = {("compiler", "coordinate"): {"local"}, ("compiler", "eulerseq"): {"xyz"}, ("compiler", "inertiafromgeom"): {"true", "auto"},
                     ("compiler", "settotalmass"): {"-1"}, ("joint", "frictionloss"): {"0"}, ("fixed", "frictionloss"): {"0"},
                     ("fixed", "stiffness"): {"0"}, ("fixed", "damping"): {"0"}, ("body", "mocap"): {"false"}, ("body", "gravcomp"): {"0"},
                     ("option", "wind"): {"0 0 0"}, ("option", "noslip_iterations"): {"0"}, ("option", "magnetic"): None}     # None: any value


def _check_attrs(tag: str, elem: ET.Element, where: str) -> None:
    key = "tendon/joint" if (tag == "joint" and where == "tendon") else tag
    if key not in _SCHEMA_ATTRS:
        raise MjcfError(f"<{tag}> in <{where}> is outside the supported subset")
    honoured, ignorable = _SCHEMA_ATTRS[key]
    rejects = _REJECT_ATTRS.get(key, {})
    for attr, val in elem.attrib.items():
        if attr in honoured or ignorable is None or attr in ignorable:
            continue
        if attr in rejects:
            ok = _REJECT_OK_VALUES.get((key, attr), set())
            if ok is None or " ".join(val.split()) in ok or (ok and _is_float(val) and any(_is_float(o) and float(o) == float(val) for o in ok)):
                continue
            what = rejects[attr] or f"{attr}={val!r}"
            raise MjcfError(f"<{tag} {attr}={val!r}>: {what} is outside the supported subset")
        raise MjcfError(f"<{tag}> attribute {attr!r} is not recognised by this compiler (supported subset; it would be ignored silently otherwise)")


def _is_float(text: str) -> bool:
    try:
        float(text)
        return True
    except ValueError:
        return False


def _validate_schema(root: ET.Element) -> None:
    """Walk the (include-expanded) tree once and reject everything the compiler would otherwise drop without a word."""
    _check_attrs("mujoco", root, "")
    for sec in root:
        if sec.tag not in _SCHEMA_TOP:
            raise MjcfError(f"<{sec.tag}> is outside the supported subset (e.g. <equality> constraints, <deformable>, <extension> are not implemented)")
        if sec.tag in ("asset", "visual", "statistic", "size", "custom"):
            continue
        if sec.tag == "compiler":
            _check_attrs("compiler", sec, "mujoco")
            for ch in sec:
                raise MjcfError(f"<compiler><{ch.tag}> is outside the supported subset")
        elif sec.tag == "option":
            _check_attrs("option", sec, "mujoco")
            for ch in sec:
                if ch.tag != "flag":
                    raise MjcfError(f"<option><{ch.tag}> is outside the supported subset")
                for attr, val in ch.attrib.items():          # every flag must sit at MuJoCo's default: none of them is implemented as a switch
                    default = "disable" if attr in ("override", "energy", "fwdinv", "invdiscrete", "multiccd", "island") else "enable"
                    if val != default:
                        raise MjcfError(f"<option><flag {attr}={val!r}>: option flags are outside the supported subset (all stay at MuJoCo's defaults)")
        elif sec.tag == "default":
            _validate_defaults(sec)
        elif sec.tag == "worldbody":
            _validate_body(sec, top=True)
        elif sec.tag == "tendon":
            for t in sec:
                if t.tag != "fixed":
                    raise MjcfError("only fixed tendons are inside the supported subset")
                _check_attrs("fixed", t, "tendon")
                for w in t:
                    _check_attrs(w.tag, w, "tendon")
        elif sec.tag == "actuator":
            for e in sec:
                if e.tag not in ("motor", "position", "general"):
                    raise MjcfError(f"actuator <{e.tag}> is outside the supported subset")
                _check_attrs(e.tag, e, "actuator")
        elif sec.tag == "sensor":
            for e in sec:
                if e.tag not in ("jointpos", "gyro", "accelerometer", "framequat"):
                    raise MjcfError(f"sensor <{e.tag}> is outside the supported subset")
                _check_attrs(e.tag, e, "sensor")
        elif sec.tag == "contact":
            for e in sec:
                if e.tag != "exclude":
                    raise MjcfError("<contact><pair> is outside the supported subset")
                _check_attrs("exclude", e, "contact")
        elif sec.tag == "keyframe":
            for e in sec:
                if e.tag != "key":
                    raise MjcfError(f"<keyframe><{e.tag}> is outside the supported subset")
                _check_attrs("key", e, "keyframe")


def _validate_defaults(elem: ET.Element) -> None:
    for ch in elem:
        if ch.tag == "default":
            _validate_defaults(ch)
        elif ch.tag in ("camera", "light", "material", "mesh"):
            continue
        elif ch.tag in ("joint", "geom", "site", "motor", "position", "general"):
            _check_attrs(ch.tag, ch, "default")
        elif ch.tag == "tendon":
            _check_attrs("fixed", ch, "default")
        else:
            raise MjcfError(f"<default><{ch.tag}> is outside the supported subset")


def _validate_body(elem: ET.Element, top: bool = False) -> None:
    for ch in elem:
        if ch.tag == "body":
            _check_attrs("body", ch, "worldbody")
            _validate_body(ch)
        elif ch.tag in ("joint", "freejoint", "geom", "site", "camera", "light"):
            _check_attrs(ch.tag, ch, "body")
        elif ch.tag == "inertial":
            raise MjcfError("<inertial> is outside the supported subset (inertia comes from geoms)")
        else:
            raise MjcfError(f"<{ch.tag}> inside a body is outside the supported subset")

# ----------------------------------------------------------------------------
# the compiler
# ----------------------------------------------------------------------------

class _Compiler:
    def __init__(self, root: ET.Element, base_dir: str):
        self.base_dir = base_dir
        self.root = self._expand_includes(root, base_dir)
        self.defaults = _Defaults()
        self.angle_scale = math.pi / 180.0
        self.autolimits = True
        self.m = CompiledModel(name=self.root.get("model", ""))
        # growing lists
        self.bodies: list[dict[str, Any]] = []
        self.joints: list[dict[str, Any]] = []
        self.geoms: list[dict[str, Any]] = []
        self.sites: list[dict[str, Any]] = []

    # -- includes -------------------------------------------------------------
    def _expand_includes(self, elem: ET.Element, base_dir: str) -> ET.Element:
        new_children: list[ET.Element] = []
        for child in list(elem):
            if child.tag == "include":
                path = os.path.join(base_dir, child.get("file", ""))
                if not os.path.exists(path):
                    raise MjcfError(f"include file not found: {path}")
                sub = ET.parse(path).getroot()
                sub = self._expand_includes(sub, os.path.dirname(path))
                new_children.extend(list(sub))
            else:
                new_children.append(self._expand_includes(child, base_dir))
        for c in list(elem):
            elem.remove(c)
        for c in new_children:
            elem.append(c)
        return elem

    # -- orientation ------------------------------------------------------------
    def _orientation(self, a: dict[str, str]) -> np.ndarray:
        if "quat" in a:
            q = _floats(a["quat"], 4)
            return q / np.linalg.norm(q)
        if "axisangle" in a:
            v = _floats(a["axisangle"], 4)
            ang = v[3] * self.angle_scale
            ax = v[:3] / np.linalg.norm(v[:3])
            return np.concatenate([[math.cos(ang / 2)], ax * math.sin(ang / 2)])
        if "euler" in a:
            e = _floats(a["euler"], 3) * self.angle_scale
            q = np.array([1.0, 0, 0, 0])
            for i, ang in enumerate(e):  # default eulerseq "xyz", intrinsic
                ax = np.zeros(3)
                ax[i] = 1.0
                q = quat_mul(q, np.concatenate([[math.cos(ang / 2)], ax * math.sin(ang / 2)]))
            return q
        if "xyaxes" in a:
            v = _floats(a["xyaxes"], 6)
            x = v[:3] / np.linalg.norm(v[:3])
            y = v[3:] - np.dot(v[3:], x) * x
            y /= np.linalg.norm(y)
            z = np.cross(x, y)
            return mat_to_quat(np.stack([x, y, z], axis=1))
        if "zaxis" in a:
            return z_to_quat(_floats(a["zaxis"], 3))
        return np.array([1.0, 0.0, 0.0, 0.0])

    # -- top level ---------------------------------------------------------------
    def compile(self) -> CompiledModel:
        root, m = self.root, self.m
        if root.tag != "mujoco":
            raise MjcfError("root element must be <mujoco>")
        _validate_schema(root)
        for comp in root.findall("compiler"):
            ang = comp.get("angle")
            if ang == "radian":
                self.angle_scale = 1.0
            elif ang == "degree":
                self.angle_scale = math.pi / 180.0
            if comp.get("autolimits") is not None:
                self.autolimits = comp.get("autolimits") == "true"
        for opt in root.findall("option"):
            if opt.get("timestep"):
                m.timestep = float(opt.get("timestep"))
            if opt.get("gravity"):
                m.gravity = _floats(opt.get("gravity"), 3)
            integ = opt.get("integrator")
            if integ is not None:
                if integ == "Euler":
                    m.integrator = INT_EULER
                elif integ == "RK4":
                    m.integrator = INT_RK4
                else:
                    raise MjcfError(f"integrator {integ!r} is outside the supported subset (Euler, RK4)")
            if opt.get("density"):
                m.density = float(opt.get("density"))
            if opt.get("viscosity"):
                m.viscosity = float(opt.get("viscosity"))
            if opt.get("impratio"):
                m.impratio = float(opt.get("impratio"))
            if opt.get("tolerance"):
                m.tolerance = float(opt.get("tolerance"))
            if opt.get("iterations"):
                m.iterations = int(opt.get("iterations"))
            for unsupported in ("cone", "solver", "jacobian"):
                val = opt.get(unsupported)
                if val is not None and val not in ("pyramidal", "Newton", "dense", "auto"):
                    raise MjcfError(f"option {unsupported}={val!r} is outside the supported subset")
        for d in root.findall("default"):
            self.defaults.add(d, None)

        # world body
        self.bodies.append(dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]),
                                jnt=[], geoms=[], childclass=None))
        for wb in root.findall("worldbody"):
            self._body_children(wb, 0, None)
        self._finalize_tree()
        self._tendons(root)
        self._actuators(root)
        self._sensors(root)
        self._contacts(root)
        self._keyframes(root)
        self._set_const()
        return m

    # -- kinematic tree -------------------------------------------------------
    def _body_children(self, elem: ET.Element, body_id: int, childclass: str | None) -> None:
        for child in elem:
            if child.tag == "body":
                cc = child.get("childclass", childclass)
                a = child.attrib
                b = dict(name=a.get("name", ""), parent=body_id,
                         pos=_floats(a.get("pos", "0 0 0"), 3), quat=self._orientation(dict(a)),
                         jnt=[], geoms=[], childclass=cc)
                self.bodies.append(b)
                self._body_children(child, len(self.bodies) - 1, cc)
            elif child.tag in ("joint", "freejoint"):
                self._joint(child, body_id, childclass)
            elif child.tag == "geom":
                self._geom(child, body_id, childclass)
            elif child.tag == "site":
                self._site(child, body_id, childclass)
            elif child.tag == "inertial":
                raise MjcfError("<inertial> is outside the supported subset (inertia comes from geoms)")
            # camera / light / others: not part of the physics path

    def _limited(self, a: dict[str, str], key: str, rng_key: str) -> bool:
        val = a.get(key, "auto")
        if val == "true":
            return True
        if val == "false":
            return False
        return self.autolimits and rng_key in a

    def _joint(self, elem: ET.Element, body_id: int, childclass: str | None) -> None:
        if body_id == 0:
            raise MjcfError("joints cannot be attached to the world body")
        if elem.tag == "freejoint":
            a = dict(elem.attrib)
            jtype = JNT_FREE
        else:
            a = self.defaults.resolve("joint", elem, childclass)
            jtype = _JNT_NAMES[a.get("type", "hinge")]
        if jtype == JNT_BALL:
            raise MjcfError("ball joints are outside the supported subset")
        axis = _floats(a.get("axis", "0 0 1"), 3)
        axis = axis / np.linalg.norm(axis)
        rng = _floats(a.get("range", "0 0"), 2)
        scale = self.angle_scale if jtype == JNT_HINGE else 1.0
        solimp = np.array(DEFAULT_SOLIMP)
        if "solimplimit" in a:
            v = _floats(a["solimplimit"])
            solimp[: v.size] = v
        solref = np.array(DEFAULT_SOLREF)
        if "solreflimit" in a:
            v = _floats(a["solreflimit"])
            solref[: v.size] = v
        limited = self._limited(a, "limited", "range") and jtype in (JNT_HINGE, JNT_SLIDE)
        if jtype == JNT_FREE:
            damping = stiffness = armature = 0.0
        else:
            damping = float(a.get("damping", 0))
            stiffness = float(a.get("stiffness", 0))
            armature = float(a.get("armature", 0))
        j = dict(name=a.get("name", ""), type=jtype, body=body_id,
                 pos=_floats(a.get("pos", "0 0 0"), 3), axis=axis, limited=limited,
                 range=rng * scale, damping=damping, stiffness=stiffness, armature=armature,
                 margin=float(a.get("margin", 0)), solref=solref, solimp=solimp,
                 ref=float(a.get("ref", 0)) * scale, springref=float(a.get("springref", 0)) * scale)
        self.joints.append(j)
        self.bodies[body_id]["jnt"].append(len(self.joints) - 1)

    def _geom(self, elem: ET.Element, body_id: int, childclass: str | None) -> None:
        a = self.defaults.resolve("geom", elem, childclass)
        gtype = _GEOM_NAMES[a.get("type", "sphere")]
        size = np.zeros(3)
        if "size" in a:
            v = _floats(a["size"])
            size[: v.size] = v
        pos = _floats(a.get("pos", "0 0 0"), 3)
        quat = self._orientation(a)
        if "fromto" in a:
            if gtype not in (GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_ELLIPSOID):
                raise MjcfError("fromto requires capsule/cylinder/box/ellipsoid")
            ft = _floats(a["fromto"], 6)
            vec = ft[:3] - ft[3:]
            length = np.linalg.norm(vec)
            pos = 0.5 * (ft[:3] + ft[3:])
            quat = z_to_quat(vec)
            if gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
                size[1] = length / 2
            else:
                size[2] = length / 2
        if gtype == GEOM_MESH:
            size[:] = 0.0
            # meshes are accepted as VISUALS only: a mesh that should carry mass would need its volume (the .obj is never read)
            if not ("mass" in a and float(a["mass"]) == 0.0):
                raise MjcfError(f"mesh geom {a.get('name', '')!r} needs mass=\"0\": mesh inertia is outside the supported subset")

        solref = np.array(DEFAULT_SOLREF)
        if "solref" in a:
            v = _floats(a["solref"])
            solref[: v.size] = v
        solimp = np.array(DEFAULT_SOLIMP)
        if "solimp" in a:
            v = _floats(a["solimp"])
            solimp[: v.size] = v
        friction = np.array([1.0, 0.005, 0.0001])
        if "friction" in a:
            v = _floats(a["friction"])
            friction[: v.size] = v
        g = dict(name=a.get("name", ""), type=gtype, body=body_id, pos=pos, quat=quat, size=size,
                 contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)),
                 condim=int(a.get("condim", 3)), friction=friction, solref=solref, solimp=solimp,
                 solmix=float(a.get("solmix", 1)), margin=float(a.get("margin", 0)), gap=float(a.get("gap", 0)),
                 priority=int(a.get("priority", 0)), density=float(a.get("density", 1000)),
                 mass=(float(a["mass"]) if "mass" in a else None), group=int(a.get("group", 0)))
        if g["condim"] not in (1, 3):
            raise MjcfError("only condim 1 and 3 are inside the supported subset")
        self.geoms.append(g)
        self.bodies[body_id]["geoms"].append(len(self.geoms) - 1)

    def _site(self, elem: ET.Element, body_id: int, childclass: str | None) -> None:
        a = self.defaults.resolve("site", elem, childclass)
        self.sites.append(dict(name=a.get("name", ""), body=body_id,
                               pos=_floats(a.get("pos", "0 0 0"), 3), quat=self._orientation(a)))

    # -- geom mass properties ------------------------------------------------
    @staticmethod
    def _geom_volume_inertia(gtype: int, size: np.ndarray) -> tuple[float, np.ndarray]:
        """Volume and unit-density diagonal inertia (about geom centre, geom frame)."""
        if gtype == GEOM_SPHERE:
            r = size[0]
            vol = 4.0 / 3.0 * math.pi * r ** 3
            return vol, np.full(3, 0.4 * vol * r * r)
        if gtype == GEOM_CAPSULE:
            r, h = size[0], size[1]
            vc = math.pi * r * r * 2 * h
            vs = 4.0 / 3.0 * math.pi * r ** 3
            izz = vc * r * r / 2 + vs * 0.4 * r * r
            ixx = vc * (3 * r * r + 4 * h * h) / 12 + vs * (0.4 * r * r + h * h + 0.75 * h * r)
            return vc + vs, np.array([ixx, ixx, izz])
        if gtype == GEOM_CYLINDER:
            r, h = size[0], size[1]
            vol = math.pi * r * r * 2 * h
            return vol, np.array([vol * (3 * r * r + 4 * h * h) / 12] * 2 + [vol * r * r / 2])
        if gtype == GEOM_ELLIPSOID:
            a, b, c = size
            vol = 4.0 / 3.0 * math.pi * a * b * c
            return vol, vol / 5 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
        if gtype == GEOM_BOX:
            a, b, c = size
            vol = 8 * a * b * c
            return vol, vol / 3 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
        return 0.0, np.zeros(3)  # plane, mesh (mass-0 visual only), hfield

    def _finalize_tree(self) -> None:
        m = self.m
        nb = len(self.bodies)
        A = m.arrays
        m.nbody, m.njnt, m.ngeom, m.nsite = nb, len(self.joints), len(self.geoms), len(self.sites)
        A["body_parentid"] = np.array([b["parent"] for b in self.bodies], dtype=np.int32)
        A["body_pos"] = np.array([b["pos"] for b in self.bodies])
        A["body_quat"] = np.array([b["quat"] for b in self.bodies])
        # joints / dofs / qpos
        qposadr, dofadr = [], []
        nq = nv = 0
        qpos0: list[float] = []
        qpos_spring: list[float] = []
        dof_body, dof_jnt, dof_arm, dof_damp = [], [], [], []
        for jid, j in enumerate(self.joints):
            qposadr.append(nq)
            dofadr.append(nv)
            if j["type"] == JNT_FREE:
                b = self.bodies[j["body"]]
                qpos0 += list(b["pos"]) + list(b["quat"])
                qpos_spring += list(b["pos"]) + list(b["quat"])
                nq += 7
                ndof = 6
            else:
                qpos0.append(j["ref"])
                qpos_spring.append(j["springref"])
                nq += 1
                ndof = 1
            for _ in range(ndof):
                dof_body.append(j["body"])
                dof_jnt.append(jid)
                dof_arm.append(j["armature"])
                dof_damp.append(j["damping"])
            nv += ndof
        m.nq, m.nv = nq, nv
        # a free joint must be the only joint of a top-level body
        for j in self.joints:
            if j["type"] == JNT_FREE and (self.bodies[j["body"]]["parent"] != 0 or len(self.bodies[j["body"]]["jnt"]) != 1):
                raise MjcfError("free joint must be the only joint of a child of the world body")
        A["jnt_type"] = np.array([j["type"] for j in self.joints], dtype=np.int32)
        A["jnt_qposadr"] = np.array(qposadr, dtype=np.int32)
        A["jnt_dofadr"] = np.array(dofadr, dtype=np.int32)
        A["jnt_bodyid"] = np.array([j["body"] for j in self.joints], dtype=np.int32)
        A["jnt_pos"] = np.array([j["pos"] for j in self.joints]).reshape(-1, 3)
        A["jnt_axis"] = np.array([j["axis"] for j in self.joints]).reshape(-1, 3)
        A["jnt_limited"] = np.array([j["limited"] for j in self.joints], dtype=np.int32)
        A["jnt_range"] = np.array([j["range"] for j in self.joints]).reshape(-1, 2)
        A["jnt_stiffness"] = np.array([j["stiffness"] for j in self.joints], dtype=np.float64)
        A["jnt_margin"] = np.array([j["margin"] for j in self.joints], dtype=np.float64)
        A["jnt_solref"] = np.array([j["solref"] for j in self.joints]).reshape(-1, 2)
        A["jnt_solimp"] = np.array([j["solimp"] for j in self.joints]).reshape(-1, 5)
        A["qpos0"] = np.array(qpos0, dtype=np.float64)
        A["qpos_spring"] = np.array(qpos_spring, dtype=np.float64)
        A["dof_bodyid"] = np.array(dof_body, dtype=np.int32)
        A["dof_jntid"] = np.array(dof_jnt, dtype=np.int32)
        A["dof_armature"] = np.array(dof_arm, dtype=np.float64)
        A["dof_damping"] = np.array(dof_damp, dtype=np.float64)
        body_jntadr = np.full(nb, -1, dtype=np.int32)
        body_jntnum = np.zeros(nb, dtype=np.int32)
        body_dofadr = np.full(nb, -1, dtype=np.int32)
        body_dofnum = np.zeros(nb, dtype=np.int32)
        for bid, b in enumerate(self.bodies):
            if b["jnt"]:
                body_jntadr[bid] = b["jnt"][0]
                body_jntnum[bid] = len(b["jnt"])
                body_dofadr[bid] = dofadr[b["jnt"][0]]
                body_dofnum[bid] = sum(6 if self.joints[j]["type"] == JNT_FREE else 1 for j in b["jnt"])
        A["body_jntadr"], A["body_jntnum"] = body_jntadr, body_jntnum
        A["body_dofadr"], A["body_dofnum"] = body_dofadr, body_dofnum
        # dof parent chain: previous dof in the same body, else last dof of the nearest ancestor with dofs
        dof_parent = np.full(nv, -1, dtype=np.int32)
        for d in range(nv):
            bid = dof_body[d]
            if d > body_dofadr[bid]:
                dof_parent[d] = d - 1
            else:
                p = self.bodies[bid]["parent"]
                while p > 0 and body_dofnum[p] == 0:
                    p = self.bodies[p]["parent"]
                if p > 0:
                    dof_parent[d] = body_dofadr[p] + body_dofnum[p] - 1
        A["dof_parentid"] = dof_parent
        # weld / root ids, depth
        weld = np.zeros(nb, dtype=np.int32)
        rootid = np.zeros(nb, dtype=np.int32)
        depth = np.zeros(nb, dtype=np.int32)
        for bid in range(1, nb):
            p = self.bodies[bid]["parent"]
            weld[bid] = bid if body_jntnum[bid] > 0 else weld[p]
            rootid[bid] = bid if p == 0 else rootid[p]
            depth[bid] = depth[p] + 1
        A["body_weldid"], A["body_rootid"], A["body_depth"] = weld, rootid, depth
        # geoms / sites
        A["geom_type"] = np.array([g["type"] for g in self.geoms], dtype=np.int32)
        A["geom_bodyid"] = np.array([g["body"] for g in self.geoms], dtype=np.int32)
        A["geom_pos"] = np.array([g["pos"] for g in self.geoms]).reshape(-1, 3)
        A["geom_quat"] = np.array([g["quat"] for g in self.geoms]).reshape(-1, 4)
        A["geom_size"] = np.array([g["size"] for g in self.geoms]).reshape(-1, 3)
        A["geom_contype"] = np.array([g["contype"] for g in self.geoms], dtype=np.int32)
        A["geom_conaffinity"] = np.array([g["conaffinity"] for g in self.geoms], dtype=np.int32)
        A["geom_condim"] = np.array([g["condim"] for g in self.geoms], dtype=np.int32)
        A["geom_friction"] = np.array([g["friction"] for g in self.geoms]).reshape(-1, 3)
        A["geom_solref"] = np.array([g["solref"] for g in self.geoms]).reshape(-1, 2)
        A["geom_solimp"] = np.array([g["solimp"] for g in self.geoms]).reshape(-1, 5)
        A["geom_margin"] = np.array([g["margin"] for g in self.geoms], dtype=np.float64)
        A["geom_gap"] = np.array([g["gap"] for g in self.geoms], dtype=np.float64)
        A["site_bodyid"] = np.array([s["body"] for s in self.sites], dtype=np.int32)
        A["site_pos"] = np.array([s["pos"] for s in self.sites]).reshape(-1, 3)
        A["site_quat"] = np.array([s["quat"] for s in self.sites]).reshape(-1, 4)
        # body inertial properties from geoms
        body_mass = np.zeros(nb)
        body_ipos = np.zeros((nb, 3))
        body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
        body_inertia = np.zeros((nb, 3))
        for bid, b in enumerate(self.bodies):
            if bid == 0:
                continue
            parts = []
            for gid in b["geoms"]:
                g = self.geoms[gid]
                vol, inr = self._geom_volume_inertia(g["type"], g["size"])
                if vol <= 0:
                    continue
                mass = g["mass"] if g["mass"] is not None else g["density"] * vol
                if mass <= 0:
                    continue
                parts.append((mass, g["pos"], quat_to_mat(g["quat"]), inr * (mass / vol)))
            mtot = sum(p[0] for p in parts)
            if mtot <= 0:
                if body_jntnum[bid] > 0:
                    raise MjcfError(f"moving body {b['name']!r} has zero mass")
                continue
            com = sum(p[0] * p[1] for p in parts) / mtot
            I = np.zeros((3, 3))
            for mass, pos, R, inr in parts:
                d = pos - com
                I += R @ np.diag(inr) @ R.T + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            evals, evecs = np.linalg.eigh(I)
            order = np.argsort(-evals)  # descending, as mju_eig3
            evals, evecs = evals[order], evecs[:, order]
            if np.allclose(I, np.diag(np.diag(I)), atol=1e-14 * max(1.0, np.abs(I).max())) :
                # already diagonal: keep the body axes (no frame permutation), like MuJoCo does for aligned geoms
                evals, evecs = np.diag(I).copy(), np.eye(3)
            if np.linalg.det(evecs) < 0:
                evecs[:, 2] = -evecs[:, 2]
            body_mass[bid] = mtot
            body_ipos[bid] = com
            body_iquat[bid] = mat_to_quat(evecs)
            body_inertia[bid] = evals
        A["body_mass"], A["body_ipos"], A["body_iquat"], A["body_inertia"] = body_mass, body_ipos, body_iquat, body_inertia
        # subtree mass
        sub = body_mass.copy()
        for bid in range(nb - 1, 0, -1):
            sub[self.bodies[bid]["parent"]] += sub[bid]
        A["body_subtreemass"] = sub
        m.names[OBJ_BODY] = [b["name"] for b in self.bodies]
        m.names[OBJ_JOINT] = [j["name"] for j in self.joints]
        m.names[OBJ_GEOM] = [g["name"] for g in self.geoms]
        m.names[OBJ_SITE] = [s["name"] for s in self.sites]

    # -- tendons ---------------------------------------------------------------
    def _tendons(self, root: ET.Element) -> None:
        m, A = self.m, self.m.arrays
        tend: list[dict[str, Any]] = []
        wrap_obj: list[int] = []
        wrap_prm: list[float] = []
        for tsec in root.findall("tendon"):
            for t in tsec:
                if t.tag != "fixed":
                    raise MjcfError("only fixed tendons are inside the supported subset")
                a = self.defaults.resolve("tendon", t, None)
                adr = len(wrap_obj)
                for w in t.findall("joint"):
                    jid = m.name2id(OBJ_JOINT, w.get("joint", ""))
                    if jid < 0:
                        raise MjcfError(f"tendon joint not found: {w.get('joint')}")
                    if self.joints[jid]["type"] not in (JNT_HINGE, JNT_SLIDE):
                        raise MjcfError("fixed tendon joints must be hinge or slide")
                    wrap_obj.append(jid)
                    wrap_prm.append(float(w.get("coef", 1)))
                solref = np.array(DEFAULT_SOLREF)
                solimp = np.array(DEFAULT_SOLIMP)
                if "solreflimit" in a:
                    v = _floats(a["solreflimit"]); solref[: v.size] = v
                if "solimplimit" in a:
                    v = _floats(a["solimplimit"]); solimp[: v.size] = v
                tend.append(dict(name=a.get("name", ""), adr=adr, num=len(wrap_obj) - adr,
                                 limited=self._limited(a, "limited", "range"),
                                 range=_floats(a.get("range", "0 0"), 2), margin=float(a.get("margin", 0)),
                                 solref=solref, solimp=solimp))
        m.ntendon, m.nwrap = len(tend), len(wrap_obj)
        A["tendon_adr"] = np.array([t["adr"] for t in tend], dtype=np.int32)
        A["tendon_num"] = np.array([t["num"] for t in tend], dtype=np.int32)
        A["tendon_limited"] = np.array([t["limited"] for t in tend], dtype=np.int32)
        A["tendon_range"] = np.array([t["range"] for t in tend]).reshape(-1, 2)
        A["tendon_margin"] = np.array([t["margin"] for t in tend], dtype=np.float64)
        A["tendon_solref"] = np.array([t["solref"] for t in tend]).reshape(-1, 2)
        A["tendon_solimp"] = np.array([t["solimp"] for t in tend]).reshape(-1, 5)
        A["wrap_objid"] = np.array(wrap_obj, dtype=np.int32)
        A["wrap_prm"] = np.array(wrap_prm, dtype=np.float64)
        m.names[OBJ_TENDON] = [t["name"] for t in tend]

    # -- actuators ---------------------------------------------------------------
    def _actuators(self, root: ET.Element) -> None:
        m, A = self.m, self.m.arrays
        acts: list[dict[str, Any]] = []
        for asec in root.findall("actuator"):
            for e in asec:
                if e.tag not in ("motor", "position", "general"):
                    raise MjcfError(f"actuator <{e.tag}> is outside the supported subset")
                a = self.defaults.resolve(e.tag, e, None)
                gear = np.zeros(6)
                gear[0] = 1.0
                if "gear" in a:
                    v = _floats(a["gear"])
                    gear[:] = 0.0
                    gear[: v.size] = v
                if "joint" in a:
                    trntype = TRN_JOINT
                    trnid = m.name2id(OBJ_JOINT, a["joint"])
                    if trnid < 0:
                        raise MjcfError(f"actuator joint not found: {a['joint']}")
                    if self.joints[trnid]["type"] not in (JNT_HINGE, JNT_SLIDE):
                        raise MjcfError("joint transmission supports hinge/slide only")
                elif "site" in a:
                    trntype = TRN_SITE
                    trnid = m.name2id(OBJ_SITE, a["site"])
                    if trnid < 0:
                        raise MjcfError(f"actuator site not found: {a['site']}")
                    if "refsite" in a:
                        raise MjcfError("refsite is outside the supported subset")
                else:
                    raise MjcfError("actuator needs joint= or site= transmission")
                gainprm = np.zeros(3)
                biasprm = np.zeros(3)
                biastype = BIAS_NONE
                if e.tag == "motor":
                    gainprm[0] = 1.0
                elif e.tag == "position":
                    kp = float(a.get("kp", 1))
                    kv = float(a.get("kv", 0))
                    gainprm[0] = kp
                    biasprm[:] = [0.0, -kp, -kv]
                    biastype = BIAS_AFFINE
                else:  # general: fixed gain / affine bias only
                    if a.get("dyntype", "none") != "none" or a.get("gaintype", "fixed") != "fixed":
                        raise MjcfError("general actuator: only dyntype=none, gaintype=fixed supported")
                    gainprm[0] = 1.0
                    if "gainprm" in a:
                        v = _floats(a["gainprm"]); gainprm[: min(3, v.size)] = v[:3]
                    bt = a.get("biastype", "none")
                    if bt == "affine":
                        biastype = BIAS_AFFINE
                        v = _floats(a.get("biasprm", "0 0 0")); biasprm[: min(3, v.size)] = v[:3]
                    elif bt != "none":
                        raise MjcfError("general actuator: biastype must be none/affine")
                acts.append(dict(
                    name=a.get("name", ""), trntype=trntype, trnid=trnid, gear=gear,
                    gainprm=gainprm, biasprm=biasprm, biastype=biastype,
                    ctrllimited=self._limited(a, "ctrllimited", "ctrlrange"),
                    ctrlrange=_floats(a.get("ctrlrange", "0 0"), 2),
                    forcelimited=self._limited(a, "forcelimited", "forcerange"),
                    forcerange=_floats(a.get("forcerange", "0 0"), 2),
                    group=int(a.get("group", 0))))
        m.nu = len(acts)
        A["actuator_trntype"] = np.array([x["trntype"] for x in acts], dtype=np.int32)
        A["actuator_trnid"] = np.array([[x["trnid"], -1] for x in acts], dtype=np.int32).reshape(-1, 2)
        A["actuator_gear"] = np.array([x["gear"] for x in acts]).reshape(-1, 6)
        A["actuator_gainprm"] = np.array([x["gainprm"] for x in acts]).reshape(-1, 3)
        A["actuator_biasprm"] = np.array([x["biasprm"] for x in acts]).reshape(-1, 3)
        A["actuator_biastype"] = np.array([x["biastype"] for x in acts], dtype=np.int32)
        A["actuator_ctrllimited"] = np.array([x["ctrllimited"] for x in acts], dtype=bool)
        A["actuator_ctrlrange"] = np.array([x["ctrlrange"] for x in acts]).reshape(-1, 2)
        A["actuator_forcelimited"] = np.array([x["forcelimited"] for x in acts], dtype=bool)
        A["actuator_forcerange"] = np.array([x["forcerange"] for x in acts]).reshape(-1, 2)
        A["actuator_actlimited"] = np.zeros(len(acts), dtype=bool)
        A["actuator_actrange"] = np.zeros((len(acts), 2))
        A["actuator_group"] = np.array([x["group"] for x in acts], dtype=np.int32)
        m.names[OBJ_ACTUATOR] = [x["name"] for x in acts]

    # -- sensors -------------------------------------------------------------------
    def _sensors(self, root: ET.Element) -> None:
        m, A = self.m, self.m.arrays
        sens: list[tuple[str, int, int, int]] = []
        adr = 0
        for ssec in root.findall("sensor"):
            for e in ssec:
                if e.tag == "jointpos":
                    jid = m.name2id(OBJ_JOINT, e.get("joint", ""))
                    if jid < 0:
                        raise MjcfError(f"sensor joint not found: {e.get('joint')}")
                    stype, obj = SENS_JOINTPOS, jid
                elif e.tag in ("gyro", "accelerometer"):
                    sid = m.name2id(OBJ_SITE, e.get("site", ""))
                    if sid < 0:
                        raise MjcfError(f"sensor site not found: {e.get('site')}")
                    stype, obj = (SENS_GYRO if e.tag == "gyro" else SENS_ACCELEROMETER), sid
                elif e.tag == "framequat":
                    if e.get("objtype") != "site":
                        raise MjcfError("framequat: only objtype=site supported")
                    sid = m.name2id(OBJ_SITE, e.get("objname", ""))
                    if sid < 0:
                        raise MjcfError(f"sensor site not found: {e.get('objname')}")
                    stype, obj = SENS_FRAMEQUAT, sid
                else:
                    raise MjcfError(f"sensor <{e.tag}> is outside the supported subset")
                sens.append((e.get("name", ""), stype, obj, adr))
                adr += _SENSOR_DIM[stype]
        m.nsensor, m.nsensordata = len(sens), adr
        A["sensor_type"] = np.array([s[1] for s in sens], dtype=np.int32)
        A["sensor_objid"] = np.array([s[2] for s in sens], dtype=np.int32)
        A["sensor_adr"] = np.array([s[3] for s in sens], dtype=np.int32)
        m.names[OBJ_SENSOR] = [s[0] for s in sens]

    # -- collision pair list ----------------------------------------------------------
    def _contacts(self, root: ET.Element) -> None:
        """Static candidate pair list = MuJoCo's per-step filter applied once.

        Filters [MJ-KNOWLEDGE]: same weld body; contype/conaffinity masks;
        weld-parent/child unless one side is the world; ``<exclude>``;
        both geoms fixed to the world.  Pair parameters follow
        ``mj_contactParam`` for equal priority: condim = max, friction =
        element-wise max, solref/solimp mixed by solmix weights,
        margin/gap = max.
        """
        m, A = self.m, self.m.arrays
        excl: set[tuple[int, int]] = set()
        for csec in root.findall("contact"):
            for e in csec:
                if e.tag == "exclude":
                    b1 = m.name2id(OBJ_BODY, e.get("body1", ""))
                    b2 = m.name2id(OBJ_BODY, e.get("body2", ""))
                    if b1 < 0 or b2 < 0:
                        raise MjcfError("exclude: body not found")
                    excl.add((min(b1, b2), max(b1, b2)))
                else:
                    raise MjcfError("<contact><pair> is outside the supported subset")
        m.nexclude = len(excl)
        weld = A["body_weldid"]
        parent = A["body_parentid"]
        supported = {
            (GEOM_PLANE, GEOM_SPHERE), (GEOM_PLANE, GEOM_CAPSULE), (GEOM_PLANE, GEOM_BOX),
            (GEOM_PLANE, GEOM_ELLIPSOID), (GEOM_SPHERE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_CAPSULE),
            (GEOM_CAPSULE, GEOM_CAPSULE),
        }
        pairs: list[dict[str, Any]] = []
        for g1 in range(m.ngeom):
            for g2 in range(g1 + 1, m.ngeom):
                G1, G2 = self.geoms[g1], self.geoms[g2]
                b1, b2 = G1["body"], G2["body"]
                if b1 == b2:
                    continue
                if not ((G1["contype"] & G2["conaffinity"]) or (G2["contype"] & G1["conaffinity"])):
                    continue
                w1, w2 = weld[b1], weld[b2]
                if w1 == w2:
                    continue
                wp1, wp2 = weld[parent[w1]], weld[parent[w2]]
                if w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                    continue
                if (min(b1, b2), max(b1, b2)) in excl:
                    continue
                a, b = (g1, g2) if G1["type"] <= G2["type"] else (g2, g1)
                GA, GB = self.geoms[a], self.geoms[b]
                if (GA["type"], GB["type"]) not in supported:
                    raise MjcfError(
                        f"collision pair types ({GA['type']},{GB['type']}) are outside the supported subset")
                if GA["priority"] != GB["priority"]:
                    raise MjcfError("geom priority is outside the supported subset")
                mix = GA["solmix"] / (GA["solmix"] + GB["solmix"]) if (GA["solmix"] + GB["solmix"]) > MINVAL else 0.5
                fr = np.maximum(GA["friction"], GB["friction"])
                if GA["solref"][0] > 0 and GB["solref"][0] > 0:
                    solref = mix * GA["solref"] + (1 - mix) * GB["solref"]
                else:
                    solref = np.minimum(GA["solref"], GB["solref"])
                pairs.append(dict(
                    g1=a, g2=b, condim=max(GA["condim"], GB["condim"]),
                    friction=np.array([fr[0], fr[0], fr[1], fr[2], fr[2]]),
                    solref=solref, solimp=mix * GA["solimp"] + (1 - mix) * GB["solimp"],
                    margin=max(GA["margin"], GB["margin"]), gap=max(GA["gap"], GB["gap"])))
        m.npair = len(pairs)
        A["pair_geom1"] = np.array([p["g1"] for p in pairs], dtype=np.int32)
        A["pair_geom2"] = np.array([p["g2"] for p in pairs], dtype=np.int32)
        A["pair_condim"] = np.array([p["condim"] for p in pairs], dtype=np.int32)
        A["pair_friction"] = np.array([p["friction"] for p in pairs]).reshape(-1, 5)
        A["pair_solref"] = np.array([p["solref"] for p in pairs]).reshape(-1, 2)
        A["pair_solimp"] = np.array([p["solimp"] for p in pairs]).reshape(-1, 5)
        A["pair_margin"] = np.array([p["margin"] for p in pairs], dtype=np.float64)
        A["pair_gap"] = np.array([p["gap"] for p in pairs], dtype=np.float64)

    # -- keyframes ---------------------------------------------------------------------
    def _keyframes(self, root: ET.Element) -> None:
        m, A = self.m, self.m.arrays
        keys: list[dict[str, Any]] = []
        for ksec in root.findall("keyframe"):
            for e in ksec.findall("key"):
                qpos = _floats(e.get("qpos")) if e.get("qpos") else A["qpos0"].copy()
                qvel = _floats(e.get("qvel")) if e.get("qvel") else np.zeros(m.nv)
                ctrl = _floats(e.get("ctrl")) if e.get("ctrl") else np.zeros(m.nu)
                if qpos.size != m.nq or qvel.size != m.nv or ctrl.size != m.nu:
                    raise MjcfError(f"keyframe {e.get('name')!r}: size mismatch (qpos {qpos.size} vs nq {m.nq})")
                keys.append(dict(name=e.get("name", ""), qpos=qpos, qvel=qvel, ctrl=ctrl, time=float(e.get("time", 0))))
        m.nkey = len(keys)
        A["key_qpos"] = np.array([k["qpos"] for k in keys], dtype=np.float64).reshape(len(keys), m.nq)
        A["key_qvel"] = np.array([k["qvel"] for k in keys], dtype=np.float64).reshape(len(keys), m.nv)
        A["key_ctrl"] = np.array([k["ctrl"] for k in keys], dtype=np.float64).reshape(len(keys), m.nu)
        A["key_time"] = np.array([k["time"] for k in keys], dtype=np.float64)
        m.names[OBJ_KEY] = [k["name"] for k in keys]

    # -- constants that need physics at qpos0 (mj_setConst) -------------------------------
    def _set_const(self) -> None:
        m, A = self.m, self.m.arrays
        kin = kinematics_numpy(m, A["qpos0"])
        M = mass_matrix_numpy(m, kin)
        nv = m.nv
        A["qM0"] = M
        if nv > 0:
            Minv = np.linalg.inv(M)
            m.meaninertia = float(np.trace(M) / nv)
        else:
            Minv = np.zeros((0, 0))
            m.meaninertia = 1.0
        # dof_invweight0: diag(M^-1), averaged over the 3 translational / 3 rotational dofs of a free joint
        dinv = np.diag(Minv).copy() if nv else np.zeros(0)
        for jid in range(m.njnt):
            if A["jnt_type"][jid] == JNT_FREE:
                d0 = A["jnt_dofadr"][jid]
                dinv[d0:d0 + 3] = dinv[d0:d0 + 3].mean()
                dinv[d0 + 3:d0 + 6] = dinv[d0 + 3:d0 + 6].mean()
        A["dof_invweight0"] = dinv
        # body_invweight0: (trace of J M^-1 J^T)/3 for translation and rotation at the body com
        binv = np.zeros((m.nbody, 2))
        for b in range(1, m.nbody):
            if A["body_weldid"][b] == 0:
                continue
            jp, jr = jac_point_numpy(m, kin, b, kin["xipos"][b])
            binv[b, 0] = max(MINVAL, np.trace(jp @ Minv @ jp.T) / 3)
            binv[b, 1] = max(MINVAL, np.trace(jr @ Minv @ jr.T) / 3)
        A["body_invweight0"] = binv
        # tendon_invweight0 and length at qpos0
        tinv = np.zeros(m.ntendon)
        for t in range(m.ntendon):
            J = np.zeros(nv)
            for w in range(A["tendon_adr"][t], A["tendon_adr"][t] + A["tendon_num"][t]):
                J[A["jnt_dofadr"][A["wrap_objid"][w]]] = A["wrap_prm"][w]
            tinv[t] = max(MINVAL, float(J @ Minv @ J))
        A["tendon_invweight0"] = tinv
        # ancestor table: dof d acts on body b  (used for Jacobians)
        anc = np.zeros((m.nbody, max(nv, 1)), dtype=np.int32)
        for b in range(1, m.nbody):
            p = b
            while p > 0:
                if A["body_dofnum"][p] > 0:
                    anc[b, A["body_dofadr"][p]:A["body_dofadr"][p] + A["body_dofnum"][p]] = 1
                p = A["body_parentid"][p]
        A["body_dofmask"] = anc[:, :nv] if nv else np.zeros((m.nbody, 0), dtype=np.int32)




def compile_xml_string(xml_text: str, base_dir: str = ".") -> CompiledModel:
    try:
        root = ET.fromstring(xml_text)
    except ET.ParseError as exc:
        raise MjcfError(f"XML parse error: {exc}") from exc
    return _Compiler(root, base_dir).compile()


def compile_xml_path(xml_path: str) -> CompiledModel:
    if not os.path.exists(xml_path):
        raise MjcfError(f"XML file not found: {xml_path}")
    with open(xml_path, "r", encoding="utf-8") as fh:
        text = fh.read()
    return compile_xml_string(text, os.path.dirname(os.path.abspath(xml_path)))
