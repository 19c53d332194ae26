"""``mj`` — the slice of the third-party ``mujoco`` Python surface that the reference's
hot path uses (stub: reference ``mujoco_template/mujoco.pyi:1-75``), re-implemented as a
thin front of the batched MI355X engine (C ABI ``include/mjbatch.h``).

``MjModel`` wraps a model compiled by :mod:`mujoco_template_amd.mjcf`; ``MjData`` owns
``batch`` replicas resident on one GPU.  With ``batch == 1`` attribute shapes are those of
real MuJoCo (``data.qpos.shape == (nq,)``); with ``batch > 1`` every array gains a leading
batch axis.  Host arrays are float64 *mirrors* of the device state:

* reading an attribute pulls it from the device if the device copy is newer and always
  returns the same numpy object (so ``np.shares_memory(obs["qpos"], data.qpos)`` holds, as the
  reference's zero-copy test requires, tests/test_mujoco_template.py:241-252);
* in-place edits of a mirror (``data.qpos[:] = ...``, as reference controllers do in
  ``prepare``) are detected against a shadow copy and pushed before the next device call.

There is no CPU physics here: every ``mj_*`` call below launches HIP kernels.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from . import mjcf
from ._capi import MIRROR_FIELDS, BatchSim, DeviceModel
from .exceptions import ConfigError, LinearizationError, TemplateError
from .mjcf import CompiledModel

# ----------------------------------------------------------------------------------
# enums
# ----------------------------------------------------------------------------------


class mjtObj:
    mjOBJ_UNKNOWN = 0
    mjOBJ_BODY = mjcf.OBJ_BODY
    mjOBJ_XBODY = mjcf.OBJ_XBODY
    mjOBJ_JOINT = mjcf.OBJ_JOINT
    mjOBJ_DOF = mjcf.OBJ_DOF
    mjOBJ_GEOM = mjcf.OBJ_GEOM
    mjOBJ_SITE = mjcf.OBJ_SITE
    mjOBJ_TENDON = mjcf.OBJ_TENDON
    mjOBJ_ACTUATOR = mjcf.OBJ_ACTUATOR
    mjOBJ_SENSOR = mjcf.OBJ_SENSOR
    mjOBJ_KEY = mjcf.OBJ_KEY


class mjtJoint:
    mjJNT_FREE = mjcf.JNT_FREE
    mjJNT_BALL = mjcf.JNT_BALL
    mjJNT_SLIDE = mjcf.JNT_SLIDE
    mjJNT_HINGE = mjcf.JNT_HINGE


class FatalError(RuntimeError):
    pass


# ----------------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------------


class _Option:
    def __init__(self, model: "MjModel"):
        object.__setattr__(self, "_model", model)

    @property
    def timestep(self) -> float:
        return float(self._model._c.timestep)

    @property
    def gravity(self) -> np.ndarray:
        return self._model._c.gravity

    @property
    def disableactuator(self) -> int:
        return int(self._model._c.disableactuator)

    @disableactuator.setter
    def disableactuator(self, mask: int) -> None:
        self._model._c.disableactuator = int(mask)
        self._model._device_model().set_disableactuator(int(mask))

    @property
    def iterations(self) -> int:
        return int(self._model._c.iterations)

    @property
    def tolerance(self) -> float:
        return float(self._model._c.tolerance)

    def __setattr__(self, key: str, value: Any) -> None:
        if key == "disableactuator":
            type(self).disableactuator.fset(self, value)  # type: ignore[attr-defined]
        elif key in ("iterations", "tolerance"):
            setattr(self._model._c, key, value)
            self._model._device_model().set_solver(int(self._model._c.iterations), float(self._model._c.tolerance))
        else:
            raise AttributeError(f"opt.{key} is read-only in the batched engine (timestep is baked into the device model)")


class MjModel:
    """Compiled model (immutable except ``opt.disableactuator`` / solver knobs)."""

    def __init__(self, compiled: CompiledModel):
        self._c = compiled
        self._dm: DeviceModel | None = compiled.__dict__.get("_device_model")      # the handle the native compiler returned, if any
        self.opt = _Option(self)

    # construction ---------------------------------------------------------------
    @classmethod
    def from_xml_path(cls, xml_path: str) -> "MjModel":
        return cls(mjcf.compile_xml_path(xml_path))

    @classmethod
    def from_xml_string(cls, xml_text: str) -> "MjModel":
        return cls(mjcf.compile_xml_string(xml_text))

    @classmethod
    def from_binary_path(cls, mjb_path: str) -> "MjModel":
        """Model saved by :meth:`save_binary` (C ABI ``mjb_model_load``; the flat table format of this engine, not MuJoCo's .mjb)."""
        dm = DeviceModel.load(mjb_path)
        model = cls(dm.compiled)
        model._dm = dm
        return model

    def save_binary(self, mjb_path: str) -> None:
        self._device_model().save(mjb_path)

    def _device_model(self) -> DeviceModel:
        if self._dm is None:
            self._dm = DeviceModel(self._c)
        return self._dm

    # sizes / tables ---------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        c = self.__dict__.get("_c")
        if c is None:
            raise AttributeError(name)
        if name in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ntendon", "nsensor", "nsensordata", "nkey"):
            return int(getattr(c, name))
        if name in c.arrays:
            arr = c.arrays[name]
            per = {"actuator_ctrlrange": 2, "actuator_forcerange": 2, "actuator_actrange": 2, "jnt_range": 2}
            return arr.reshape(-1, per[name]) if name in per else arr
        raise AttributeError(name)

    @property
    def compiled(self) -> CompiledModel:
        return self._c

    # named accessors ---------------------------------------------------------------
    # ``model.body("torso").id`` / ``model.joint(j).name``: the slice of the ``mujoco`` bindings' named-access API the reference
    # uses (``examples/humanoid/controllers/lqr.py:93-95,222``).  Names resolve through the C ABI (mjb_model_name2id / id2name).
    def _named(self, objtype: int, count: int, kind: str, key: "int | str") -> "_NamedView":
        if isinstance(key, str):
            idx = mj_name2id(self, objtype, key)
            if idx < 0:
                valid = [n for n in (mj_id2name(self, objtype, i) for i in range(count)) if n]
                raise KeyError(f"Invalid name '{key}'. Valid names: {valid}")
            return _NamedView(int(idx), key)
        idx = int(key)
        if not 0 <= idx < count:
            raise IndexError(f"{kind} id {idx} out of range [0, {count})")
        return _NamedView(idx, mj_id2name(self, objtype, idx) or "")

    def body(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_BODY, self.nbody, "body", key)

    def joint(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_JOINT, self.njnt, "joint", key)

    def geom(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_GEOM, self.ngeom, "geom", key)

    def site(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_SITE, self.nsite, "site", key)

    def tendon(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_TENDON, self.ntendon, "tendon", key)

    def actuator(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_ACTUATOR, self.nu, "actuator", key)

    def sensor(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_SENSOR, self.nsensor, "sensor", key)

    def key(self, key: "int | str") -> "_NamedView":
        return self._named(mjtObj.mjOBJ_KEY, self.nkey, "keyframe", key)


class _NamedView:
    """What ``model.body(...)`` and friends return: the element's ``id`` and ``name``."""

    __slots__ = ("id", "name")

    def __init__(self, idx: int, name: str):
        self.id, self.name = idx, name

    def __repr__(self) -> str:
        return f"<_NamedView id={self.id} name={self.name!r}>"


# ----------------------------------------------------------------------------------
# data
# ----------------------------------------------------------------------------------

_STATE = ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart")       # writable, pushed when edited
_DERIVED = ("xpos", "xquat", "xipos", "site_xpos", "geom_xpos", "subtree_com", "sensordata",
            "qfrc_inverse", "actuator_moment")  # read-only outputs (the last two: filled by mj_inverse)
_SHAPE3 = {"xpos": 3, "xquat": 4, "xipos": 3, "site_xpos": 3, "geom_xpos": 3, "subtree_com": 3}


class MjData:
    def __init__(self, model: MjModel, batch: int = 1, *, dtype: str = "float32", device: int = 0, lanes: int = 0,
                 nconmax: int = 0, nefcmax: int = 0, env0: int = 0, specialize: bool | None = None):
        if not isinstance(model, MjModel):
            raise TypeError("MjData(model): model must be an MjModel")
        if batch < 1:
            raise ConfigError("batch must be >= 1")
        self.model = model
        self.batch = int(batch)
        self._sim = BatchSim(model._device_model(), self.batch, dtype=dtype, lanes=lanes, nconmax=nconmax, nefcmax=nefcmax,
                             device=device, env0=env0, specialize=specialize)
        # state mirrors = numpy views over the library's pinned float64 block (mjb_host_view): ONE packed copy per direction
        self._mirror: dict[str, np.ndarray] = {name: self._sim.host_view(name) for name in MIRROR_FIELDS}
        self._flags = self._sim.host_view("engine_flags")          # one sticky word, refreshed by the same packed copy
        self._state_stale = True                                   # the device state is newer than the mirror block
        self._dev_newer: set[str] = set(_DERIVED)                  # derived arrays are pulled one by one, on demand
        self.act = np.zeros(0) if self.batch == 1 else np.zeros((self.batch, 0))
        self.engine_warnings: list[str] = []
        self._warned: set[str] = set()
        for name in _DERIVED:
            _, n, _ = self._sim.array_ptr(name)
            if name == "actuator_moment":
                shape = (self.batch, model.nu, model.nv)        # dense; MuJoCo's CSR triplet is derived in the properties below
            elif name in _SHAPE3:
                shape = (self.batch, n // _SHAPE3[name], _SHAPE3[name])
            else:
                shape = (self.batch, n)
            self._mirror[name] = np.zeros(shape)
        self._views = {k: (v[0] if self.batch == 1 else v) for k, v in self._mirror.items() if k != "time"}

    # -- mirror protocol ---------------------------------------------------------
    @property
    def sim(self) -> BatchSim:
        return self._sim

    def _refresh_shadow(self) -> None:
        # the shadow copy the in-place edits are detected against lives in the library (mjb_mirror_commit: one memcpy per field)
        self._sim.mirror_commit(63)
        self._state_stale = False

    def _pull(self, name: str) -> None:
        if name in MIRROR_FIELDS:
            if self._state_stale:
                self._sim.sync_to_host()
                self._refresh_shadow()
                self._check_engine_counters()
        elif name in self._dev_newer:
            m = self._mirror[name]
            m[...] = self._sim.get(name).reshape(m.shape)
            self._dev_newer.discard(name)

    def _edited_mask(self) -> int:
        """Bit mask (order of ``MIRROR_FIELDS``) of the mirrors the user edited in place since they were last refreshed."""
        if self._state_stale:
            return 0                                               # nothing pulled since the last launch: nothing to compare against
        return self._sim.mirror_edited_mask()                      # bitwise comparison with the library's shadow copy (C, no temporaries)

    def push_host_edits(self) -> None:
        """Upload mirrors the user edited in place since they were last pulled (one packed copy)."""
        mask = self._edited_mask()
        if mask:
            self._sim.sync_to_device(mask)
            self._sim.mirror_commit(mask)

    def step_host(self, nstep: int) -> None:
        """Host-driven step: edited fields up, ``nstep`` x mj_step (0 = mj_forward), whole state block back — ONE library call
        (edit detection and the shadow refresh included: ``mjb_step_host_auto``)."""
        self._sim.step_host_auto(int(nstep), compare=not self._state_stale)
        self._state_stale = False
        self._dev_newer = set(_DERIVED)
        self._check_engine_counters()

    def float64_twin(self, batch: int) -> BatchSim:
        """A float64 ``BatchSim`` of the same model / caps with ``batch`` environments, kept for reuse (the finite-difference
        fallback of ``linearization.py`` steps its perturbed replicas there)."""
        tw = self.__dict__.get("_twin")
        if tw is None or tw.batch != int(batch):
            s = self._sim
            tw = BatchSim(self.model._device_model(), int(batch), dtype="float64", nconmax=s.nconmax, nefcmax=s.nefcmax, device=s.device)
            object.__setattr__(self, "_twin", tw)
        return tw

    def mark_device_newer(self, eager: bool = False) -> None:
        self._state_stale = True
        self._dev_newer = set(_DERIVED)
        if eager:
            self.sync_host()

    def sync_host(self) -> None:
        self._pull("qpos")

    def _check_engine_counters(self) -> None:
        """Surface truncated physics once: contacts / constraint rows beyond the LDS caps, bad-state auto-resets (the device
        counters).  Cheap: one sticky flags word rides along with every mirror refresh; the [batch, 8] counters are
        only fetched once it is non-zero."""
        if int(self._flags[0]) == 0:
            return
        if int(self._flags[0]) & 8:
            raise TemplateError("the step kernel's work scheduler gave up waiting for a chunk of steps (engine flag 8): the results of "
                                "this launch are invalid; set MJB_CHUNK_STEPS=0 to use the static map and report the configuration")
        cn = self._sim.counters()
        msgs = []
        if int(cn["con_dropped"].sum()) or int(cn["efc_dropped"].sum()):
            msgs.append(("dropped", f"{int(cn['con_dropped'].sum())} contact(s) / {int(cn['efc_dropped'].sum())} constraint row(s) beyond the per-environment "
                         f"caps (nconmax={self._sim.nconmax}, nefcmax={self._sim.nefcmax}) were dropped: raise the caps at creation"))
        nbad = int(cn["warn_badqpos"].sum() + cn["warn_badqvel"].sum() + cn["warn_badqacc"].sum())
        if nbad:
            msgs.append(("badstate", f"{nbad} bad-state auto-reset(s) (NaN / >1e10 in qpos, qvel or qacc): those environments restarted from qpos0"))
        for key, text in msgs:
            if key not in self._warned:
                import warnings

                self._warned.add(key)
                self.engine_warnings.append(text)
                warnings.warn(text, RuntimeWarning, stacklevel=3)

    def __getattr__(self, name: str) -> Any:
        views = self.__dict__.get("_views")
        if views is not None and name in views:
            self._pull(name)
            return views[name]
        raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        if name in _STATE and "_views" in self.__dict__:
            self._pull(name)
            self._views[name][...] = value
            return
        object.__setattr__(self, name, value)

    @property
    def time(self):
        self._pull("time")
        t = self._mirror["time"]
        return float(t[0, 0]) if self.batch == 1 else t[:, 0]

    @time.setter
    def time(self, value) -> None:
        self._pull("time")
        self._mirror["time"][:, 0] = value

    # data.actuator_moment is CSR in MuJoCo >= 3.1 (moment_rownnz / moment_rowadr / moment_colind); the reference densifies it
    # with mju_sparse2dense (mujoco_template/setpoints.py:40-47).  The engine keeps it dense [nu, nv]; these views present
    # the dense rows as CSR rows of full length so that the reference's densification code runs unchanged.
    @property
    def moment_rownnz(self) -> np.ndarray:
        nu, nv = self.model.nu, self.model.nv
        r = np.full(nu, nv, dtype=np.int32)
        return r if self.batch == 1 else np.tile(r, (self.batch, 1))

    @property
    def moment_rowadr(self) -> np.ndarray:
        nu, nv = self.model.nu, self.model.nv
        r = (np.arange(nu, dtype=np.int32) * nv)
        return r if self.batch == 1 else np.tile(r, (self.batch, 1))

    @property
    def moment_colind(self) -> np.ndarray:
        nu, nv = self.model.nu, self.model.nv
        r = np.tile(np.arange(nv, dtype=np.int32), nu)
        return r if self.batch == 1 else np.tile(r, (self.batch, 1))

    def counters(self) -> dict[str, np.ndarray]:
        """Per-environment diagnostics: ncon, nefc, solver_niter, dropped contacts/rows, bad-state resets."""
        return self._sim.counters()


# ----------------------------------------------------------------------------------
# functions
# ----------------------------------------------------------------------------------

def _names(model: MjModel) -> CompiledModel:
    return model._c


def mj_name2id(model: MjModel, objtype: int, name: str) -> int:
    return model._device_model().name2id(int(objtype), name)               # C ABI: mjb_model_name2id


def mj_id2name(model: MjModel, objtype: int, idx: int) -> str | None:
    return model._device_model().id2name(int(objtype), int(idx))           # C ABI: mjb_model_id2name


def _check(model: MjModel, data: MjData) -> None:
    if data.model is not model:
        raise FatalError("data was created for a different model")


def mj_forward(model: MjModel, data: MjData) -> None:
    _check(model, data)
    data.step_host(0)


def mj_inverse(model: MjModel, data: MjData) -> None:
    """Inverse dynamics at the current (qpos, qvel, qacc): fills ``data.qfrc_inverse`` and ``data.actuator_moment``
    (reference call sites: ``mujoco_template/setpoints.py:29-31``, ``examples/humanoid/controllers/lqr.py:57-70``)."""
    _check(model, data)
    data.push_host_edits()
    data._sim.inverse()
    data._dev_newer |= {"qfrc_inverse", "actuator_moment"} | set(_DERIVED)


def mju_sparse2dense(res: np.ndarray, mat: np.ndarray, rownnz: np.ndarray, rowadr: np.ndarray, colind: np.ndarray) -> None:
    """CSR -> dense, in place on ``res`` [nr, nc] (the slice of ``mujoco.mju_sparse2dense`` the reference uses)."""
    res[...] = 0.0
    mat = np.asarray(mat).reshape(-1)
    colind = np.asarray(colind).reshape(-1)
    for r in range(res.shape[0]):
        a, n = int(rowadr[r]), int(rownnz[r])
        res[r, colind[a:a + n]] = mat[a:a + n]


def mj_step(model: MjModel, data: MjData, nstep: int = 1) -> None:
    _check(model, data)
    if int(nstep) < 1:
        raise ConfigError("mj_step: nstep must be >= 1")
    data.step_host(int(nstep))


def mj_saveModel(model: MjModel, filename: str, buffer=None) -> None:
    """``mujoco.mj_saveModel(m, filename, buffer)`` (reference model.py:49): the compiled table through ``mjb_model_save``."""
    if buffer is not None:
        raise ConfigError("mj_saveModel: saving into a caller buffer is not supported; pass a file name")
    model.save_binary(filename)


def mj_resetData(model: MjModel, data: MjData) -> None:
    _check(model, data)
    data._sim.reset(-1)
    data.mark_device_newer(eager=True)


def mj_resetDataKeyframe(model: MjModel, data: MjData, key: int) -> None:
    _check(model, data)
    data._sim.reset(int(key))
    data.mark_device_newer(eager=True)


def mj_subtreeCoM(model: MjModel, data: MjData) -> None:
    """subtree_com is produced by every forward pass; nothing further to compute."""
    _check(model, data)


def mjd_transitionFD(model: MjModel, data: MjData, eps: float, centered: bool, A, B, C=None, D=None, *unsupported) -> None:
    """Batched finite-difference transition matrices (float64 on device)."""
    if unsupported:
        raise TypeError("mjd_transitionFD takes 8 positional arguments")
    _check(model, data)
    data.push_host_edits()
    Ab, Bb = data._sim.transition_fd(float(eps), bool(centered), copy=False)       # pinned views: copied once, into the caller's arrays
    if A is not None:
        A[...] = Ab[0] if A.ndim == 2 else Ab
    if B is not None and model.nu > 0:
        B[...] = Bb[0] if B.ndim == 2 else Bb


def _jac(model: MjModel, data: MjData, kind: int, idx: int, jacp, jacr) -> None:
    _check(model, data)
    data.push_host_edits()
    jp, jr = data._sim.jac([kind], [int(idx)])
    if jacp is not None:
        jacp[...] = jp[0, 0] if jacp.ndim == 2 else jp[:, 0]
    if jacr is not None:
        jacr[...] = jr[0, 0] if jacr.ndim == 2 else jr[:, 0]


def mj_jacSite(model, data, jacp, jacr, site_id) -> None:
    _jac(model, data, 0, site_id, jacp, jacr)


def mj_jacBody(model, data, jacp, jacr, body_id) -> None:
    _jac(model, data, 1, body_id, jacp, jacr)


def mj_jacBodyCom(model, data, jacp, jacr, body_id) -> None:
    _jac(model, data, 2, body_id, jacp, jacr)


def mj_jacSubtreeCom(model, data, jacp, body_id) -> None:
    _jac(model, data, 3, body_id, jacp, None)


# position manifold helpers: C ABI (mjb_integrate_pos / mjb_differentiate_pos), in place on caller-owned float64 vectors,
# [n] or [batch, n] (controllers call them every step, reference examples/humanoid/controllers/lqr.py:153)

def mj_integratePos(model: MjModel, qpos: np.ndarray, qvel: np.ndarray, dt: float) -> None:
    model._device_model().integrate_pos(qpos, qvel, float(dt))


def mj_differentiatePos(model: MjModel, qvel: np.ndarray, dt: float, qpos1: np.ndarray, qpos2: np.ndarray) -> None:
    model._device_model().differentiate_pos(qvel, float(dt), qpos1, qpos2)


__all__ = [
    "MjModel", "MjData", "mjtObj", "mjtJoint", "FatalError", "mj_name2id", "mj_id2name", "mj_forward", "mj_step",
    "mj_inverse", "mju_sparse2dense", "mj_saveModel", "mj_resetData", "mj_resetDataKeyframe", "mj_subtreeCoM", "mjd_transitionFD", "mj_jacSite", "mj_jacBody",
    "mj_jacBodyCom", "mj_jacSubtreeCom", "mj_integratePos", "mj_differentiatePos",
]
