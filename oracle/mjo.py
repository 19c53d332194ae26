"""ctypes wrapper of the CPU oracle (``oracle/libmjo.so``).

TEST INFRASTRUCTURE ONLY — see the header of ``oracle/mjo.h``.  Importable from
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg;
the product package never imports this module.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libmjo.so")
    src = os.path.join(_HERE, "mjo.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "mjo.h"))):
        subprocess.check_call(["make", "-C", _HERE, "libmjo.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        # MJO_ORACLE_LIB: another build of the same source for this process (the instrumented flop-counting build libmjo_flops.so,
        # scripts/flop_count.py; the sanitizer build)
        L = ctypes.CDLL(os.environ.get("MJO_ORACLE_LIB") or build())
        vp, ci, cd, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_uint
        pd = ctypes.POINTER(ctypes.c_double)
        L.mjo_model_create.restype = vp
        L.mjo_model_create.argtypes = [ci, vp, vp, vp, vp]
        L.mjo_model_free.argtypes = [vp]
        L.mjo_last_error.restype = ctypes.c_char_p
        L.mjo_set_disableactuator.argtypes = [vp, ci]
        L.mjo_set_limits.argtypes = [vp, ci, ci]
        L.mjo_set_solver.argtypes = [vp, ci, cd]
        L.mjo_set_round_mask.argtypes = [vp, ci]
        L.mjo_data_create.restype = vp
        L.mjo_data_create.argtypes = [vp]
        L.mjo_data_free.argtypes = [vp]
        L.mjo_data_array.restype = pd
        L.mjo_data_array.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
        L.mjo_data_iarray.restype = ctypes.POINTER(ctypes.c_int)
        L.mjo_data_iarray.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
        L.mjo_get_contacts.restype = ctypes.c_long
        L.mjo_get_contacts.argtypes = [vp, vp, ctypes.c_long]
        L.mjo_get_time.restype = cd
        L.mjo_get_time.argtypes = [vp]
        L.mjo_set_time.argtypes = [vp, cd]
        L.mjo_reset.argtypes = [vp, vp]
        L.mjo_reset_keyframe.restype = ci
        L.mjo_reset_keyframe.argtypes = [vp, vp, ci]
        L.mjo_forward.argtypes = [vp, vp]
        L.mjo_inverse.argtypes = [vp, vp]
        L.mjo_step.argtypes = [vp, vp]
        L.mjo_random_ctrl.argtypes = [vp, vp, cu, cu, cu, cd]
        L.mjo_rollout_random.argtypes = [vp, vp, ci, cu, cu, cu, cd]
        L.mjo_rollout_batch.restype = ctypes.c_long
        L.mjo_rollout_batch.argtypes = [vp, ci, ci, cu, cu, cd, ci, vp, vp, vp, vp]
        L.mjo_transition_fd.argtypes = [vp, vp, cd, ci, vp, vp]
        L.mjo_jac.argtypes = [vp, vp, ci, ci, vp, vp]
        L.mjo_integrate_pos.argtypes = [vp, vp, vp, cd]
        L.mjo_differentiate_pos.argtypes = [vp, vp, cd, vp, vp]
        _LIB = L
    return _LIB


class OracleModel:
    def __init__(self, compiled):
        from mujoco_template_amd._pack import PackedTable  # schema shared with the product's C ABI

        self.compiled = compiled
        self._packed = PackedTable(compiled)
        p = self._packed
        self.ptr = lib().mjo_model_create(p.n, ctypes.cast(p.names, ctypes.c_void_p), ctypes.cast(p.ptrs, ctypes.c_void_p),
                                          ctypes.cast(p.dtypes, ctypes.c_void_p), ctypes.cast(p.counts, ctypes.c_void_p))
        if not self.ptr:
            raise RuntimeError(lib().mjo_last_error().decode())
        for k in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "ntendon", "nsensordata", "nkey"):
            setattr(self, k, int(getattr(compiled, k)))

    def set_disableactuator(self, mask: int) -> None:
        lib().mjo_set_disableactuator(self.ptr, int(mask))

    def set_limits(self, nconmax: int, nefcmax: int) -> None:
        lib().mjo_set_limits(self.ptr, int(nconmax), int(nefcmax))

    def set_solver(self, iterations: int, tolerance: float) -> None:
        lib().mjo_set_solver(self.ptr, int(iterations), float(tolerance))

    def set_round_mask(self, mask: int) -> None:
        """Precision study only: round the outputs of the phases in ``mask`` to fp32 (oracle/mjo.c RM_*); 0 = the oracle proper."""
        lib().mjo_set_round_mask(self.ptr, int(mask))

    def __del__(self):
        try:
            if self.ptr:
                lib().mjo_model_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def rollout_batch(model: "OracleModel", nenv: int, nstep: int, seed: int = 0, env0: int = 0, scale: float = 1.0, nthreads: int = 0,
                  qpos_init=None, qvel_init=None):
    """nenv independent random-ctrl rollouts (OpenMP, one env per task). Returns (qpos [nenv,nq], qvel [nenv,nv])."""
    qo = np.zeros((nenv, model.nq)); vo = np.zeros((nenv, model.nv))
    qi = None if qpos_init is None else np.ascontiguousarray(qpos_init, dtype=np.float64)
    vi = None if qvel_init is None else np.ascontiguousarray(qvel_init, dtype=np.float64)
    lib().mjo_rollout_batch(model.ptr, int(nenv), int(nstep), seed, env0, float(scale), int(nthreads),
                            None if qi is None else qi.ctypes.data, None if vi is None else vi.ctypes.data, qo.ctypes.data, vo.ctypes.data)
    return qo, vo


class OracleData:
    """One float64 environment.  Array attributes are live numpy views."""

    _FIELDS = ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "qacc_smooth", "qfrc_applied", "qfrc_bias",
               "qfrc_passive", "qfrc_actuator", "qfrc_smooth", "qfrc_constraint", "qfrc_inverse", "xpos", "xquat", "xmat", "xipos",
               "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat", "site_xpos", "site_xmat", "subtree_com",
               "cinert", "crb", "cdof", "cdof_dot", "cvel", "cacc", "cfrc", "qM", "qL", "ten_length", "ten_J",
               "actuator_length", "actuator_velocity", "actuator_force", "actuator_moment", "sensordata")
    _EFC = ("efc_J", "efc_pos", "efc_D", "efc_R", "efc_aref", "efc_vel", "efc_force", "efc_diagApprox")

    def __init__(self, model: OracleModel):
        self.model = model
        self.ptr = lib().mjo_data_create(model.ptr)
        self._views: dict[str, np.ndarray] = {}
        for f in self._FIELDS:
            self._views[f] = self._array(f)

    def _array(self, name: str) -> np.ndarray:
        cnt = ctypes.c_long(0)
        p = lib().mjo_data_array(self.ptr, name.encode(), ctypes.byref(cnt))
        if cnt.value < 0:
            raise KeyError(name)
        if cnt.value == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(cnt.value,))

    def __getattr__(self, name: str):
        views = self.__dict__.get("_views", {})
        if name in views:
            return views[name]
        if name in OracleData._EFC:
            return np.array(self._array(name))
        raise AttributeError(name)

    @property
    def time(self) -> float:
        return float(lib().mjo_get_time(self.ptr))

    @time.setter
    def time(self, t: float) -> None:
        lib().mjo_set_time(self.ptr, float(t))

    def counters(self) -> dict[str, int]:
        cnt = ctypes.c_long(0)
        p = lib().mjo_data_iarray(self.ptr, b"counters", ctypes.byref(cnt))
        v = [p[i] for i in range(8)]
        return dict(ncon=v[0], nefc=v[1], solver_niter=v[2], ncon_dropped=v[3], nefc_dropped=v[4],
                    warn_badqpos=v[5], warn_badqvel=v[6], warn_badqacc=v[7])

    def efc_type(self) -> np.ndarray:
        cnt = ctypes.c_long(0)
        p = lib().mjo_data_iarray(self.ptr, b"efc_type", ctypes.byref(cnt))
        return np.array([p[i] for i in range(cnt.value)], dtype=np.int32)

    def contacts(self) -> dict[str, np.ndarray]:
        n = self.counters()["ncon"]
        buf = np.zeros((max(n, 1), 15))
        lib().mjo_get_contacts(self.ptr, buf.ctypes.data, n)
        buf = buf[:n]
        return dict(dist=buf[:, 0], pos=buf[:, 1:4], frame=buf[:, 4:13].reshape(-1, 3, 3),
                    geom1=buf[:, 13].astype(int), geom2=buf[:, 14].astype(int))

    def reset(self) -> None:
        lib().mjo_reset(self.model.ptr, self.ptr)

    def reset_keyframe(self, key: int) -> None:
        if lib().mjo_reset_keyframe(self.model.ptr, self.ptr, int(key)) != 0:
            raise IndexError(key)

    def forward(self) -> None:
        lib().mjo_forward(self.model.ptr, self.ptr)

    def inverse(self) -> None:
        """mj_inverse: qfrc_inverse from the current (qpos, qvel, qacc)."""
        lib().mjo_inverse(self.model.ptr, self.ptr)

    def step(self, n: int = 1) -> None:
        for _ in range(n):
            lib().mjo_step(self.model.ptr, self.ptr)

    def random_ctrl(self, seed: int, env: int, step: int, scale: float = 1.0) -> np.ndarray:
        out = np.zeros(self.model.nu)
        lib().mjo_random_ctrl(self.model.ptr, out.ctypes.data, seed, env, step, scale)
        return out

    def rollout_random(self, nstep: int, seed: int, env: int, step0: int = 0, scale: float = 1.0) -> None:
        lib().mjo_rollout_random(self.model.ptr, self.ptr, int(nstep), seed, env, step0, float(scale))

    def transition_fd(self, eps: float = 1e-6, centered: bool = True) -> tuple[np.ndarray, np.ndarray]:
        nv, nu = self.model.nv, self.model.nu
        A = np.zeros((2 * nv, 2 * nv))
        B = np.zeros((2 * nv, max(nu, 1)))
        lib().mjo_transition_fd(self.model.ptr, self.ptr, float(eps), int(bool(centered)), A.ctypes.data, B.ctypes.data)
        return A, B[:, :nu]

    def jac(self, kind: int, idx: int) -> tuple[np.ndarray, np.ndarray]:
        nv = self.model.nv
        jp = np.zeros((3, nv))
        jr = np.zeros((3, nv))
        lib().mjo_jac(self.model.ptr, self.ptr, int(kind), int(idx), jp.ctypes.data, jr.ctypes.data)
        return jp, jr

    def integrate_pos(self, qpos: np.ndarray, qvel: np.ndarray, dt: float) -> np.ndarray:
        q = np.array(qpos, dtype=np.float64)
        v = np.ascontiguousarray(qvel, dtype=np.float64)
        lib().mjo_integrate_pos(self.model.ptr, q.ctypes.data, v.ctypes.data, float(dt))
        return q

    def differentiate_pos(self, qpos1: np.ndarray, qpos2: np.ndarray, dt: float = 1.0) -> np.ndarray:
        out = np.zeros(self.model.nv)
        q1 = np.ascontiguousarray(qpos1, dtype=np.float64)
        q2 = np.ascontiguousarray(qpos2, dtype=np.float64)
        lib().mjo_differentiate_pos(self.model.ptr, out.ctypes.data, float(dt), q1.ctypes.data, q2.ctypes.data)
        return out

    def __del__(self):
        try:
            if self.ptr:
                lib().mjo_data_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def load(xml_path: str) -> tuple[OracleModel, OracleData]:
    from mujoco_template_amd.mjcf import compile_xml_path

    m = OracleModel(compile_xml_path(xml_path))
    return m, OracleData(m)


__all__ = ["OracleModel", "OracleData", "load", "build", "lib", "rollout_batch"]
