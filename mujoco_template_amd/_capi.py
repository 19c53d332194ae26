"""ctypes binding of ``libmjbatch.so`` (C ABI: ``include/mjbatch.h``) and the thin
:class:`BatchSim` object the Python front (``model.py`` / ``env.py``) dispatches through.

There is no CPU fallback: if the HIP library is missing or no device is
visible, construction raises :class:`~mujoco_template_amd.exceptions.TemplateError`.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Sequence

import numpy as np

from ._pack import PackedTable
from .exceptions import ConfigError, NameLookupError, TemplateError, raise_for_status
from .mjcf import CompiledModel

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MJB_LIB") or os.path.join(_HERE, "libmjbatch_prof.so" if os.environ.get("MJB_PROFILE") == "1" else "libmjbatch.so")
_LIB: ctypes.CDLL | None = None

MJB_F32, MJB_F64 = 0, 1
CTRL_KEEP, CTRL_ZERO, CTRL_RANDOM, CTRL_FEEDBACK = 0, 1, 2, 3
COUNTER_NAMES = ("ncon", "nefc", "solver_niter", "con_dropped", "efc_dropped", "warn_badqpos", "warn_badqvel", "warn_badqacc")


def build_library(force: bool = False) -> str:
    """Compile the HIP sources in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j6"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise TemplateError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the batched engine has no CPU fallback)")
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): import it first so both share one HIP runtime
    import torch  # noqa: F401

    L = ctypes.CDLL(_LIB_PATH)
    vp, ci, cd, cu, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_uint, ctypes.c_long
    pvp = ctypes.POINTER(ctypes.c_void_p)
    pci = ctypes.POINTER(ctypes.c_int)
    L.mjb_last_error.restype = ctypes.c_char_p
    L.mjb_device_count.restype = ci
    L.mjb_model_create.argtypes = [ci, vp, vp, vp, vp, pvp]
    L.mjb_model_free.argtypes = [vp]
    L.mjb_model_set_disableactuator.argtypes = [vp, ci]
    L.mjb_model_set_solver.argtypes = [vp, ci, cd]
    L.mjb_data_create.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, pvp]
    L.mjb_data_free.argtypes = [vp]
    L.mjb_set_stream.argtypes = [vp, vp]
    L.mjb_sync.argtypes = [vp]
    L.mjb_engine_flags.argtypes = [vp, pci]
    L.mjb_data_info.argtypes = [vp, pci, pci, pci, pci, pci, pci]
    L.mjb_array_ptr.argtypes = [vp, ctypes.c_char_p, pvp, ctypes.POINTER(cl), pci]
    L.mjb_get_array.argtypes = [vp, ctypes.c_char_p, vp]
    L.mjb_set_array.argtypes = [vp, ctypes.c_char_p, vp]
    L.mjb_get_counters.argtypes = [vp, vp]
    L.mjb_reset.argtypes = [vp, ci]
    L.mjb_forward.argtypes = [vp]
    L.mjb_model_spec_source.argtypes = [vp, ci, ci, ci, ci, ctypes.c_char_p, cl]
    L.mjb_model_spec_source.restype = cl
    L.mjb_spec_source.argtypes = [vp, ctypes.c_char_p, cl]
    L.mjb_spec_source.restype = cl
    L.mjb_spec_load.argtypes = [vp, ctypes.c_char_p, cl]
    L.mjb_spec_unload.argtypes = [vp]
    L.mjb_model_step2_spec_source.argtypes = [vp, ci, ci, ci, ctypes.c_char_p, cl]
    L.mjb_model_step2_spec_source.restype = cl
    L.mjb_step2_spec_source.argtypes = [vp, ctypes.c_char_p, cl]
    L.mjb_step2_spec_source.restype = cl
    L.mjb_step2_spec_load.argtypes = [vp, ctypes.c_char_p, cl]
    L.mjb_step2_spec_unload.argtypes = [vp]
    L.mjb_model_fd_spec_source.argtypes = [vp, ci, ci, ci, ci, ctypes.c_char_p, cl]
    L.mjb_model_fd_spec_source.restype = cl
    L.mjb_fd_spec_source.argtypes = [vp, ctypes.c_char_p, cl]
    L.mjb_fd_spec_source.restype = cl
    L.mjb_fd_spec_load.argtypes = [vp, ctypes.c_char_p, cl]
    L.mjb_fd_spec_unload.argtypes = [vp]
    L.mjb_inverse.argtypes = [vp]
    L.mjb_step.argtypes = [vp, ci]
    L.mjb_rollout.argtypes = [vp, ci, ci, cu, cu, cd, vp, vp, ci]
    L.mjb_set_feedback.argtypes = [vp, vp, vp, vp, vp]
    L.mjb_set_feedback.restype = ci
    L.mjb_set_feedback_noise.argtypes = [vp, vp, vp, ci, ci]
    L.mjb_set_feedback_noise.restype = ci
    L.mjb_feedback_ctrl.argtypes = [vp, ci]
    L.mjb_feedback_ctrl.restype = ci
    L.mjb_obs_spec_create.argtypes = [vp, ci, ci, vp, ci, vp, ci, vp, ci, vp, pvp]
    L.mjb_obs_spec_free.argtypes = [vp]
    L.mjb_obs_dim.argtypes = [vp]
    L.mjb_obs_gather.argtypes = [vp, vp, vp]
    L.mjb_transition_fd.argtypes = [vp, cd, ci, vp, vp]
    L.mjb_transition_fd_pinned.argtypes = [vp, cd, ci, ctypes.POINTER(ctypes.POINTER(cd)), ctypes.POINTER(ctypes.POINTER(cd))]
    L.mjb_transition_fd_pinned.restype = ci
    L.mjb_jac.argtypes = [vp, ci, vp, vp, vp, vp]
    L.mjb_profile_get.argtypes = [vp, vp]
    L.mjb_profile_get.restype = ci
    L.mjb_profile_env_get.argtypes = [vp, vp]
    L.mjb_step_schedule.argtypes = [vp, vp]
    L.mjb_step_schedule.restype = ci
    L.mjb_profile_env_get.restype = ci
    L.mjb_debug_forward.argtypes = [vp]
    L.mjb_debug_get.argtypes = [vp, ctypes.c_char_p, vp, cl]
    pcl = ctypes.POINTER(cl)
    L.mjb_model_name2id.argtypes = [vp, ci, ctypes.c_char_p]
    L.mjb_model_name2id.restype = ci
    L.mjb_model_id2name.argtypes = [vp, ci, ci]
    L.mjb_model_id2name.restype = ctypes.c_char_p
    L.mjb_model_field.argtypes = [vp, ctypes.c_char_p, pvp, pcl, pci]
    L.mjb_model_field_at.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_char_p), pvp, pcl, pci]
    L.mjb_model_load_xml.argtypes = [ctypes.c_char_p, pvp]
    L.mjb_model_load_xml_string.argtypes = [ctypes.c_char_p, ctypes.c_char_p, pvp]
    L.mjb_model_save.argtypes = [vp, ctypes.c_char_p]
    L.mjb_model_load.argtypes = [ctypes.c_char_p, pvp]
    L.mjb_integrate_pos.argtypes = [vp, ci, vp, vp, cd]
    L.mjb_differentiate_pos.argtypes = [vp, ci, vp, cd, vp, vp]
    L.mjb_host_view.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.POINTER(cd)), pcl]
    L.mjb_sync_to_host.argtypes = [vp]
    L.mjb_sync_to_device.argtypes = [vp, ci]
    L.mjb_step_host.argtypes = [vp, ci, ci]
    L.mjb_mirror_edited_mask.argtypes = [vp, ctypes.POINTER(ci)]
    L.mjb_mirror_commit.argtypes = [vp, ci]
    L.mjb_step_host_auto.argtypes = [vp, ci, ci, ctypes.POINTER(ci)]
    for name in ("mjb_model_create", "mjb_model_set_disableactuator", "mjb_model_set_solver", "mjb_data_create", "mjb_set_stream",
                 "mjb_sync", "mjb_data_info", "mjb_array_ptr", "mjb_get_array", "mjb_set_array", "mjb_get_counters", "mjb_reset",
                 "mjb_forward", "mjb_inverse", "mjb_spec_load", "mjb_spec_unload", "mjb_fd_spec_load", "mjb_fd_spec_unload", "mjb_step2_spec_load", "mjb_step2_spec_unload", "mjb_step", "mjb_rollout", "mjb_obs_spec_create", "mjb_obs_dim", "mjb_obs_gather",
                 "mjb_transition_fd", "mjb_jac", "mjb_debug_forward", "mjb_debug_get", "mjb_model_field", "mjb_model_field_at", "mjb_model_save",
                 "mjb_model_load", "mjb_model_load_xml", "mjb_model_load_xml_string", "mjb_integrate_pos", "mjb_differentiate_pos", "mjb_host_view", "mjb_sync_to_host", "mjb_sync_to_device",
                 "mjb_step_host", "mjb_mirror_edited_mask", "mjb_mirror_commit", "mjb_step_host_auto", "mjb_engine_flags"):
        getattr(L, name).restype = ci
    _LIB = L
    return L


def _check(rc: int) -> None:
    """Turn a C status code into the exception of ``exceptions.raise_for_status``."""
    if rc != 0:
        raise_for_status(rc, load_library().mjb_last_error().decode())


class _CudaArray:
    """``__cuda_array_interface__`` shim so torch can wrap library-owned device memory zero-copy."""

    def __init__(self, ptr: int, shape: tuple[int, ...], typestr: str, owner: object):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


def _manifold_batch(arr: np.ndarray, n: int, what: str) -> int:
    """Batch size of a caller-owned [n] / [batch, n] float64 vector the C side edits in place."""
    if not isinstance(arr, np.ndarray) or arr.dtype != np.float64 or not arr.flags.c_contiguous or not arr.flags.writeable:
        raise ConfigError(f"{what} must be a writeable C-contiguous float64 numpy array (it is modified in place)")
    if n == 0 or arr.size % n or arr.ndim not in (1, 2) or arr.shape[-1] != n:
        raise ConfigError(f"{what} must have shape [{n}] or [batch, {n}]")
    return arr.size // n


def _ids(seq: Sequence[int]):
    arr = np.ascontiguousarray(np.asarray(list(seq), dtype=np.int32))
    return arr, (arr.ctypes.data if arr.size else None)



# ----------------------------------------------------------------------------------
# per-model specialisation of the fp32 step kernel (include/mjbatch.h: mjb_*spec*)
# ----------------------------------------------------------------------------------
_WARNED_NO_SPEC = False
_CSRC = os.path.join(_HERE, "csrc")
_JIT_DIR = os.path.join(_HERE, "_jit")


def _source_from(fn, *args) -> str:
    n = fn(*args, None, 0)
    if n < 0:
        _check(-1)
    buf = ctypes.create_string_buffer(int(n) + 1)
    fn(*args, buf, int(n) + 1)
    return buf.value.decode()


def spec_scheduler(source: str) -> str | None:
    """The machine scheduler the specialised kernel is compiled with — a stated rule, not a retry: the iterative ILP strategy
    (measured on the same box, default scheduler -> iterative: humanoid +4 %, drone2 +8 %, cart-pole +5 %) for every model that can
    have constraints worth scheduling for (row cap ``nefc_max >= 8``, or one wave per environment); LLVM's default for the
    degenerate ones (pendulum: 2 rows, the tests' BASE_XML: 2 rows), whose kernels are tiny and where the one compiler crash
    with this flag was seen (ROCm 7.2.0 clang, greedy register allocator, ``profiles/r02_hipcc_iterative_ilp_crash.txt``)."""
    import re

    forced = os.environ.get("MJB_SPEC_SCHED")                  # experiments: "iterative-ilp", "default", ...
    if forced is not None:
        return None if forced in ("", "default") else forced
    g = re.search(r"#define MJB_SPEC_G (\d+)", source)
    ne = re.search(r"nefc_max == (\d+)", source)
    lanes, rows = (int(g.group(1)) if g else 0), (int(ne.group(1)) if ne else 0)
    return "iterative-ilp" if lanes == 64 or rows >= 8 else None


def _private_cache_dir() -> str:
    """Per-user cache for read-only installs: created 0700, must be a real directory owned by this user with no group / other
    write bit (another local user must not be able to plant a code object that ``hipModuleLoadData`` would then run)."""
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    path = os.path.join(base, "mujoco_template_amd", "jit")
    try:
        os.makedirs(path, mode=0o700, exist_ok=True)
        st = os.lstat(path)
    except OSError as exc:
        raise TemplateError(f"no writable cache directory for the specialised kernel: {exc}") from exc
    import stat

    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise TemplateError(f"refusing the kernel cache {path}: it must be a directory owned by uid {os.getuid()} without group/other write access")
    return path


def _read_private(path: str) -> bytes:
    """Read a cached code object without following a symlink at the final component.  Objects in the in-tree cache carry the
    trust of the package itself (they ship beside ``libmjbatch.so``); anything else must be owned by this user and not writable
    by others."""
    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    try:
        st = os.fstat(fd)
        in_tree = os.path.dirname(os.path.abspath(path)) == os.path.abspath(_JIT_DIR)
        if not in_tree and (st.st_uid != os.getuid() or (st.st_mode & 0o022)):
            raise TemplateError(f"refusing the cached kernel {path}: not owned by uid {os.getuid()} or writable by others")
        with os.fdopen(fd, "rb", closefd=False) as fh:
            return fh.read()
    finally:
        os.close(fd)


def compile_spec(source: str, *, force: bool = False) -> str:
    """Compile a specialised translation unit (``mjb_*spec_source``) to a gfx950 code object; returns its path.

    Cached in-tree (``mujoco_template_amd/_jit/``, keyed by the source text, the kernel headers and the scheduler rule), so
    the objects built by ``__graft_entry__.build()`` on a GPU-less machine travel with the tree; a read-only install uses a
    private per-user cache (0700, ownership checked).  hipcc cross-compiles without a GPU.
    """
    import hashlib
    import shutil
    import subprocess

    extra = os.environ.get("MJB_SPEC_FLAGS", "").split()        # experiments, e.g. -DMJB_WPS=3 (register budget for 3 waves/SIMD)
    sched = spec_scheduler(source)
    h = hashlib.sha1((source + " ".join(extra) + f" sched={sched} nolicm").encode())
    for f in ("mjb_types.hpp", "mjb_device.hpp", "mjb_kernels.hpp"):
        with open(os.path.join(_CSRC, f), "rb") as fh:
            h.update(fh.read())
    key = h.hexdigest()[:20]
    jit_dir = _JIT_DIR
    try:
        os.makedirs(jit_dir, exist_ok=True)
        if not os.access(jit_dir, os.W_OK):
            raise OSError("not writable")
    except OSError:                                            # read-only install
        jit_dir = _private_cache_dir()
    out = os.path.join(jit_dir, f"k_step_spec_{key}.hsaco")
    if os.path.exists(out) and not force:
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise TemplateError("hipcc not found: cannot specialise the step kernel (the generic kernel remains available)")
    src = os.path.join(jit_dir, f"k_step_spec_{key}.hip")
    tmp = out + f".tmp{os.getpid()}"
    try:
        fd = os.open(src, os.O_WRONLY | os.O_CREAT | os.O_TRUNC | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "w") as fh:
            fh.write(source)
    except OSError as exc:
        raise TemplateError(f"cannot write the specialised kernel source: {exc}") from exc
    # same floating-point flags as csrc/Makefile (contraction per source expression, correctly rounded division / sqrt): bitwise equal to the generic kernel
    # -disable-machine-licm: the step kernel is one persistent loop around the whole forward pipeline; hoisting the body's literal
    # constants out of it costs more registers than the kernel has (csrc/Makefile, STEPFLAGS)
    base = [hipcc, "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-ffp-contract=on", "-mllvm", "-disable-machine-licm"]
    tail = [*extra, "-I", _CSRC, "-o", tmp, src]
    if "#define MJB_SPEC_KERNEL 2" in source:                   # the finite-difference kernel has no persistent loop around its body
        base = [x for x in base if x not in ("-mllvm", "-disable-machine-licm")]
    attempts = [[*base, "-mllvm", f"-amdgpu-sched-strategy={sched}", *tail], [*base, *tail]] if sched else [[*base, *tail]]
    err = ""
    for k, cmd in enumerate(attempts):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        except (OSError, subprocess.SubprocessError) as exc:
            raise TemplateError(f"hipcc could not be run: {exc}") from exc
        if r.returncode == 0 and os.path.exists(tmp):
            if k > 0:                                          # the rule's scheduler failed: say so, with the evidence
                import warnings

                warnings.warn(f"hipcc failed with -amdgpu-sched-strategy={sched} on this specialised kernel; compiled with the default "
                              f"scheduler instead (same arithmetic).  stderr tail:\n{err[-1200:]}", RuntimeWarning, stacklevel=2)
            break
        err = r.stderr
        if os.path.exists(tmp):
            os.remove(tmp)
    else:
        raise TemplateError("specialised kernel failed to compile:\n" + err[-2000:])
    os.replace(tmp, out)
    return out


class ObsSpecHandle:
    def __init__(self, sim: "BatchSim", flags: int, site_ids, body_ids, geom_ids, subtree_ids):
        L = load_library()
        self.sim = sim
        self.ptr = ctypes.c_void_p()
        s, sp = _ids(site_ids); b, bp = _ids(body_ids); g, gp = _ids(geom_ids); t, tp = _ids(subtree_ids)
        _check(L.mjb_obs_spec_create(sim.ptr, flags, s.size, sp, b.size, bp, g.size, gp, t.size, tp, ctypes.byref(self.ptr)))
        self.dim = int(L.mjb_obs_dim(self.ptr))

    def __del__(self):
        try:
            if self.ptr:
                load_library().mjb_obs_spec_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


MIRROR_FIELDS = ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "time")     # bit order of the field masks of mjb_sync_to_device / mjb_step_host


class DeviceModel:
    """Handle of ``mjbModel`` (host copy of the compiled model inside the library)."""

    def __init__(self, compiled: CompiledModel | None, *, _ptr: ctypes.c_void_p | None = None):
        L = load_library()
        if _ptr is not None:                                   # adopted handle (DeviceModel.load)
            self.ptr, self._packed = _ptr, None
            self.compiled = self._compiled_from_library()
            return
        self.compiled = compiled
        self._packed = PackedTable(compiled)
        p = self._packed
        self.ptr = ctypes.c_void_p()
        _check(L.mjb_model_create(p.n, ctypes.cast(p.names, ctypes.c_void_p), ctypes.cast(p.ptrs, ctypes.c_void_p),
                                  ctypes.cast(p.dtypes, ctypes.c_void_p), ctypes.cast(p.counts, ctypes.c_void_p), ctypes.byref(self.ptr)))

    # -- MjModel.from_binary_path / mj_saveModel (reference model.py:28-31,:49) -----------------------
    @classmethod
    def load(cls, path: str) -> "DeviceModel":
        ptr = ctypes.c_void_p()
        _check(load_library().mjb_model_load(os.fsencode(path), ctypes.byref(ptr)))
        return cls(None, _ptr=ptr)

    def save(self, path: str) -> None:
        _check(load_library().mjb_model_save(self.ptr, os.fsencode(path)))

    # -- MjModel.from_xml_path / from_xml_string (reference model.py:22-27): the native MJCF compiler -----
    @classmethod
    def load_xml(cls, path: str) -> "DeviceModel":
        ptr = ctypes.c_void_p()
        _check(load_library().mjb_model_load_xml(os.fsencode(os.path.abspath(path)), ctypes.byref(ptr)))
        return cls(None, _ptr=ptr)

    @classmethod
    def load_xml_string(cls, xml_text: str, base_dir: str = ".") -> "DeviceModel":
        ptr = ctypes.c_void_p()
        _check(load_library().mjb_model_load_xml_string(xml_text.encode(), os.fsencode(os.path.abspath(base_dir)), ctypes.byref(ptr)))
        return cls(None, _ptr=ptr)

    def fields(self) -> dict[str, np.ndarray]:
        """Every field of the model table as a numpy copy (``mjb_model_field_at``)."""
        L = load_library()
        out: dict[str, np.ndarray] = {}
        name, ptr, cnt, dt = ctypes.c_char_p(), ctypes.c_void_p(), ctypes.c_long(), ctypes.c_int()
        i = 0
        while L.mjb_model_field_at(self.ptr, i, ctypes.byref(name), ctypes.byref(ptr), ctypes.byref(cnt), ctypes.byref(dt)) == 0:
            ctype, npdt = {0: (ctypes.c_double, np.float64), 1: (ctypes.c_int, np.int32), 2: (ctypes.c_ubyte, np.uint8)}[dt.value]
            n = int(cnt.value)
            out[name.value.decode()] = (np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape=(n,)).astype(npdt, copy=True)
                                        if n else np.zeros(0, dtype=npdt))
            i += 1
        return out

    def _compiled_from_library(self) -> CompiledModel:
        from ._pack import compiled_from_fields

        return compiled_from_fields(self.fields())

    # -- names (mj_name2id / mj_id2name) ----------------------------------------------------------------
    def name2id(self, objtype: int, name: str) -> int:
        return int(load_library().mjb_model_name2id(self.ptr, int(objtype), str(name).encode()))

    def id2name(self, objtype: int, idx: int) -> str | None:
        r = load_library().mjb_model_id2name(self.ptr, int(objtype), int(idx))
        return r.decode() if r is not None else None

    # -- position manifold (mj_integratePos / mj_differentiatePos), host float64, batched -------------
    def integrate_pos(self, qpos: np.ndarray, qvel: np.ndarray, dt: float) -> None:
        """In place on ``qpos`` ([nq] or [batch, nq], float64, C-contiguous)."""
        batch = _manifold_batch(qpos, self.compiled.nq, "qpos")
        v = np.ascontiguousarray(qvel, dtype=np.float64)
        if v.size != batch * self.compiled.nv:
            raise ConfigError("mj_integratePos: qvel must have nv entries per environment")
        _check(load_library().mjb_integrate_pos(self.ptr, batch, qpos.ctypes.data, v.ctypes.data, float(dt)))

    def differentiate_pos(self, qvel: np.ndarray, dt: float, qpos1: np.ndarray, qpos2: np.ndarray) -> None:
        """``qvel <- (qpos2 (-) qpos1) / dt`` in place on ``qvel`` ([nv] or [batch, nv], float64, C-contiguous)."""
        batch = _manifold_batch(qvel, self.compiled.nv, "qvel")
        q1 = np.ascontiguousarray(qpos1, dtype=np.float64)
        q2 = np.ascontiguousarray(qpos2, dtype=np.float64)
        if q1.size != batch * self.compiled.nq or q2.size != q1.size:
            raise ConfigError("mj_differentiatePos: qpos1 / qpos2 must have nq entries per environment")
        _check(load_library().mjb_differentiate_pos(self.ptr, batch, qvel.ctypes.data, float(dt), q1.ctypes.data, q2.ctypes.data))

    def set_disableactuator(self, mask: int) -> None:
        _check(load_library().mjb_model_set_disableactuator(self.ptr, int(mask)))

    def set_solver(self, iterations: int, tolerance: float) -> None:
        _check(load_library().mjb_model_set_solver(self.ptr, int(iterations), float(tolerance)))

    def spec_source(self, *, lanes: int = 0, nconmax: int = 0, nefcmax: int = 0) -> str:
        """Translation unit of the specialised fp32 step kernel for these creation arguments (no GPU needed)."""
        return _source_from(load_library().mjb_model_spec_source, self.ptr, MJB_F32, int(lanes), int(nconmax), int(nefcmax))

    def step2_spec_source(self, *, lanes: int = 0, nconmax: int = 0, nefcmax: int = 0) -> str | None:
        """Translation unit of the specialised two-wave step kernel (small batches), or None when that kernel does not apply to the model."""
        try:
            return _source_from(load_library().mjb_model_step2_spec_source, self.ptr, int(lanes), int(nconmax), int(nefcmax))
        except TemplateError:
            return None

    def fd_spec_source(self, *, dtype: str = "float32", lanes: int = 0, nconmax: int = 0, nefcmax: int = 0) -> str:
        """Translation unit of the specialised float64 finite-difference kernel for these creation arguments (no GPU needed)."""
        return _source_from(load_library().mjb_model_fd_spec_source, self.ptr, MJB_F32 if dtype == "float32" else MJB_F64, int(lanes), int(nconmax), int(nefcmax))

    def __del__(self):
        try:
            if self.ptr:
                load_library().mjb_model_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class BatchSim:
    """``batch`` replicas of one model resident on one GPU (handle of ``mjbData``)."""

    def __init__(self, model: DeviceModel, batch: int, *, dtype: str = "float32", lanes: int = 0, nconmax: int = 0,
                 nefcmax: int = 0, device: int = 0, env0: int = 0, specialize: bool | None = None):
        L = load_library()
        if dtype not in ("float32", "float64"):
            raise ConfigError("dtype must be 'float32' or 'float64'")
        self.model = model
        self.batch = int(batch)
        self.dtype = dtype
        self.np_dtype = np.float32 if dtype == "float32" else np.float64
        self.device = int(device)
        self.ptr = ctypes.c_void_p()
        _check(L.mjb_data_create(model.ptr, self.batch, MJB_F32 if dtype == "float32" else MJB_F64, int(lanes), int(nconmax),
                                 int(nefcmax), self.device, int(env0), ctypes.byref(self.ptr)))
        info = [ctypes.c_int() for _ in range(6)]
        _check(L.mjb_data_info(self.ptr, *[ctypes.byref(x) for x in info]))
        self.lanes, self.nconmax, self.nefcmax, self.lds_bytes_per_env = info[2].value, info[3].value, info[4].value, info[5].value
        # per-model specialised fp32 kernel: on by default (MJB_SPECIALIZE=0 turns the default off); an explicit True raises
        # if it cannot be built, the default falls back to the generic kernel with one warning
        self.specialized = False
        self.fd_specialized = False
        self._fd_spec_tried = False
        self._spec_policy = specialize                             # None: default (on unless MJB_SPECIALIZE=0); True: required; False: never
        want = specialize if specialize is not None else (dtype == "float32" and os.environ.get("MJB_SPECIALIZE", "1") != "0")
        if want:
            try:
                self.specialize()
            except TemplateError as exc:
                if specialize:
                    raise
                global _WARNED_NO_SPEC
                if not _WARNED_NO_SPEC:
                    import warnings

                    warnings.warn(f"step kernel not specialised, using the generic kernel: {exc}", RuntimeWarning, stacklevel=2)
                    _WARNED_NO_SPEC = True

    # -- per-model specialised kernel ------------------------------------------------
    def spec_source(self) -> str:
        return _source_from(load_library().mjb_spec_source, self.ptr)

    def specialize(self) -> None:
        """Compile (or fetch from the in-tree cache) the step kernel specialised to this model's sizes and LDS layout and
        use it for every later launch on this object.  float32 only; identical arithmetic to the generic kernel."""
        if self.dtype != "float32":
            raise ConfigError("only the float32 step kernel is specialised")
        image = _read_private(compile_spec(self.spec_source()))
        _check(load_library().mjb_spec_load(self.ptr, image, len(image)))
        self.specialized = True
        # small batches are stepped by the two-wave kernel where it applies: specialise that one too
        if self.batch <= 1024:
            try:
                src2 = _source_from(load_library().mjb_step2_spec_source, self.ptr)
            except TemplateError:
                src2 = None
            if src2 is not None:
                image2 = _read_private(compile_spec(src2))
                _check(load_library().mjb_step2_spec_load(self.ptr, image2, len(image2)))

    def unspecialize(self) -> None:
        _check(load_library().mjb_step2_spec_unload(self.ptr))
        _check(load_library().mjb_spec_unload(self.ptr))
        self.specialized = False

    def fd_spec_source(self) -> str:
        return _source_from(load_library().mjb_fd_spec_source, self.ptr)

    def specialize_fd(self) -> None:
        """The float64 finite-difference kernel behind ``transition_fd`` specialised to this model (sizes, float64 LDS layout,
        the model baked in as constants); identical arithmetic to the generic kernel.  Done lazily by the first ``transition_fd``."""
        image = _read_private(compile_spec(self.fd_spec_source()))
        _check(load_library().mjb_fd_spec_load(self.ptr, image, len(image)))
        self.fd_specialized = True

    def unspecialize_fd(self) -> None:
        _check(load_library().mjb_fd_spec_unload(self.ptr))
        self.fd_specialized = False

    # -- plumbing -----------------------------------------------------------------
    def set_stream(self, stream_handle: int) -> None:
        _check(load_library().mjb_set_stream(self.ptr, ctypes.c_void_p(stream_handle)))

    def use_torch_stream(self) -> None:
        import torch

        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self) -> None:
        _check(load_library().mjb_sync(self.ptr))

    def array_ptr(self, name: str) -> tuple[int, int, int]:
        p, n, dt = ctypes.c_void_p(), ctypes.c_long(), ctypes.c_int()
        _check(load_library().mjb_array_ptr(self.ptr, name.encode(), ctypes.byref(p), ctypes.byref(n), ctypes.byref(dt)))
        return int(p.value or 0), int(n.value), int(dt.value)

    def torch_view(self, name: str):
        """Zero-copy torch tensor [batch, n] over the library-owned device array."""
        import torch

        ptr, n, dt = self.array_ptr(name)
        typestr = {0: "<f4", 1: "<f8", 2: "<i4"}[dt]
        if n == 0 or ptr == 0:
            return torch.zeros((self.batch, 0), device=f"cuda:{self.device}")
        return torch.as_tensor(_CudaArray(ptr, (self.batch, n), typestr, self), device=f"cuda:{self.device}")

    def get(self, name: str) -> np.ndarray:
        _, n, _ = self.array_ptr(name)
        out = np.zeros((self.batch, n), dtype=np.float64)
        if n:
            _check(load_library().mjb_get_array(self.ptr, name.encode(), out.ctypes.data))
        return out

    def set(self, name: str, value: np.ndarray) -> None:
        _, n, _ = self.array_ptr(name)
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64).reshape(-1, n) if n else np.zeros((self.batch, 0)), (self.batch, n)))
        if n:
            _check(load_library().mjb_set_array(self.ptr, name.encode(), arr.ctypes.data))

    # -- host mirror (mjb_host_view): pinned float64 block, one packed copy per direction ------------
    def host_view(self, name: str) -> np.ndarray:
        """numpy view [batch, n] over the library's pinned float64 mirror of ``name`` (valid while this object lives)."""
        p, n = ctypes.POINTER(ctypes.c_double)(), ctypes.c_long()
        _check(load_library().mjb_host_view(self.ptr, name.encode(), ctypes.byref(p), ctypes.byref(n)))
        if name == "engine_flags":
            return np.ctypeslib.as_array(p, shape=(1,))
        if n.value == 0:
            return np.zeros((self.batch, 0))
        arr = np.ctypeslib.as_array(p, shape=(self.batch, int(n.value)))
        self._keep = getattr(self, "_keep", [])
        self._keep.append(arr)
        return arr

    def sync_to_host(self) -> None:
        _check(load_library().mjb_sync_to_host(self.ptr))

    def sync_to_device(self, field_mask: int) -> None:
        _check(load_library().mjb_sync_to_device(self.ptr, int(field_mask)))

    def mirror_edited_mask(self) -> int:
        """Fields of the pinned mirror block the host edited in place since the library last refreshed / uploaded them (bit order of MIRROR_FIELDS)."""
        out = ctypes.c_int(0)
        _check(load_library().mjb_mirror_edited_mask(self.ptr, ctypes.byref(out)))
        return out.value

    def mirror_commit(self, field_mask: int = 63) -> None:
        _check(load_library().mjb_mirror_commit(self.ptr, int(field_mask)))

    def step_host_auto(self, nstep: int, compare: bool = True) -> int:
        """Edit detection + ``step_host`` + shadow refresh in ONE library call; returns the mask that was uploaded."""
        out = ctypes.c_int(0)
        _check(load_library().mjb_step_host_auto(self.ptr, int(nstep), 1 if compare else 0, ctypes.byref(out)))
        return out.value

    def step_host(self, nstep: int, field_mask: int = 0) -> None:
        """upload the edited mirror fields, ``nstep`` x mj_step (0: mj_forward), refresh the mirror — one library call."""
        _check(load_library().mjb_step_host(self.ptr, int(nstep), int(field_mask)))

    def engine_flags(self) -> int:
        """Sticky flag word of the batch after waiting for the stream (``mjb_engine_flags``): bit 0 contacts dropped, 1 constraint rows
        dropped, 2 bad-state auto-reset, 3 a ticket-mode launch timed out on a hand-over (the one that makes every synchronising call
        raise ``TemplateError`` until ``reset``).  No copy: the kernels keep the word in pinned host memory."""
        out = ctypes.c_int(0)
        _check(load_library().mjb_engine_flags(self.ptr, ctypes.byref(out)))
        return int(out.value)

    def counters(self) -> dict[str, np.ndarray]:
        out = np.zeros((self.batch, 8), dtype=np.int32)
        _check(load_library().mjb_get_counters(self.ptr, out.ctypes.data))
        return {k: out[:, i] for i, k in enumerate(COUNTER_NAMES)}

    # -- physics ------------------------------------------------------------------
    def reset(self, key: int = -1) -> None:
        _check(load_library().mjb_reset(self.ptr, int(key)))

    def forward(self) -> None:
        _check(load_library().mjb_forward(self.ptr))

    def inverse(self) -> None:
        """mj_inverse on every environment: fills the arrays ``qfrc_inverse`` [B, nv] and ``actuator_moment`` [B, nu*nv]."""
        _check(load_library().mjb_inverse(self.ptr))

    def step(self, nstep: int = 1) -> None:
        _check(load_library().mjb_step(self.ptr, int(nstep)))

    def rollout(self, nstep: int, ctrl_mode: int = CTRL_KEEP, seed: int = 0, step0: int = 0, ctrl_scale: float = 1.0,
                obs_spec: ObsSpecHandle | None = None, obs_out_ptr: int = 0, obs_every: int = 0) -> None:
        _check(load_library().mjb_rollout(self.ptr, int(nstep), int(ctrl_mode), int(seed) & 0xFFFFFFFF, int(step0) & 0xFFFFFFFF,
                                          float(ctrl_scale), obs_spec.ptr if obs_spec else None,
                                          ctypes.c_void_p(obs_out_ptr) if obs_out_ptr else None, int(obs_every)))

    def set_feedback(self, K: np.ndarray, u0: np.ndarray, q0: np.ndarray, v0: np.ndarray | None = None) -> None:
        m = self.model.compiled
        K = np.ascontiguousarray(K, dtype=np.float64)
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        v0 = np.zeros(m.nv) if v0 is None else np.ascontiguousarray(v0, dtype=np.float64)
        if K.shape != (m.nu, 2 * m.nv) or u0.shape != (m.nu,) or q0.shape != (m.nq,) or v0.shape != (m.nv,):
            raise ConfigError("feedback gains must have shapes K [nu, 2nv], u0 [nu], q0 [nq], v0 [nv]")
        _check(load_library().mjb_set_feedback(self.ptr, K.ctypes.data, u0.ctypes.data, q0.ctypes.data, v0.ctypes.data))

    def set_feedback_noise(self, std: np.ndarray | None, table: np.ndarray | None, env_stride: int = 0) -> None:
        """ctrl noise of the feedback law: ``+ std[a] * table[(step + env * env_stride) % nsteps, a]`` before the clip; None switches it off."""
        if std is None or table is None:
            _check(load_library().mjb_set_feedback_noise(self.ptr, None, None, 0, 0))
            return
        m = self.model.compiled
        std = np.ascontiguousarray(std, dtype=np.float64)
        table = np.ascontiguousarray(table, dtype=np.float64)
        if std.shape != (m.nu,) or table.ndim != 2 or table.shape[1] != m.nu or table.shape[0] < 1:
            raise ConfigError("feedback noise must have shapes std [nu], table [nsteps, nu]")
        _check(load_library().mjb_set_feedback_noise(self.ptr, std.ctypes.data, table.ctypes.data, int(table.shape[0]), int(env_stride)))

    def feedback_ctrl(self, step: int = 0) -> None:
        """Evaluate the feedback law for every environment on the device (fp32: MFMA GEMM) and write ``ctrl`` there."""
        _check(load_library().mjb_feedback_ctrl(self.ptr, int(step)))

    def make_obs_spec(self, flags: int, site_ids=(), body_ids=(), geom_ids=(), subtree_ids=()) -> ObsSpecHandle:
        return ObsSpecHandle(self, flags, site_ids, body_ids, geom_ids, subtree_ids)

    def obs_gather(self, spec: ObsSpecHandle, out_ptr: int) -> None:
        _check(load_library().mjb_obs_gather(self.ptr, spec.ptr, ctypes.c_void_p(out_ptr)))

    def transition_fd(self, eps: float = 1e-6, centered: bool = True, *, copy: bool = True) -> tuple[np.ndarray, np.ndarray]:
        """(A [batch, 2nv, 2nv], B [batch, 2nv, nu]).  ``copy=False`` returns views of the library's pinned result blocks (no
        16 MB host copy at humanoid batch 512), valid until the next ``transition_fd`` on this object."""
        m = self.model.compiled
        if not self._fd_spec_tried:                                # first call: the per-model specialised kernel, by the same policy as the step kernel
            self._fd_spec_tried = True
            want = self._spec_policy if self._spec_policy is not None else os.environ.get("MJB_SPECIALIZE", "1") != "0"
            if want:
                try:
                    self.specialize_fd()
                except TemplateError as exc:
                    if self._spec_policy:
                        raise
                    import warnings

                    warnings.warn(f"finite-difference kernel not specialised, using the generic kernel: {exc}", RuntimeWarning, stacklevel=2)
        pa, pb = ctypes.POINTER(ctypes.c_double)(), ctypes.POINTER(ctypes.c_double)()
        _check(load_library().mjb_transition_fd_pinned(self.ptr, float(eps), int(bool(centered)), ctypes.byref(pa), ctypes.byref(pb)))
        A = np.ctypeslib.as_array(pa, shape=(self.batch, 2 * m.nv, 2 * m.nv))
        Bv = np.ctypeslib.as_array(pb, shape=(self.batch, 2 * m.nv, m.nu)) if m.nu else np.zeros((self.batch, 2 * m.nv, 0))
        return (A.copy(), Bv.copy()) if copy else (A, Bv)

    def jac(self, kinds: Sequence[int], ids: Sequence[int]) -> tuple[np.ndarray, np.ndarray]:
        m = self.model.compiled
        k, kp = _ids(kinds); i, ip = _ids(ids)
        jp = np.zeros((self.batch, k.size, 3, m.nv))
        jr = np.zeros((self.batch, k.size, 3, m.nv))
        _check(load_library().mjb_jac(self.ptr, k.size, kp, ip, jp.ctypes.data, jr.ctypes.data))
        return jp, jr

    def profile_get(self) -> np.ndarray:
        out = np.zeros(24, dtype=np.uint64)
        _check(load_library().mjb_profile_get(self.ptr, out.ctypes.data))
        return out

    def schedule_info(self) -> dict:
        """How the last stepping launch mapped work to workgroups (``mjb_step_schedule``)."""
        out = np.zeros(6, dtype=np.int32)
        _check(load_library().mjb_step_schedule(self.ptr, out.ctypes.data))
        return {"launch_steps": int(out[0]), "env_blocks": int(out[1]), "resident_slots": int(out[2]), "chunk_steps": int(out[3]),
                "map": "tickets" if out[3] > 0 else "static", "fair_bit": int(out[4]), "waves_per_env": 2 if out[5] else 1}

    def profile_env_get(self) -> np.ndarray:
        out = np.zeros((self.batch, 4), dtype=np.uint64)
        _check(load_library().mjb_profile_env_get(self.ptr, out.ctypes.data))
        return out

    def debug_forward(self) -> None:
        _check(load_library().mjb_debug_forward(self.ptr))

    def debug_get(self, name: str) -> np.ndarray:
        m = self.model.compiled
        nv = m.nv
        per = {"qM": nv * nv, "qfrc_bias": nv, "qfrc_passive": nv, "qfrc_actuator": nv, "qacc_smooth": nv, "qfrc_constraint": nv,
               "efc_J": self.nefcmax * nv, "efc_aref": self.nefcmax, "efc_D": self.nefcmax, "efc_pos": self.nefcmax,
               "efc_force": self.nefcmax, "con": self.nconmax * 11, "cdof": 6 * nv, "cinert": 10 * m.nbody, "cvel": 6 * m.nbody,
               "efc_type": self.nefcmax}[name]
        out = np.zeros((self.batch, per), dtype=np.int32 if name == "efc_type" else np.float64)
        _check(load_library().mjb_debug_get(self.ptr, name.encode(), out.ctypes.data, out.size))
        return out

    def __del__(self):
        try:
            if self.ptr:
                load_library().mjb_data_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


__all__ = ["BatchSim", "DeviceModel", "ObsSpecHandle", "MIRROR_FIELDS", "build_library", "load_library", "CTRL_KEEP", "CTRL_ZERO", "CTRL_RANDOM", "CTRL_FEEDBACK"]
