"""TEST-ONLY driver of the host emulation of the HIP kernel source (see mjb_emu.cpp)."""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, "libmjb_emu.so")
    srcs = [os.path.join(_HERE, "mjb_emu.cpp")] + [
        os.path.join(_ROOT, "mujoco_template_amd", "csrc", f)
        for f in ("mjb_device.hpp", "mjb_host.hpp", "mjb_hostemu.hpp", "mjb_types.hpp")
    ]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas",
                               "-o", so, srcs[0]])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.mjbemu_last_error.restype = ctypes.c_char_p
    return _LIB


class EmuEnv:
    """One environment advanced by the emulated kernel; arrays are float64 numpy."""

    def __init__(self, compiled, G: int = 16, use_double: bool = True, ncon_max: int = 0, nefc_max: int = 0):
        from mujoco_template_amd._pack import PackedTable

        self.m = compiled
        self.packed = PackedTable(compiled)
        self.G, self.use_double = G, use_double
        self.ncon_max = ncon_max or 64
        self.nefc_max = nefc_max or 160
        m = compiled
        nv = m.nv
        self.io = {
            "qpos": np.array(m.qpos0, dtype=np.float64), "qvel": np.zeros(nv), "ctrl": np.zeros(max(m.nu, 1)),
            "qacc": np.zeros(nv), "qacc_warmstart": np.zeros(nv), "time": np.zeros(1), "counters": np.zeros(8, dtype=np.int32),
            "xpos": np.zeros(3 * m.nbody), "xquat": np.zeros(4 * m.nbody), "xipos": np.zeros(3 * m.nbody),
            "site_xpos": np.zeros(max(3 * m.nsite, 1)), "geom_xpos": np.zeros(max(3 * m.ngeom, 1)), "subtree_com": np.zeros(3 * m.nbody),
            "sensordata": np.zeros(max(m.nsensordata, 1)),
            "qfrc_inverse": np.zeros(nv), "actuator_moment": np.zeros(max(m.nu, 1) * nv),
            "qM": np.zeros(nv * nv), "qfrc_bias": np.zeros(nv), "qfrc_passive": np.zeros(nv), "qfrc_actuator": np.zeros(nv),
            "qacc_smooth": np.zeros(nv), "qfrc_constraint": np.zeros(nv),
            "efc_J": np.zeros(self.nefc_max * nv), "efc_aref": np.zeros(self.nefc_max), "efc_D": np.zeros(self.nefc_max),
            "efc_pos": np.zeros(self.nefc_max), "efc_force": np.zeros(self.nefc_max), "efc_type": np.zeros(self.nefc_max, dtype=np.int32),
            "con": np.zeros(self.ncon_max * 11), "cdof": np.zeros(6 * nv), "cinert": np.zeros(10 * m.nbody), "cvel": np.zeros(6 * m.nbody),
        }

    def __getattr__(self, k):
        io = self.__dict__.get("io", {})
        if k in io:
            return io[k]
        raise AttributeError(k)

    def run(self, nstep: int = 1, mode: int = 0, ctrl_mode: int = 0, seed: int = 0, step0: int = 0, env0: int = 0, scale: float = 1.0) -> None:
        p = self.packed
        names = list(self.io.keys())
        n = len(names)
        cn = (ctypes.c_char_p * n)(*[k.encode() for k in names])
        cp = (ctypes.c_void_p * n)(*[self.io[k].ctypes.data for k in names])
        rc = lib().mjbemu_run(p.n, p.names, p.ptrs, p.dtypes, p.counts, n, cn, cp,
                              ctypes.c_int(self.G), ctypes.c_int(int(self.use_double)), ctypes.c_int(self.ncon_max), ctypes.c_int(self.nefc_max),
                              ctypes.c_int(nstep), ctypes.c_int(ctrl_mode), ctypes.c_uint(seed), ctypes.c_uint(step0), ctypes.c_uint(env0),
                              ctypes.c_float(scale), ctypes.c_int(mode))
        if rc != 0:
            raise RuntimeError(lib().mjbemu_last_error().decode())

    def inverse(self) -> None:
        self.run(1, mode=2)

    def forward(self) -> None:
        self.run(mode=1)

    def step(self, n: int = 1) -> None:
        self.run(nstep=n, mode=0)
