"""Humanoid free-running drift: fp32 HIP and float64 HIP against the float64 oracle, next to the INTRINSIC error growth of the
system (float64 HIP started 1e-7 away from the same state) — VERDICT r1 "Next round" #3.

    python scripts/gpu_humanoid_drift.py [B] [ctrl_scale ...]      -> profiles/r02_humanoid_drift.log (stdout)

Columns per horizon T: error of qpos vs the oracle (max over the coordinates of an environment): median / 90th pct / max over the
batch, fraction of environments above 1e-4; the same for the perturbed float64 twin; fraction of environments in contact.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mujoco_template_amd._capi import CTRL_RANDOM, BatchSim, DeviceModel
from mujoco_template_amd.mjcf import compile_xml_path
from oracle import mjo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scales = [float(s) for s in sys.argv[2:]] or [1.0, 0.1, 0.0]
cm = compile_xml_path(os.path.join(ROOT, "models/humanoid.xml"))
om = mjo.OracleModel(cm)
dm = DeviceModel(cm)
if os.environ.get("MJB_TOL"):                                          # experiment: solver tolerance of the fp32 / float64 kernels (the oracle keeps 1e-8)
    dm.set_solver(cm.iterations, float(os.environ["MJB_TOL"]))
rng = np.random.default_rng(0)
HORIZONS = (1, 10, 20, 40, 60, 100, 150, 200, 300, 400, 500, 600, 800, 1000)


def stats(e):
    m = e.max(1)
    return f"med {np.median(m):.1e} p90 {np.quantile(m, 0.9):.1e} max {m.max():.1e} >1e-4 {(m > 1e-4).mean():.3f}"


for scale in scales:
    q0 = np.tile(cm.qpos0, (B, 1))
    q0[:, 2] += rng.uniform(0, 0.01, size=B)                       # root z jitter
    qp = q0.copy()
    qp[:, :3] += rng.uniform(-1e-7, 1e-7, size=(B, 3))             # the twin: 1e-7 away (one fp32 ulp of a coordinate of size 1)
    qp[:, 7:] += rng.uniform(-1e-7, 1e-7, size=(B, cm.nq - 7))
    sims = {"f32": BatchSim(dm, B, dtype="float32"), "f64": BatchSim(dm, B, dtype="float64"), "f64+1e-7": BatchSim(dm, B, dtype="float64")}
    for k, s in sims.items():
        s.set("qpos", qp if k == "f64+1e-7" else q0)
    done = 0
    print(f"# humanoid B={B} ctrl_scale={scale} seed=1 (error = max over coordinates of |qpos - oracle|, per environment)", flush=True)
    for T in HORIZONS:
        for s in sims.values():
            s.rollout(T - done, CTRL_RANDOM, seed=1, step0=done, ctrl_scale=scale)
        done = T
        qT, _ = mjo.rollout_batch(om, B, T, seed=1, scale=scale, nthreads=16, qpos_init=q0)
        cn = sims["f32"].counters()
        line = f"T {T:4d} | "
        for k, s in sims.items():
            line += f"{k}: {stats(np.abs(s.get('qpos') - qT))} | "
        line += f"in contact {(cn['ncon'] > 0).mean():.2f} mean nefc {cn['nefc'].mean():.1f} z_root med {np.median(qT[:, 2]):.2f}"
        print(line, flush=True)
