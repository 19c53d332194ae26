import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import mjo
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cm = compile_xml_path(os.path.join(ROOT, "models/cartpole.xml")); om = mjo.OracleModel(cm); dm = DeviceModel(cm)
B = 1024
rng = np.random.default_rng(0)
for scale in (0.005, 0.05, 1.0):
    q = np.zeros((B, 2)); q[:, 1] = rng.uniform(-0.05, 0.05, size=B)
    sim = BatchSim(dm, B, dtype="float32"); sim.set("qpos", q)
    sim64 = BatchSim(dm, B, dtype="float64"); sim64.set("qpos", q)
    done = 0
    for T in (50, 100, 150, 200, 300, 500, 1000):
        sim.rollout(T - done, CTRL_RANDOM, seed=1, step0=done, ctrl_scale=scale)
        sim64.rollout(T - done, CTRL_RANDOM, seed=1, step0=done, ctrl_scale=scale)
        done = T
        qT, _ = mjo.rollout_batch(om, B, T, seed=1, scale=scale, nthreads=16, qpos_init=q)
        e32 = np.abs(sim.get("qpos") - qT); e64 = np.abs(sim64.get("qpos") - qT)
        cn = sim.counters()
        print(f"scale {scale} T {T}: fp32 err max {e32.max():.2e} median {np.median(e32.max(1)):.2e} frac>1e-4 {(e32.max(1)>1e-4).mean():.3f} | f64 err {e64.max():.2e} | nefc>0 frac {(cn['nefc']>0).mean():.3f} |theta| max {np.abs(qT[:,1]).max():.2f}", flush=True)
