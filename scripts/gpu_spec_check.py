"""Generic vs per-model specialised fp32 step kernel: bitwise state comparison and throughput (humanoid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, scale in (("humanoid", 1.0), ("drone2/scene", 0.3), ("cartpole", 0.01)):
    dm = DeviceModel(compile_xml_path(os.path.join(ROOT, f"models/{name}.xml")))
    B = 4096
    out = {}
    for spec in (False, True):
        t0 = time.time()
        sim = BatchSim(dm, B, dtype="float32", specialize=spec)
        tc = time.time() - t0
        sim.rollout(20, CTRL_RANDOM, seed=1, ctrl_scale=scale); sim.sync()
        t = time.time(); sim.rollout(200, CTRL_RANDOM, seed=1, step0=20, ctrl_scale=scale); sim.sync(); dt = time.time() - t
        out[spec] = (sim.get("qpos"), sim.get("qvel"), sim.counters())
        print(f"{name:14s} specialised={spec!s:5s} create {tc:5.1f}s  {B*200/dt:.3e} env-steps/s", flush=True)
    same = np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
    print(f"{name:14s} states bitwise equal: {same}; max|dqpos| {np.abs(out[False][0]-out[True][0]).max():.2e}; nefc equal {np.array_equal(out[False][2]['nefc'], out[True][2]['nefc'])}", flush=True)
