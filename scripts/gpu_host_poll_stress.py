"""Stress of the polled completion of host-driven steps: after EVERY mjb_step_host the pinned mirror must already hold the state the device
arrays hold (fetched afterwards through the synchronising getter).  A stale mirror at return time would show as a mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd import Env, mj
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, B, N in (("humanoid", 1, 4000), ("humanoid", 8, 2000), ("cartpole", 3, 4000), ("drone2/scene", 5, 3000)):
    env = Env.from_xml_path(os.path.join(ROOT, f"models/{name}.xml"), batch=B, dtype="float32")
    d, m, sim = env.data, env.model, env.data.sim
    rng = np.random.default_rng(1)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    bad = 0
    for s in range(N):
        d.ctrl[...] = rng.uniform(lo, hi, size=np.shape(d.ctrl)) * 0.3
        mj.mj_step(m, d)
        host = (np.array(d.qpos, dtype=np.float64).reshape(B, -1), np.array(d.qvel, dtype=np.float64).reshape(B, -1), np.array(d.qacc_warmstart, dtype=np.float64).reshape(B, -1))
        dev = (sim.get("qpos"), sim.get("qvel"), sim.get("qacc_warmstart"))
        if not all(np.array_equal(a, b) for a, b in zip(host, dev)) or abs(float(np.ravel(d.time)[0]) - float(sim.get("time")[0, 0])) > 0:
            bad += 1
    print(f"{name} B={B}: {N} polled host-driven steps, mirror != device after return: {bad}; schedule {sim.schedule_info()['waves_per_env']} wave(s) per env")
