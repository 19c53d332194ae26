"""k_step work scheduling on the humanoid: static map vs (block, chunk) tickets, priority hand-over on / off.
MJB_CHUNK_STEPS / MJB_FAIR_BIT are read when a data object first launches, so every setting gets a fresh BatchSim."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))

def run(B, nstep, chunk, fair, reps=5):
    for k, v in (("MJB_CHUNK_STEPS", chunk), ("MJB_FAIR_BIT", fair)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    sim = BatchSim(dm, B, dtype="float32")
    sim.rollout(100, CTRL_RANDOM, seed=1); sim.sync()
    ts = []
    for r in range(reps):
        t = time.perf_counter(); sim.rollout(nstep, CTRL_RANDOM, seed=1, step0=100 + nstep * r); sim.sync(); ts.append(time.perf_counter() - t)
    q = sim.get("qpos").copy(); c = sim.counters().copy() if hasattr(sim, "counters") else None
    del sim
    return min(ts), sorted(ts)[len(ts) // 2], q

ref = {}
for B in [int(x) for x in os.environ.get("SCHED_B", "4096").split(",")]:
    for nstep in (20, 100):
        for chunk, fair in ((0, 0), (0, 15), (None, None), (2, 0), (3, 0), (4, 0), (6, 0), (8, 0), (12, 0), (16, 0)):
            best, med, q = run(B, nstep, chunk, fair)
            key = (B, nstep)
            if key not in ref: ref[key] = q
            same = bool(np.array_equal(ref[key], q))
            print(f"B={B} steps/launch={nstep:4d} chunk={str(chunk):>4s} fair={str(fair):>4s}: best {best*1e3:7.3f} ms  median {med*1e3:7.3f} ms  {B*nstep/best/1e6:6.2f} M env-steps/s   qpos bitwise == static: {same}", flush=True)
