"""Small, time-boxed check of k_step's ticket mode: bitwise equal to the static map?  (run under `timeout`)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["MJB_SCHED_DEBUG"] = "1"
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 4
res = {}
for chunk in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,2").split(",")]:
    os.environ["MJB_CHUNK_STEPS"] = str(chunk); os.environ["MJB_FAIR_BIT"] = "0"
    print(f"chunk={chunk}: create", flush=True)
    sim = BatchSim(dm, B, dtype="float32")
    print("  launch", flush=True)
    t = time.perf_counter(); sim.rollout(nstep, CTRL_RANDOM, seed=1); sim.sync(); dt = time.perf_counter() - t
    print(f"  done in {dt*1e3:.2f} ms", flush=True)
    t = time.perf_counter(); sim.rollout(nstep, CTRL_RANDOM, seed=1, step0=nstep); sim.sync(); dt = time.perf_counter() - t
    res[chunk] = (sim.get("qpos").copy(), sim.get("qvel").copy(), sim.get("time").copy())
    fl = sim.host_view("engine_flags"); sim.sync_to_host()
    print(f"  second launch {dt*1e3:.2f} ms; engine flags {int(fl[0])}; qpos finite {np.isfinite(res[chunk][0]).all()}", flush=True)
    del sim
k0 = sorted(res)[0]
for k in sorted(res):
    print(f"chunk {k} vs {k0}: qpos equal {np.array_equal(res[k][0], res[k0][0])} qvel equal {np.array_equal(res[k][1], res[k0][1])} time equal {np.array_equal(res[k][2], res[k0][2])}"
          f" max|dqpos| {np.abs(res[k][0]-res[k0][0]).max():.3e}")
