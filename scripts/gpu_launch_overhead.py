"""Where do the ~0.17 ms between the kernel time and the wall time of ONE short fused launch go (driver's bench shape: 20 steps)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_template_amd import Env, ObservationSpec, RandomCtrlController
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), obs_spec=ObservationSpec(as_dict=False), controller=RandomCtrlController(seed=0), batch=4096, dtype="float32")
sim = env.data.sim
sim.use_torch_stream()
env.rollout(5, obs_every=5); torch.cuda.synchronize()
N = 20
rows = []
for rep in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    t1 = time.perf_counter()
    obs = env.rollout(N, obs_every=N)
    t2 = time.perf_counter()
    e1.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, e0.elapsed_time(e1)))
import numpy as np
r = np.array(rows[5:]) * np.array([1e3, 1e3, 1e3, 1e3, 1e3, 1.0])
m = np.median(r, axis=0)
print(f"median over {len(r)} launches of {N} steps (ms): event.record {m[0]:.4f} | Env.rollout (enqueue) {m[1]:.4f} | event.record {m[2]:.4f} | synchronize {m[3]:.4f} | wall {m[4]:.4f} | events (kernel) {m[5]:.4f} | wall - kernel {m[4]-m[5]:.4f}")
# the same through the C ABI only
from mujoco_template_amd._capi import CTRL_RANDOM
rows = []
for rep in range(30):
    t0 = time.perf_counter(); sim.rollout(N, CTRL_RANDOM, seed=0, step0=1000 + N * rep); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t2 - t0))
m = np.median(np.array(rows[5:]) * 1e3, axis=0)
print(f"BatchSim.rollout only (ms): enqueue {m[0]:.4f} | synchronize {m[1]:.4f} | wall {m[2]:.4f}")
