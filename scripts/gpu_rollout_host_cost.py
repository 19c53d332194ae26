"""Host-side cost of one fused launch through the Env API (what a short launch pays on top of its kernel): wall time of the
asynchronous `Env.rollout` call, of the `BatchSim.rollout` call inside it, of `torch.cuda.synchronize`, and the event-timed kernel.

    python scripts/gpu_rollout_host_cost.py [steps_per_launch=20] [batch=4096]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mujoco_template_amd as mt

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = mt.Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), controller=mt.RandomCtrlController(seed=0, scale=1.0), batch=B, dtype="float32")
sim = env.data.sim
inner = []
orig = sim.rollout
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); inner.append(time.perf_counter() - t); return r
sim.rollout = timed
for _ in range(5):
    env.rollout(steps, obs_every=steps)
torch.cuda.synchronize()
inner.clear()
call, sync, wall, kern = [], [], [], []
for it in range(50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    obs = env.rollout(steps, obs_every=steps)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    call.append(t1 - t0); sync.append(t2 - t1); wall.append(t2 - t0); kern.append(e0.elapsed_time(e1) * 1e-3)
us = lambda x: f"{1e6 * float(np.median(x)):8.1f} us"
print(f"B = {B}, {steps} steps per launch, medians of 50")
print("Env.rollout call (asynchronous, incl. the two event records):", us(call))
print("  of which BatchSim.rollout (ctypes -> mjb_rollout)          :", us(inner))
print("event-timed launch (kernel + gaps on the stream)              :", us(kern))
print("wall, record .. synchronize returned                          :", us(wall))
print("wall - event-timed                                            :", us(np.array(wall) - np.array(kern)))
