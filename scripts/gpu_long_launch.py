"""One fused launch of 5000 steps (ticket map, 32-step chunks + taper) against 50 launches of 100 steps: same final state?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
B = 4096
a = BatchSim(dm, B, dtype="float32"); b = BatchSim(dm, B, dtype="float32")
t = time.perf_counter(); a.rollout(5000, CTRL_RANDOM, seed=0); a.sync(); ta = time.perf_counter() - t
t = time.perf_counter()
for k in range(50): b.rollout(100, CTRL_RANDOM, seed=0, step0=100 * k)
b.sync(); tb = time.perf_counter() - t
print(f"one 5000-step launch {ta*1e3:.1f} ms ({B*5000/ta/1e6:.2f} M env-steps/s, {a.schedule_info()}); 50 x 100 steps {tb*1e3:.1f} ms ({B*5000/tb/1e6:.2f} M)")
print("bitwise equal final qpos / qvel / time:", np.array_equal(a.get("qpos"), b.get("qpos")), np.array_equal(a.get("qvel"), b.get("qvel")), np.array_equal(a.get("time"), b.get("time")))
a.sync_to_host(); print("engine flags", int(a.host_view("engine_flags")[0]))
