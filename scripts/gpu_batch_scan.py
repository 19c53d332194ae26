"""Launch time vs batch size (residency rounds of 2048 environments): humanoid, 100-step launches after 200 warm-up steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
dm = DeviceModel(compile_xml_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models/humanoid.xml")))
for B in (1024, 2048, 3072, 4096, 6144, 8192):
    sim = BatchSim(dm, B, dtype="float32")
    sim.rollout(200, CTRL_RANDOM, seed=0); sim.sync()
    t = time.time(); sim.rollout(100, CTRL_RANDOM, seed=0, step0=200); sim.sync(); dt = time.time() - t
    t = time.time(); sim.rollout(100, CTRL_RANDOM, seed=0, step0=300); sim.sync(); dt2 = time.time() - t
    print(f"B={B}: {min(dt,dt2)*1e3:.2f} ms per 100-step launch, {B*100/min(dt,dt2):.3e} env-steps/s", flush=True)
