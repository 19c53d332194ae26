import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
dm = DeviceModel(compile_xml_path("models/humanoid.xml"))
for B in (256, 512, 768, 1024, 1536, 2048):
    sim = BatchSim(dm, B, dtype="float32")
    sim.rollout(150, CTRL_RANDOM, seed=1); sim.sync()
    ts = []
    for r in range(7):
        t = time.perf_counter(); sim.rollout(100, CTRL_RANDOM, seed=1, step0=150 + 100 * r); sim.sync(); ts.append(time.perf_counter() - t)
    print(f"MJB_TWO_WAVE={os.environ.get('MJB_TWO_WAVE','auto')} B={B}: {min(ts)/100*1e6:6.2f} us/step  {B*100/min(ts)/1e6:6.2f} M/s  sched {sim.schedule_info()}", flush=True)
