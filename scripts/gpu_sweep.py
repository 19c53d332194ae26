"""Throughput of the humanoid rollout for several LDS caps / batch sizes (fp32)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
for B, ne, nc in ((512, 0, 0), (4096, 0, 0), (4096, 96, 32), (4096, 80, 24), (4096, 64, 24), (4096, 48, 16), (16384, 64, 24)):
    sim = BatchSim(dm, B, dtype="float32", nefcmax=ne, nconmax=nc)
    sim.rollout(20, CTRL_RANDOM, seed=1); sim.sync()
    t = time.time(); sim.rollout(200, CTRL_RANDOM, seed=1, step0=20); sim.sync(); dt = time.time() - t
    cn = sim.counters()
    print(f"B={B} caps {sim.nefcmax}/{sim.nconmax} lds/env={sim.lds_bytes_per_env}: {B*200/dt:.3e} env-steps/s ({dt*1e3/200:.3f} ms/step) dropped {cn['efc_dropped'].sum()}/{cn['con_dropped'].sum()} max nefc {cn['nefc'].max()}", flush=True)
