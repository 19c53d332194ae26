"""What would 3 waves/SIMD buy?  Same (deliberately small) LDS caps with the 2-wave and the 3-wave register budget.
Usage: MJB_LIB=<path to .so> python scripts/gpu_occupancy_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
for B, ne, nc in ((12288, 20, 8), (12288, 64, 24)):
    sim = BatchSim(dm, B, dtype="float32", nefcmax=ne, nconmax=nc)
    sim.rollout(20, CTRL_RANDOM, seed=1); sim.sync()
    t = time.time(); sim.rollout(200, CTRL_RANDOM, seed=1, step0=20); sim.sync(); dt = time.time() - t
    cn = sim.counters()
    print(f"{os.environ.get('MJB_LIB','default')}: B={B} caps {sim.nefcmax}/{sim.nconmax} lds/env={sim.lds_bytes_per_env}: {B*200/dt:.3e} env-steps/s dropped {cn['efc_dropped'].sum()}/{cn['con_dropped'].sum()} mean nefc {cn['nefc'].mean():.1f}", flush=True)
