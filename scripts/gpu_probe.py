import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def probe(xml, B, dtype="float32", lanes=0, nsteps=200, nefcmax=0, nconmax=0, scale=1.0):
    cm = compile_xml_path(os.path.join(ROOT, xml))
    dm = DeviceModel(cm)
    sim = BatchSim(dm, B, dtype=dtype, lanes=lanes, nefcmax=nefcmax, nconmax=nconmax)
    sim.rollout(10, CTRL_RANDOM, seed=1, ctrl_scale=scale); sim.sync()
    t = time.time(); sim.rollout(nsteps, CTRL_RANDOM, seed=1, step0=10, ctrl_scale=scale); sim.sync(); dt = time.time() - t
    cn = sim.counters()
    print(f"{xml} B={B} {dtype} lanes={sim.lanes} lds/env={sim.lds_bytes_per_env} nefcmax={sim.nefcmax}: {B*nsteps/dt:.3e} env-steps/s ({dt*1e3/nsteps:.3f} ms/step) "
          f"nefc mean {cn['nefc'].mean():.1f} max {cn['nefc'].max()} niter mean {cn['solver_niter'].mean():.2f} dropped {cn['efc_dropped'].sum()} {cn['con_dropped'].sum()} "
          f"bad {cn['warn_badqacc'].sum()+cn['warn_badqpos'].sum()+cn['warn_badqvel'].sum()}", flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "humanoid"):
    for B in (512, 1024, 4096, 16384):
        probe("models/humanoid.xml", B)
    probe("models/humanoid.xml", 4096, nefcmax=96, nconmax=32)
    probe("models/humanoid.xml", 4096, nefcmax=64, nconmax=24)
    probe("models/humanoid.xml", 4096, nsteps=1000)
    probe("models/humanoid.xml", 1024, dtype="float64")
if which in ("all", "small"):
    probe("models/cartpole.xml", 1024, scale=0.005)
    probe("models/cartpole.xml", 65536, scale=0.005)
    probe("models/cartpole.xml", 65536, scale=0.005, lanes=16)
    probe("models/drone2/scene.xml", 2048, scale=0.3)
    probe("models/drone2/scene.xml", 65536, scale=0.3)
    probe("models/pendulum.xml", 65536)
