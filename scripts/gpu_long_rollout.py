"""Row / contact cap headroom over a long humanoid rollout (random ctrl, the humanoids fall and thrash on the floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 4096
sim = BatchSim(DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml"))), B, dtype="float32")
mx = mc = 0
for k in range(50):
    sim.rollout(100, CTRL_RANDOM, seed=0, step0=100 * k)
    cn = sim.counters()
    mx = max(mx, int(cn["nefc"].max())); mc = max(mc, int(cn["ncon"].max()))
    if k % 10 == 9:
        print(f"step {100*(k+1)}: max nefc so far (sampled every 100 steps) {mx} of {sim.nefcmax}, max ncon {mc} of {sim.nconmax}, dropped rows {int(cn['efc_dropped'].sum())} contacts {int(cn['con_dropped'].sum())}, "
              f"bad-state resets {int(cn['warn_badqpos'].sum() + cn['warn_badqvel'].sum() + cn['warn_badqacc'].sum())}, mean niter {cn['solver_niter'].mean():.2f}", flush=True)
