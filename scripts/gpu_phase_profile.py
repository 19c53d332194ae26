"""Per-phase cycle shares of k_step: the per-model specialised kernel compiled with in-kernel cycle stamps (MJB_SPEC_FLAGS=-DMJB_PROFILE;
read the SHARES, the stamps themselves cost time)."""
import os, sys
os.environ["MJB_SPEC_FLAGS"] = (os.environ.get("MJB_SPEC_FLAGS", "") + " -DMJB_PROFILE").strip()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MODEL = sys.argv[2] if len(sys.argv) > 2 else "humanoid"
XML = {"humanoid": "models/humanoid.xml", "cartpole": "models/cartpole.xml", "drone2": "models/drone2/scene.xml", "pendulum": "models/pendulum.xml"}[MODEL]
SCALE = {"humanoid": 1.0, "cartpole": 0.005, "drone2": 0.3, "pendulum": 1.0}[MODEL]
sim = BatchSim(DeviceModel(compile_xml_path(os.path.join(ROOT, XML))), B, dtype="float32")
sim.rollout(50, CTRL_RANDOM, seed=1, ctrl_scale=SCALE); sim.sync(); sim.profile_get()
n = 100
sim.rollout(n, CTRL_RANDOM, seed=1, step0=50, ctrl_scale=SCALE); sim.sync()
p = sim.profile_get().astype(float) / (B * n)
names = ["kinematics", "com_pos", "crb+factorM", "collision", "constraints", "vel/bias/passive", "actuation+Msolve", "solver(rest)", "integrate(euler)", "other", "solver:direction(H,chol,solve)", "solver:linesearch"]
tot = p[:12].sum() + p[15] + p[20:24].sum()
print(f"{MODEL} B={B} lanes={sim.lanes}: clock ticks per env-step {tot:.0f} (s_memtime)")
for k, nm in enumerate(names):
    print(f"  {nm:34s} {p[k]:9.0f}  {100*p[k]/tot:5.1f}%")
print(f"  {'solver:Mv,jv products':34s} {p[15]:9.0f}  {100*p[15]/tot:5.1f}%")
print(f"  per env-step: line-search iterations {p[12]:.2f}, Newton directions {p[13]:.2f}, Hessian factorisations {p[14]:.2f}")
print(f"  inside all factorisations (M, M+hD, H): {p[19]:.0f} cycles = load/assembly {p[16]:.0f} + panel loop {p[17]:.0f} + store/back-substitution {p[18]:.0f}")
if p[20:24].sum() > 0:      # scratch slots for ad-hoc MJB_STAMP(c, 20..23) inside a phase
    print(f"  sub-stamps: [20] {p[20]:.0f}  [21] {p[21]:.0f}  [22] {p[22]:.0f}  [23] {p[23]:.0f}")
