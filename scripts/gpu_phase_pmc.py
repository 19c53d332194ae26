"""Workload of the per-phase PMC passes (scripts/run_phase_pmc_r03.sh): the bench's rollout (humanoid, B = 4096, random ctrl, fp32,
specialised kernel) in 100-step launches - 2 warm-up + 10 counted - with the DIAGNOSTIC kernel build -DMJB_PHASE_REPEAT, in which the
phase named by MJB_REPEAT_PHASE (index PH_*, mjb_types.hpp; -1 = none) runs twice per step.  The difference of the hardware counters
between a pass with phase k repeated and the pass with none is phase k's own count.  A full-lane torch kernel runs beside it as the
unit check of SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (64 active lanes)."""
import os, sys
os.environ["MJB_SPEC_FLAGS"] = (os.environ.get("MJB_SPEC_FLAGS", "") + " -DMJB_PHASE_REPEAT").strip()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sim = BatchSim(DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml"))), B, dtype="float32")
assert sim.specialized
for k in range(12):
    sim.rollout(100, CTRL_RANDOM, seed=0, step0=100 * k)
sim.sync()
x = torch.ones(1 << 24, device="cuda")
for _ in range(4):
    x.mul_(1.0001)
torch.cuda.synchronize()
cn = sim.counters()
print("rep", os.environ.get("MJB_REPEAT_PHASE", "-1"), "mean nefc", float(cn["nefc"].mean()), "iters", float(cn["solver_niter"].mean()), "flags", sim.engine_flags())
