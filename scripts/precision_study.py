"""Would a mixed-precision step kernel lower the fp32 drift of the headline workload?

Answered on the CPU with the float64 oracle and its fp32-STORAGE model (``mjo_set_round_mask``: the outputs of the chosen phases
are rounded to fp32 where they are handed to the next phase, the arithmetic inside a phase stays float64 - a LOWER bound of what
a real fp32 phase injects).  Humanoid, full-range random ctrl, the bench's streams; drift = max |dqpos| against the unrounded
float64 oracle per environment, median / 90th percentile over the sample.  The kernel's own measured curve (profiles/
r02_humanoid_drift.log: median 2.5e-5 @ 60, 1.0e-4 @ 100) is the reference for how much the real fp32 arithmetic adds.

    python scripts/precision_study.py [--envs 128]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mujoco_template_amd.mjcf import compile_xml_path  # noqa: E402
from oracle import mjo  # noqa: E402

KIN, COM, CRB, CONS, VEL, RNE, FRC, ACC, SOLVER, STATE = (1 << k for k in range(10))
ALL = (1 << 10) - 1
VARIANTS = [
    ("everything stored in fp32 (model of today's kernel)", ALL),
    ("VERDICT's proposal: CRB->M, RNE bias, M^-1 f in float64", ALL & ~(CRB | RNE | ACC)),
    ("  + COM-frame quantities and velocities in float64", ALL & ~(CRB | RNE | ACC | COM | VEL)),
    ("  + kinematics in float64 (fp32: constraints, forces, solver, state)", ALL & ~(CRB | RNE | ACC | COM | VEL | KIN)),
    ("  + passive / actuator forces in float64 (fp32: constraints, solver, state)", CONS | SOLVER | STATE),
    ("only the solver's output and the state in fp32", SOLVER | STATE),
    ("only the STATE (qpos, qvel) in fp32, all arithmetic float64", STATE),
    ("only kinematics outputs in fp32", KIN),
    ("only the solver's output (qacc, forces) in fp32", SOLVER),
]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=128)
    args = ap.parse_args()
    cm = compile_xml_path(os.path.join(ROOT, "models", "humanoid.xml"))
    marks = (20, 60, 100, 200)

    def run(mask: int) -> np.ndarray:
        om = mjo.OracleModel(cm)
        om.set_round_mask(mask)
        out = np.zeros((len(marks), args.envs, cm.nq))
        for e in range(args.envs):
            od = mjo.OracleData(om)
            # the reset pose has the feet at distance -6e-17 from the floor (exactly touching, margin 0): whether those eight contacts
            # exist at step 0 is decided by the last bit, so EVERY variant starts from the float64 state 10 steps in (feet in contact)
            od.qpos[:], od.qvel[:], od.qacc_warmstart[:] = start[e]
            done = 0
            for k, t in enumerate(marks):
                od.rollout_random(t - done, seed=0, env=e, step0=T0 + done, scale=1.0)
                done = t
                out[k, e] = od.qpos
        return out

    T0 = 10
    start = []
    om0 = mjo.OracleModel(cm)
    for e in range(args.envs):
        od = mjo.OracleData(om0)
        od.rollout_random(T0, seed=0, env=e, scale=1.0)
        start.append((np.array(od.qpos), np.array(od.qvel), np.array(od.qacc_warmstart)))

    ref = run(0)
    print(f"humanoid, full-range random ctrl (seed 0), {args.envs} environments, from the float64 state at step {T0}: max |dqpos| vs the float64 oracle, median / 90th percentile")
    print(f"{'fp32-storage model':78s} " + " ".join(f"{'@' + str(t):>19s}" for t in marks))
    for name, mask in VARIANTS:
        err = np.abs(run(mask) - ref).max(axis=2)
        print(f"{name:78s} " + " ".join(f"{np.median(err[k]):9.2e}/{np.quantile(err[k], 0.9):9.2e}" for k in range(len(marks))))


if __name__ == "__main__":
    main()
