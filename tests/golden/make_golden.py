"""Generates tests/golden/*.npz from the float64 CPU oracle (SELF-ORACLE fixtures).

No golden vectors of the real `mujoco` library can be produced here (the wheel is absent from
/root/reference and from the image), so these fixtures pin the oracle against
regressions and give the GPU tests fixed targets; the independent anchors live in
tests/test_oracle_anchors.py.  If a session ever has an importable `mujoco`, point this same
generator at it to close "parity unpinned".

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from mujoco_template_amd.mjcf import compile_xml_path  # noqa: E402
from oracle import mjo  # noqa: E402

MODELS = {"pendulum": "models/pendulum.xml", "cartpole": "models/cartpole.xml", "humanoid": "models/humanoid.xml", "drone2": "models/drone2/scene.xml"}
SCALE = {"pendulum": 1.0, "cartpole": 0.005, "humanoid": 1.0, "drone2": 0.3}
NENV, NSTEP, SEED = 4, 40, 12345


def initial_states(cm, od, seed):
    rng = np.random.default_rng(seed)
    q = np.zeros((NENV, cm.nq)); v = np.zeros((NENV, cm.nv))
    for e in range(NENV):
        q[e] = od.integrate_pos(cm.qpos0, rng.normal(size=cm.nv) * 0.05, 1.0)
        v[e] = rng.normal(size=cm.nv) * 0.2
    return q, v


def main():
    for name, rel in MODELS.items():
        cm = compile_xml_path(os.path.join(ROOT, rel))
        om = mjo.OracleModel(cm)
        od = mjo.OracleData(om)
        q0, v0 = initial_states(cm, od, SEED)
        out = {"qpos0": q0, "qvel0": v0, "nstep": NSTEP, "seed": SEED, "scale": SCALE[name]}
        fwd = {k: [] for k in ("qacc", "qfrc_bias", "qfrc_passive", "qacc_smooth", "qfrc_constraint", "xpos", "subtree_com", "nefc")}
        A_all, B_all = [], []
        for e in range(NENV):
            od.reset(); od.qpos[:] = q0[e]; od.qvel[:] = v0[e]
            od.ctrl[:] = od.random_ctrl(SEED, e, 0, SCALE[name])
            od.forward()
            for k in fwd:
                fwd[k].append(od.counters()["nefc"] if k == "nefc" else np.array(getattr(od, k)))
            A, B = od.transition_fd(1e-6, True)
            A_all.append(A); B_all.append(B)
        for k, v in fwd.items():
            out["fwd_" + k] = np.array(v)
        out["A"], out["B"] = np.array(A_all), np.array(B_all)
        qT, vT = mjo.rollout_batch(om, NENV, NSTEP, seed=SEED, scale=SCALE[name], nthreads=1, qpos_init=q0, qvel_init=v0)
        out["qposT"], out["qvelT"] = qT, vT
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, "nefc", out["fwd_nefc"].tolist(), "|qT|", float(np.abs(qT).max()))


if __name__ == "__main__":
    main()
