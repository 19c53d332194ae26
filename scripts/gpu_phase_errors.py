"""Which phase of the fp32 forward pass injects the per-step error?  Humanoid states sampled from an oracle rollout (in contact),
identical inputs to the fp32 kernel and the float64 oracle; absolute error of every intermediate + of qacc.
    python scripts/gpu_phase_errors.py [B] [T_sample]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mujoco_template_amd._capi import BatchSim, DeviceModel
from mujoco_template_amd.mjcf import compile_xml_path
from oracle import mjo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cm = compile_xml_path(os.path.join(ROOT, "models/humanoid.xml"))
om = mjo.OracleModel(cm)
dm = DeviceModel(cm)
ods = [mjo.OracleData(om) for _ in range(B)]
for e, od in enumerate(ods):
    od.qpos[2] += 0.01 * e / B
    for s in range(TS):
        od.ctrl[:] = od.random_ctrl(1, e, s, 1.0)
        od.step()
    od.ctrl[:] = od.random_ctrl(1, e, TS, 1.0)
q = np.stack([od.qpos.copy() for od in ods]); v = np.stack([od.qvel.copy() for od in ods]); u = np.stack([od.ctrl.copy() for od in ods])
ws = np.stack([od.qacc_warmstart.copy() for od in ods])
for od in ods:
    od.forward()
print(f"humanoid, {B} states after {TS} oracle steps of random ctrl; tolerance floor env = {os.environ.get('MJB_F32_TOL_FLOOR')}")
for dtype in ("float32", "float64"):
    sim = BatchSim(dm, B, dtype=dtype)
    sim.set("qpos", q); sim.set("qvel", v); sim.set("ctrl", u); sim.set("qacc_warmstart", ws)
    sim.debug_forward()
    print(f"--- {dtype}: nefc mean {sim.counters()['nefc'].mean():.1f} niter mean {sim.counters()['solver_niter'].mean():.2f}")
    for key in ("qM", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "qfrc_constraint"):
        ref = np.stack([getattr(od, key) for od in ods]); got = sim.debug_get(key).reshape(ref.shape)
        err = np.abs(got - ref)
        print(f"{key:16s} |ref| max {np.abs(ref).max():9.3e}  err max {err.max():9.3e} median-of-env-max {np.median(err.reshape(B, -1).max(1)):9.3e}")
    ref = np.stack([od.qacc for od in ods]); err = np.abs(sim.get("qacc") - ref)
    print(f"{'qacc':16s} |ref| max {np.abs(ref).max():9.3e}  err max {err.max():9.3e} median-of-env-max {np.median(err.max(1)):9.3e}  per-dof median {np.array2string(np.median(err, 0), precision=1)}")
    nv = cm.nv
    J = sim.debug_get("efc_J").reshape(B, sim.nefcmax, nv); ar = sim.debug_get("efc_aref"); D = sim.debug_get("efc_D"); pos = sim.debug_get("efc_pos"); fr = sim.debug_get("efc_force")
    eJ, ea, eD, ep, ef = [], [], [], [], []
    for e, od in enumerate(ods):
        n = od.counters()["nefc"]
        if n:
            eJ.append(np.abs(J[e, :n] - od.efc_J.reshape(n, nv)).max()); ea.append(np.abs(ar[e, :n] - od.efc_aref).max())
            eD.append((np.abs(D[e, :n] - od.efc_D) / od.efc_D).max()); ep.append(np.abs(pos[e, :n] - od.efc_pos).max())
            ef.append(np.abs(fr[e, :n] - od.efc_force).max())
    for nm, x in (("efc_J", eJ), ("efc_aref", ea), ("efc_D (rel)", eD), ("efc_pos", ep), ("efc_force", ef)):
        print(f"{nm:16s} err max {max(x):9.3e} median-of-env-max {np.median(x):9.3e}")
