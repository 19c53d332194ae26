"""First GPU contact: forward / step parity of the HIP path against the oracle on all models."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import mjo
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM

np.set_printoptions(precision=5, suppress=True, linewidth=200)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def check(xml, dtype, B=8, nsteps=50, scale=1.0, lanes=0):
    cm = compile_xml_path(os.path.join(ROOT, xml))
    om = mjo.OracleModel(cm)
    dm = DeviceModel(cm)
    sim = BatchSim(dm, B, dtype=dtype, lanes=lanes)
    rng = np.random.default_rng(1)
    ods = [mjo.OracleData(om) for _ in range(B)]
    q = np.zeros((B, cm.nq)); v = np.zeros((B, cm.nv)); u = np.zeros((B, cm.nu))
    for e, od in enumerate(ods):
        q[e] = od.integrate_pos(cm.qpos0, rng.normal(size=cm.nv) * 0.05 * (e > 0), 1.0)
        v[e] = rng.normal(size=cm.nv) * 0.2 * (e > 0)
        u[e] = rng.uniform(-1, 1, size=cm.nu) * (e > 0)
        od.qpos[:] = q[e]; od.qvel[:] = v[e]; od.ctrl[:] = u[e]
    sim.set("qpos", q); sim.set("qvel", v); sim.set("ctrl", u)
    sim.debug_forward(); sim.sync()
    for od in ods: od.forward()
    qacc = sim.get("qacc")
    errs = {}
    for name in ["qM", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "qfrc_constraint"]:
        g = sim.debug_get(name)
        o = np.stack([getattr(od, name) for od in ods])
        errs[name] = np.abs(g - o).max() / (np.abs(o).max() + 1e-12)
    errs["qacc"] = np.abs(qacc - np.stack([od.qacc for od in ods])).max() / (np.abs(np.stack([od.qacc for od in ods])).max() + 1e-12)
    errs["xpos"] = np.abs(sim.get("xpos") - np.stack([od.xpos for od in ods])).max()
    cn = sim.counters()
    print(f"{xml} {dtype} lanes={sim.lanes} lds/env={sim.lds_bytes_per_env}B forward rel-err:", {k: float(f"{v:.2e}") for k, v in errs.items()},
          "nefc gpu", cn["nefc"].tolist(), "oracle", [od.counters()["nefc"] for od in ods])
    # teacher-forced single steps along the oracle trajectory + free-running drift
    max_tf = 0.0; max_free = 0.0
    sim_free = BatchSim(dm, B, dtype=dtype, lanes=lanes)
    sim_free.set("qpos", q); sim_free.set("qvel", v)
    for s in range(nsteps):
        uu = np.stack([od.random_ctrl(0, e, s, scale) for e, od in enumerate(ods)])
        sim.set("qpos", np.stack([od.qpos for od in ods])); sim.set("qvel", np.stack([od.qvel for od in ods]))
        sim.set("qacc_warmstart", np.stack([od.qacc_warmstart for od in ods])); sim.set("ctrl", uu)
        sim_free.set("ctrl", uu)
        for e, od in enumerate(ods):
            od.ctrl[:] = uu[e]; od.step()
        sim.step(1); sim_free.step(1)
        qo = np.stack([od.qpos for od in ods])
        max_tf = max(max_tf, np.abs(sim.get("qpos") - qo).max())
        max_free = max(max_free, np.abs(sim_free.get("qpos") - qo).max())
    cn = sim_free.counters()
    print(f"    {nsteps} steps: teacher-forced max|dqpos| {max_tf:.3e}  free-running drift {max_free:.3e}  dropped {cn['con_dropped'].sum()} {cn['efc_dropped'].sum()} bad {cn['warn_badqacc'].sum()}")

t0 = time.time()
for dtype in ("float64", "float32"):
    check("models/pendulum.xml", dtype, nsteps=200)
    check("models/cartpole.xml", dtype, nsteps=200, scale=0.005)
    check("models/drone2/scene.xml", dtype, nsteps=100, scale=0.3)
    check("models/humanoid.xml", dtype, nsteps=100)
print("total", time.time() - t0)
# quick throughput probe
cm = compile_xml_path(os.path.join(ROOT, "models/humanoid.xml"))
dm = DeviceModel(cm)
for B in (1024, 4096):
    sim = BatchSim(dm, B, dtype="float32")
    sim.rollout(10, CTRL_RANDOM, seed=1); sim.sync()
    t = time.time(); sim.rollout(200, CTRL_RANDOM, seed=1, step0=10); sim.sync(); dt = time.time() - t
    cn = sim.counters()
    print(f"humanoid B={B} fp32: {B*200/dt:.3e} env-steps/s ({dt*1e3/200:.3f} ms/step) mean nefc {cn['nefc'].mean():.1f} max {cn['nefc'].max()} dropped {cn['efc_dropped'].sum()} {cn['con_dropped'].sum()} bad {cn['warn_badqacc'].sum()+cn['warn_badqpos'].sum()+cn['warn_badqvel'].sum()}")
