"""Host-driven single-environment step: completion by polling the environment's completion word in the pinned mirror block vs waiting for the
end-of-kernel signal (MJB_HOST_POLL=0).  Run twice (the switch is read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd import Env, ObservationSpec, mj
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in (1, 8):
    env = Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), obs_spec=ObservationSpec(as_dict=False), batch=B, dtype="float32")
    d, m, sim = env.data, env.model, env.data.sim
    rng = np.random.default_rng(0)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    for _ in range(200): sim.step_host(1, 4)
    N = 2000
    t = time.perf_counter()
    for _ in range(N): sim.step_host(1, 4)
    a = (time.perf_counter() - t) / N
    t = time.perf_counter()
    for _ in range(N):
        d.ctrl[...] = rng.uniform(lo, hi, size=np.shape(d.ctrl))
        mj.mj_step(m, d)
    b = (time.perf_counter() - t) / N
    print(f"MJB_HOST_POLL={os.environ.get('MJB_HOST_POLL', '1')} B={B}: mjb_step_host {a * 1e6:.1f} us, numpy ctrl + mj_step {b * 1e6:.1f} us per step, qpos[2] {np.ravel(d.qpos)[2]:.6f}")
