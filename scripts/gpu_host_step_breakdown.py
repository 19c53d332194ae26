"""Where does a host-driven Env.step at B = 4096 spend its time?  (controller in numpy, mjb_step_host, observation)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd import Env, ObservationSpec
from mujoco_template_amd import mj
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), obs_spec=ObservationSpec(as_dict=False), batch=B, dtype="float32")
d, m = env.data, env.model
rng = np.random.default_rng(0)
lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
T = {"ctrl (numpy uniform + write)": 0.0, "edit detection + step_host (H2D, kernel, pack, D2H)": 0.0, "obs extraction": 0.0}
N = 60
for it in range(N + 10):
    t0 = time.perf_counter()
    d.ctrl[:] = rng.uniform(lo, hi, size=d.ctrl.shape)
    t1 = time.perf_counter()
    mj.mj_step(m, d)
    t2 = time.perf_counter()
    obs = env.extractor(d) if hasattr(env, "extractor") else None
    t3 = time.perf_counter()
    if it >= 10:
        T["ctrl (numpy uniform + write)"] += t1 - t0; T["edit detection + step_host (H2D, kernel, pack, D2H)"] += t2 - t1; T["obs extraction"] += t3 - t2
tot = sum(T.values())
print(f"B={B}: {tot/N*1e3:.3f} ms per host-driven step = {B*N/tot/1e6:.2f} M env-steps/s")
for k, v in T.items(): print(f"   {k:55s} {v/N*1e6:8.1f} us  {100*v/tot:5.1f} %")
# inside step_host: time the C call alone with an explicit mask
sim = d.sim
t = time.perf_counter()
for _ in range(N): sim.step_host(1, 4)          # ctrl edited
dt = (time.perf_counter() - t) / N
print(f"   BatchSim.step_host(1, ctrl) alone: {dt*1e6:.1f} us")
t = time.perf_counter()
for _ in range(N): d._edited_mask()
print(f"   edit detection (mjb_mirror_edited_mask: memcmp of the six fields with the library's shadow): {(time.perf_counter()-t)/N*1e6:.1f} us")
