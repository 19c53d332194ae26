import sys; sys.path.insert(0, '/root/repo')
import numpy as np
import mujoco_template_amd as mt
from tests.conftest import MODELS
env = mt.Env.from_xml_path(MODELS["humanoid"], batch=4, dtype="float32", nconmax=1, nefcmax=8, controller=None)
print(env.data.sim.nconmax, env.data.sim.nefcmax)
for s in range(30):
    r = env.step(return_obs=False)
    cn = env.data.counters()
    if s % 5 == 0 or cn["con_dropped"].sum():
        print(s, cn["ncon"], cn["nefc"], cn["con_dropped"], cn["efc_dropped"], env.data._flags, r.info.keys(), env.data.qpos[:, 2])
