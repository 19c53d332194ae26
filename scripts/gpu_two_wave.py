"""Two wavefronts per environment (k_step2) against the one-wave step kernel on small batches: bitwise equality and launch time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
spec = None if os.environ.get("TW_SPEC", "1") == "1" else False
for B in [int(x) for x in os.environ.get("TW_B", "1,64,256,512,768,1024,1100").split(",")]:
    res = {}
    for two in ("0", "policy"):
        if two == "0": os.environ["MJB_TWO_WAVE"] = "0"
        else: os.environ.pop("MJB_TWO_WAVE", None)
        sim = BatchSim(dm, B, dtype="float32", specialize=spec)
        sim.rollout(150, CTRL_RANDOM, seed=1); sim.sync()
        ts = []
        for r in range(5):
            t = time.perf_counter(); sim.rollout(100, CTRL_RANDOM, seed=1, step0=150 + 100 * r); sim.sync(); ts.append(time.perf_counter() - t)
        cn = sim.counters()
        res[two] = (min(ts), [sim.get(k) for k in ("qpos", "qvel", "qacc", "qacc_warmstart", "time", "xpos")] + [cn[k] for k in ("ncon", "nefc", "solver_niter")], sim.schedule_info())
        del sim
    same = all(np.array_equal(x, y) for x, y in zip(res["0"][1], res["policy"][1]))
    t1, t2 = res["0"][0], res["policy"][0]
    print(f"B={B:5d} specialised={spec is None}: one wave {t1*1e3:7.3f} ms per 100 steps ({B*100/t1/1e6:6.2f} M/s)   two waves {t2*1e3:7.3f} ms ({B*100/t2/1e6:6.2f} M/s, waves_per_env {res['policy'][2]['waves_per_env']})   x{t1/t2:.3f}   bitwise equal: {same}", flush=True)
