"""Ticket map vs static map on the packed models at batches beyond the resident slots (does the policy hold there?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, B, scale in (("cartpole", 65536, 0.01), ("drone2/scene", 16384, 0.3), ("pendulum", 131072, 1.0)):
    dm = DeviceModel(compile_xml_path(os.path.join(ROOT, f"models/{name}.xml")))
    row = {}
    for mode in ("0", None):
        if mode is None: os.environ.pop("MJB_CHUNK_STEPS", None)
        else: os.environ["MJB_CHUNK_STEPS"] = mode
        sim = BatchSim(dm, B, dtype="float32")
        sim.rollout(50, CTRL_RANDOM, seed=1, ctrl_scale=scale); sim.sync()
        ts = []
        for r in range(5):
            t = time.perf_counter(); sim.rollout(100, CTRL_RANDOM, seed=1, step0=50 + 100 * r, ctrl_scale=scale); sim.sync(); ts.append(time.perf_counter() - t)
        row[mode] = (min(ts), sim.schedule_info(), sim.get("qpos"))
        del sim
    a, b = row["0"], row[None]
    print(f"{name:14s} B={B}: static {a[0]*1e3:.3f} ms ({B*100/a[0]/1e6:.1f} M/s) | policy [{b[1]['map']}, chunk {b[1]['chunk_steps']}, blocks {b[1]['env_blocks']}, slots {b[1]['resident_slots']}] {b[0]*1e3:.3f} ms ({B*100/b[0]/1e6:.1f} M/s) x{a[0]/b[0]:.3f} bitwise {np.array_equal(a[2], b[2])}", flush=True)
