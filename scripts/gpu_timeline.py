"""Per-workgroup timeline of one fused k_step launch (diagnostic kernel: MJB_SPEC_FLAGS=-DMJB_TIMELINE records, per workgroup, start
and end on the 100 MHz clock, HW_ID / XCC_ID, tickets served and the Newton iterations / constraint rows / contacts it worked through).

Explains why a static map (one workgroup per environment for all steps) of B = 2 x resident slots loses 10-30 % to its tails, and
shows what the priority hand-over (MJB_FAIR_BIT) and the ticket map (MJB_CHUNK_STEPS) do about it.
usage: TIMELINE_STEPS=20 python scripts/gpu_timeline.py 2048 4096"""
import sys, os
os.environ["MJB_SPEC_FLAGS"] = (os.environ.get("MJB_SPEC_FLAGS", "") + " -DMJB_TIMELINE").strip()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
NS = int(os.environ.get("TIMELINE_STEPS", "100"))
CONFIGS = [("static map, hardware age order", "0", "0"), ("static map, priority hand-over (fair_bit 15)", "0", "15"), ("default policy", None, None)]
for B in [int(x) for x in (sys.argv[1:] or ["2048", "4096"])]:
    for label, chunk, fair in CONFIGS:
        for k, v in (("MJB_CHUNK_STEPS", chunk), ("MJB_FAIR_BIT", fair)):
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        sim = BatchSim(dm, B, dtype="float32")
        sim.rollout(100, CTRL_RANDOM, seed=1); sim.sync()
        sim.rollout(NS, CTRL_RANDOM, seed=1, step0=100); sim.sync()
        info = sim.schedule_info()
        nwg = info["resident_slots"] if info["map"] == "tickets" and 0 < info["resident_slots"] < info["env_blocks"] else info["env_blocks"]
        t = sim.profile_env_get().astype(np.int64)[:nwg]
        t0 = t[:, 0].min()
        st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0; du = en - st          # microseconds
        hw = t[:, 2] & 0xFFFF; slot = hw & 15; tickets = (t[:, 2] >> 16) & 0xFFFF
        it = (t[:, 3] >> 8) & 0xFFFF; ne = (t[:, 3] >> 24) & 0xFFFFF; nc = (t[:, 3] >> 44) & 0xFFFFF
        busy = du.sum() / (min(nwg, info["resident_slots"] or nwg) * en.max())
        print(f"B={B} steps/launch={NS} [{label}] map={info['map']} chunk_steps={info['chunk_steps']} fair_bit={info['fair_bit']}: launch span {en.max()/1e3:.3f} ms,"
              f" {nwg} workgroups, duration mean {du.mean()/1e3:.3f} min {du.min()/1e3:.3f} max {du.max()/1e3:.3f} std {du.std()/1e3:.3f} ms; slot-time in use {100*busy:.1f} %")
        print("    end-time percentiles (ms):", " ".join(f"p{q}={np.percentile(en, q)/1e3:.3f}" for q in (1, 10, 50, 90, 99, 100)))
        for s in sorted(set(slot.tolist())):
            kk = slot == s
            print(f"    wave slot {s} of its SIMD: n={kk.sum():5d} duration mean {du[kk].mean()/1e3:.3f} ms, tickets served mean {tickets[kk].mean():.1f}, Newton iterations worked mean {it[kk].mean():.0f}")
        if info["map"] == "static":
            first = st < 50.0
            A = np.stack([np.ones(nwg), it, ne, nc], 1).astype(float); coef, *_ = np.linalg.lstsq(A[first], du[first], rcond=None); fit = A @ coef
            r2 = 1 - ((du - fit)[first] ** 2).sum() / ((du[first] - du[first].mean()) ** 2).sum()
            print(f"    own-work model (workgroups started at t = 0): duration ~ {coef[0]:.0f} + {coef[1]:.2f} x Newton iterations + {coef[2]:.3f} x rows + {coef[3]:.3f} x contacts us, R^2 = {r2:.3f};"
                  f" per step: {it.mean()/NS:.2f} iterations, {ne.mean()/NS:.1f} rows, {nc.mean()/NS:.2f} contacts")
        del sim
