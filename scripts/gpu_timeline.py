"""Per-environment wave timeline of one fused launch (diagnostic kernel: MJB_SPEC_FLAGS=-DMJB_TIMELINE).
Shows when each environment's wave started / ended, on which XCD / CU / SIMD / wave slot, to explain why a launch of B = 2048 k
environments takes ~0.9 ms + 5.2 ms x k."""
import sys, os
os.environ["MJB_SPEC_FLAGS"] = (os.environ.get("MJB_SPEC_FLAGS", "") + " -DMJB_TIMELINE").strip()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
for B in [int(x) for x in (sys.argv[1:] or ["2048", "4096"])]:
  for sg in os.environ.get("TIMELINE_STAGGER", "0").split(","):
      os.environ["MJB_STAGGER"] = sg; print("MJB_STAGGER", sg)
      sim = BatchSim(dm, B, dtype="float32")
      NS = int(os.environ.get("TIMELINE_STEPS", "100"))
      sim.rollout(100, CTRL_RANDOM, seed=1); sim.sync()
      sim.rollout(NS, CTRL_RANDOM, seed=1, step0=100); sim.sync()
      t = sim.profile_env_get().astype(np.int64)
      t0 = t[:, 0].min()
      st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0; du = en - st          # microseconds
      hw = t[:, 2]; slot = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = t[:, 3] & 15; it = (t[:, 3] >> 8) & 0xFFFF; ne = (t[:, 3] >> 24) & 0xFFFFF; nc = (t[:, 3] >> 44) & 0xFFFFF
      print(f"B={B}: launch span {en.max()/1e3:.3f} ms; wave duration mean {du.mean()/1e3:.3f} ms  min {du.min()/1e3:.3f}  max {du.max()/1e3:.3f}  std {du.std()/1e3:.3f}")
      first = st < 50.0
      print(f"  waves started within 50 us: {first.sum()}  (duration mean {du[first].mean()/1e3:.3f} min {du[first].min()/1e3:.3f} max {du[first].max()/1e3:.3f});"
            f" later: {(~first).sum()}" + (f" (duration mean {du[~first].mean()/1e3:.3f} min {du[~first].min()/1e3:.3f} max {du[~first].max()/1e3:.3f}; start mean {st[~first].mean()/1e3:.3f} min {st[~first].min()/1e3:.3f} max {st[~first].max()/1e3:.3f})" if (~first).any() else ""))
      A = np.stack([np.ones(B), it, ne, nc], 1).astype(float); coef, *_ = np.linalg.lstsq(A[first], du[first], rcond=None); fit = A @ coef
      print(f"  own-work model (first round): duration ~ {coef[0]:.1f} + {coef[1]:.2f}*iters + {coef[2]:.3f}*nefc + {coef[3]:.3f}*ncon us; R^2 = {1 - ((du-fit)[first]**2).sum() / ((du[first]-du[first].mean())**2).sum():.3f};"
            f" per-step means: iters {it.mean()/NS:.2f} nefc {ne.mean()/NS:.1f} ncon {nc.mean()/NS:.2f}; corr(du, iters) {np.corrcoef(du[first], it[first])[0,1]:.3f} corr(du, nefc) {np.corrcoef(du[first], ne[first])[0,1]:.3f}")
      print("  end-time percentiles (ms):", " ".join(f"p{q}={np.percentile(en, q)/1e3:.3f}" for q in (1, 10, 50, 90, 99, 100)))
      for s in sorted(set(slot.tolist())):
          k = slot == s
          print(f"  wave slot {s}: n={k.sum():5d} duration mean {du[k].mean()/1e3:.3f} ms  (first-round only: {du[k & first].mean()/1e3:.3f})")
      for x in sorted(set(xcc.tolist())):
          k = (xcc == x) & first
          print(f"  XCD {x}: n={k.sum():4d} first-round duration mean {du[k].mean()/1e3:.3f} max {du[k].max()/1e3:.3f}; CUs used {len(set((se[k]*100+sh[k]*16+cu[k]).tolist()))}")
      # correlation of duration with the environment's own work: Newton iterations are not recorded per launch, use the SECOND measurement
      sim.rollout(NS, CTRL_RANDOM, seed=1, step0=100); sim.sync()                         # same steps again from the NEXT state (different work)
      t2 = sim.profile_env_get().astype(np.int64); du2 = (t2[:, 1] - t2[:, 0]) / 100.0
      f2 = (t2[:, 0] - t2[:, 0].min()) / 100.0 < 50.0
      both = first & f2
      print(f"  corr(duration launch A, duration launch B) over envs in the first round of both: {np.corrcoef(du[both], du2[both])[0,1]:.3f}")
      per_simd = {}
      key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
      for kk in set(key[first].tolist()):
          idx = np.where((key == kk) & first)[0]
          if len(idx) == 2:
              a, b = sorted(du[idx].tolist()); per_simd[kk] = (a, b)
      if per_simd:
          arr = np.array(list(per_simd.values()))
          print(f"  SIMDs holding exactly two first-round waves: {len(arr)}; faster of the pair mean {arr[:,0].mean()/1e3:.3f} ms, slower {arr[:,1].mean()/1e3:.3f} ms")
      del sim
