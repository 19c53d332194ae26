"""Where does the occasional 35-54 ms reading of ONE event pair in a burst of short launches come from?
Host time of every call between the two event records of a pair, for a burst of 12 launches of 20 steps with no synchronisation in
between - the shape of bench.py's roofline launches - once with FRESH torch events (created inside the burst, as bench.py did) and once
with events created and recorded once before the burst."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_template_amd import Env, ObservationSpec, RandomCtrlController
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), obs_spec=ObservationSpec(as_dict=False), controller=RandomCtrlController(seed=0), batch=4096, dtype="float32")
env.data.sim.use_torch_stream()
env.rollout(5, obs_every=5); torch.cuda.synchronize()
for mode in ("fresh events", "pre-recorded events", "fresh events again"):
    N = 12
    pool = []
    if mode.startswith("pre"):
        pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
        for a, b in pool:
            a.record(); b.record()
        torch.cuda.synchronize()
    rows, ev = [], []
    t_burst = time.perf_counter()
    for k in range(N):
        t0 = time.perf_counter()
        e0, e1 = pool[k] if pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e0.record()
        t1 = time.perf_counter()
        env.rollout(20, obs_every=20)
        t2 = time.perf_counter()
        e1.record()
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2)); ev.append((e0, e1))
    t_enq = time.perf_counter() - t_burst
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_burst
    ms = [a.elapsed_time(b) for a, b in ev]
    print(f"{mode}: burst enqueued in {t_enq * 1e3:.2f} ms, complete in {t_all * 1e3:.2f} ms")
    print("  event-pair ms  :", " ".join(f"{x:7.3f}" for x in ms))
    print("  host us e0.rec :", " ".join(f"{r[0] * 1e6:7.0f}" for r in rows))
    print("  host us rollout:", " ".join(f"{r[1] * 1e6:7.0f}" for r in rows))
    print("  host us e1.rec :", " ".join(f"{r[2] * 1e6:7.0f}" for r in rows))
