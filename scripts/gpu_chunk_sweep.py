"""Per-launch overhead of the fused step kernel: humanoid B=4096, HIP-event time of launches of 1..100 steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mujoco_template_amd import Env, ObservationSpec, RandomCtrlController

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), obs_spec=ObservationSpec(as_dict=False), controller=RandomCtrlController(seed=0),
                        batch=B, dtype="float32")
env.data.sim.use_torch_stream()
env.rollout(200, obs_every=200)
torch.cuda.synchronize()
for n in [int(x) for x in os.environ.get("SWEEP_STEPS", "1,2,4,10,20,50,100").split(",")]:
    reps = max(5, 200 // n)
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(n, obs_every=n); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    med = ms[len(ms) // 2]
    print(f"steps/launch {n:4d}: median launch {med:.4f} ms = {med / n * 1e3:.1f} us/step -> {B * n / med / 1e3:.2f} M env-steps/s (min {ms[0]:.4f} max {ms[-1]:.4f}, {reps} launches)", flush=True)
