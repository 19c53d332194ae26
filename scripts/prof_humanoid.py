import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cm = compile_xml_path(os.path.join(ROOT, "models/humanoid.xml"))
sim = BatchSim(DeviceModel(cm), B, dtype="float32")
sim.rollout(20, CTRL_RANDOM, seed=1); sim.sync()
t = time.time(); sim.rollout(n, CTRL_RANDOM, seed=1, step0=20); sim.sync(); dt = time.time() - t
print(f"B={B} n={n}: {B*n/dt:.3e} env-steps/s")
