"""Workload for rocprofv3 on the finite-difference linearisation kernel (the per-model specialised k_fd, float64, `mjb_k_fd_spec`): BASELINE
config 4 (cart-pole B=512), then humanoid B=512.   rocprofv3 --kernel-trace --stats -d OUT -- python3 scripts/prof_fd.py        (program directly after --)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_template_amd._capi import CTRL_RANDOM, BatchSim, DeviceModel
from mujoco_template_amd.mjcf import compile_xml_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, B, scale in (("cartpole", 512, 0.01), ("humanoid", 512, 1.0)):     # the humanoid LAST: the summary takes the last 6 mjb_k_fd_spec launches
    sim = BatchSim(DeviceModel(compile_xml_path(os.path.join(ROOT, f"models/{name}.xml"))), B, dtype="float32")
    sim.rollout(100, CTRL_RANDOM, seed=1, ctrl_scale=scale)
    for _ in range(6):
        sim.transition_fd(1e-6, True)
    sim.sync()
    # the two other float64 helper kernels, for the kernel-trace: Jacobians (k_jac) and the device observation gather (k_obs)
    import torch
    sim.jac([1, 2, 3], [1, 1, 1])
    spec = sim.make_obs_spec(3, body_ids=[1])
    out = torch.empty((B, spec.dim), device="cuda", dtype=torch.float32)
    for _ in range(3):
        sim.obs_gather(spec, out.data_ptr())
    sim.sync()
