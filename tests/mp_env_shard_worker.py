"""Worker of tests/test_gpu_env_api.py::test_sharded_env_gathers_the_global_observation_block (NOT a test module): one rank of a
torchrun job.  Creates the sharded Env through the public API, rolls out, gathers, and rank 0 writes what it got."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mujoco_template_amd as mt  # noqa: E402
from mujoco_template_amd.distributed import init_process_group, world  # noqa: E402


def main() -> None:
    out_dir, global_batch, backend = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, ws, _ = world()
    assert init_process_group(backend, single_rank=True)
    env = mt.Env.from_xml_path(os.path.join(ROOT, "models", "humanoid.xml"), obs_spec=mt.ObservationSpec(as_dict=False),
                               controller=mt.RandomCtrlController(seed=4), batch=global_batch, shard=True,
                               device=0 if backend == "gloo" else None)
    assert env.shard.world_size == ws and env.shard.rank == rank and env.data.batch == env.shard.count
    ring = env.rollout(12, obs_every=4, gather=True)                  # [3, GLOBAL, 55] on every rank
    now = env.observe_device(gather=True)                             # [GLOBAL, 55]
    assert tuple(ring.shape) == (3, global_batch, 55) and tuple(now.shape) == (global_batch, 55), (ring.shape, now.shape)
    np.save(os.path.join(out_dir, f"ring_{rank}.npy"), ring.cpu().numpy())
    np.save(os.path.join(out_dir, f"now_{rank}.npy"), now.cpu().numpy())
    with open(os.path.join(out_dir, f"info_{rank}.txt"), "w") as f:
        f.write(f"{env.shard.env0} {env.shard.count} {env.gather_collective}\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
