import os, time, sys
sys.path.insert(0, '/root/repo')
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
x = torch.zeros(1, device="cuda:0")
for _ in range(5): dist.barrier(device_ids=[0]); torch.cuda.synchronize()
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t = time.perf_counter(); dist.barrier(device_ids=[0]); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("dist.barrier(device_ids) + synchronize, 1 nccl rank: median %.1f us  min %.1f" % (1e6 * sorted(ts)[25], 1e6 * min(ts)))
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t = time.perf_counter(); dist.all_reduce(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("all_reduce(1 float) + synchronize: median %.1f us  min %.1f" % (1e6 * sorted(ts)[25], 1e6 * min(ts)))
ts = []
for _ in range(50):
    t = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("synchronize alone (idle): median %.1f us" % (1e6 * sorted(ts)[25]))
dist.destroy_process_group()
