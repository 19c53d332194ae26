#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched step/rollout path (BASELINE.json).

Workload (config[2] of BASELINE.json, the one the metric is quoted on): the reference's
``examples/humanoid`` model, random-ctrl rollout, GLOBAL batch 4096 sharded over the ranks
(one process per GPU, no collective on the stepping path), fp32 state, flat observation
(qpos ‖ qvel, the default ObservationSpec of Env.from_xml_path) all-gathered over RCCL once
per fused chunk.  One bench "step" = one simulation step of the whole global batch.

    python bench.py --gpus N --steps K --warmup W

N > 1 from a plain shell: this process starts the N ranks itself as a child ``torch.distributed.run`` (before importing
torch or touching HIP) and forwards rank 0's line; under torchrun (RANK / WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line (rank 0).  ``value`` = env-steps / s of the whole job with inputs resident
in HBM; ``roofline`` prices the dominant kernel (k_step) against HBM with the algorithmic
bytes of SURVEY.md §8(d); ``cpu_baseline`` times the float64 oracle (a port: the reference's
own CPU path needs the absent ``mujoco`` wheel) on the host cores for a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)


def launcher_needed(gpus: int, environ) -> bool:
    """True when this process must start the ranks itself: more than one GPU asked for and no torchrun environment present."""
    return int(gpus) > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def launcher_command(gpus: int, argv: list[str], port: int) -> list[str]:
    """The child command: ``torch.distributed.run`` (one rank per GPU) re-running THIS file with the same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def visible_gpu_count() -> int:
    """Number of AMD GPUs the kernel driver exposes, counted WITHOUT touching HIP (the launcher must stay GPU-free):
    KFD topology nodes with a non-zero SIMD count, narrowed by HIP/ROCR_VISIBLE_DEVICES when set.  -1 = unknown."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([t for t in v.split(",") if t.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(base):
        return -1
    n = 0
    for node in os.listdir(base):
        try:
            props = open(os.path.join(base, node, "properties")).read()
        except OSError:
            return -1           # topology not readable by this user: let the ranks themselves report a missing device
        for line in props.splitlines():
            if line.startswith("simd_count") and int(line.split()[1]) > 0:
                n += 1
    return n


def launch_ranks(gpus: int, argv: list[str], all_ranks_device0: bool = False) -> int:
    """Run the N ranks as a child ``torch.distributed.run`` and forward rank 0's JSON line; returns the exit code."""
    import socket
    import subprocess

    if not all_ranks_device0:
        have = visible_gpu_count()
        if 0 <= have < gpus:
            print(f"bench.py: --gpus {gpus} but only {have} GPU(s) are visible on this machine", file=sys.stderr)
            return 2
    with socket.socket() as s:          # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(launcher_command(gpus, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X fp32 vector (non-matrix) peak: 256 CUs x 4 SIMDs x 32 lanes/cycle x 2 (FMA) x 2.4 GHz (MI355X_MICROARCH.md)


def _event_pool(torch, n: int):
    """``n`` pairs of timing events, created AND recorded once before anything is timed: the first use of a HIP event allocates its
    signal, which must not happen between two records of the timed region (scripts/gpu_event_probe.py)."""
    pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in pool:
        e0.record(); e1.record()
    torch.cuda.synchronize()
    return pool


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--global-batch", type=int, default=4096)
    ap.add_argument("--chunk", type=int, default=1000, help="simulation steps fused per kernel launch (default: the whole 1000-step rollout in ONE launch)")
    ap.add_argument("--obs-every", type=int, default=100, help="an observation row (ObservationExtractor output) is written in-kernel every this many steps and all-gathered per launch")
    ap.add_argument("--weak", action="store_true", help="fixed per-GPU batch (= --global-batch per rank) instead of sharding it")
    ap.add_argument("--model", default="humanoid", choices=["humanoid", "cartpole", "drone2", "pendulum"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-loop", action="store_true", help="skip the host-driven Env.passive measurement (Python controller in the loop)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs for BASELINE configs 2, 4, 5 and the 100-step-launch continuity figure")
    ap.add_argument("--nefcmax", type=int, default=0)
    ap.add_argument("--nconmax", type=int, default=0)
    ap.add_argument("--tolerance", type=float, default=0.0, help="solver tolerance (model.opt.tolerance); 0 = the model's own (experiments)")
    ap.add_argument("--no-specialize", action="store_true", help="use the generic step kernel instead of the per-model specialised one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the multi-process path (observations staged through the host)")
    ap.add_argument("--collective", default="auto", choices=["auto", "rccl", "torch"], help="observation all-gather: the library's mjb_allgather_obs on its own RCCL communicator (auto: when the backend is nccl) or torch.distributed")
    ap.add_argument("--all-ranks-device0", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.chunk < 1 or args.obs_every < 1:
        raise SystemExit("bench.py: --steps / --chunk / --obs-every must be >= 1 and --warmup >= 0")

    if launcher_needed(args.gpus, os.environ):
        # `python bench.py --gpus N` from a plain shell: start the N ranks as CHILD processes before this process imports torch or
        # makes any HIP call (never os.exec*, never a relaunch from a process that has touched the GPU); forward rank 0's line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], all_ranks_device0=args.all_ranks_device0))

    import numpy as np
    import torch
    import torch.distributed as dist

    from mujoco_template_amd import Env, ObservationSpec, RandomCtrlController
    from mujoco_template_amd.distributed import init_process_group, world

    rank, ws, local = world()
    if ws != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: start it as `python bench.py --gpus N` (it launches its own "
                         "ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus")
    if not args.all_ranks_device0 and torch.cuda.device_count() < (local + 1 if ws > 1 else 1):
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local} but only {torch.cuda.device_count()} device(s) are visible")
    if args.all_ranks_device0 and args.backend == "nccl":
        raise SystemExit("--all-ranks-device0 needs --backend gloo (RCCL refuses two ranks on one GPU)")
    device = 0 if (ws == 1 or args.all_ranks_device0) else local
    torch.cuda.set_device(device)
    # under torchrun with ONE rank the process group is still set up and every collective below is issued (one-rank communicator):
    # the one-GPU rehearsal of the nccl path; `python bench.py` from a plain shell (no RANK) runs without torch.distributed as before
    single = ws == 1 and "RANK" in os.environ
    distributed = init_process_group(args.backend, single_rank=single) if (ws > 1 or single) else False

    xml = {"humanoid": "models/humanoid.xml", "cartpole": "models/cartpole.xml", "drone2": "models/drone2/scene.xml",
           "pendulum": "models/pendulum.xml"}[args.model]
    scale = {"humanoid": 1.0, "cartpole": 0.005, "drone2": 0.3, "pendulum": 1.0}[args.model]
    global_batch = args.global_batch * ws if args.weak else args.global_batch

    def make_env(gb: int):
        """The sharded environment through the public surface (Env.from_xml_path(..., batch=GLOBAL, shard=True)): this rank's block of
        the global batch on its GPU, random streams keyed by the global environment index, gather = the path's one collective."""
        e = Env.from_xml_path(os.path.join(ROOT, xml), obs_spec=ObservationSpec(as_dict=False), controller=RandomCtrlController(seed=0, scale=scale),
                              batch=gb, shard=True, dtype="float32", device=device, nefcmax=args.nefcmax, nconmax=args.nconmax,
                              specialize=False if args.no_specialize else None, collective=args.collective)
        if args.tolerance > 0:
            e.model.opt.tolerance = args.tolerance
        e.data.sim.use_torch_stream()
        return e

    env = make_env(global_batch)
    env0, count = env.shard.env0, env.shard.count
    sim = env.data.sim
    obs_dim = env.extractor.obs_dim
    nq, nv, nu = env.model.nq, env.model.nv, env.model.nu
    chunk = max(1, min(args.chunk, args.steps))               # steps fused per launch
    nlaunch = (args.steps + chunk - 1) // chunk
    pool = _event_pool(torch, max(nlaunch, 6) + 6)

    def run(e, nsteps: int) -> None:
        """untimed launches (warm-up, the weak-scaling companion): fused rollout + in-kernel observation ring + its all-gather"""
        done = 0
        while done < nsteps:
            n = min(chunk, nsteps - done)
            e.rollout(n, obs_every=min(args.obs_every, n), gather=True)
            done += n

    # the same with the kernel bracketed by a pair of HIP events (pre-created); the gather is issued right behind the closing event
    def run_timed(e, nsteps: int, events: list, per_launch: int | None = None) -> None:
        done = 0
        c = per_launch or chunk
        while done < nsteps:
            n = min(c, nsteps - done)
            e0, e1 = pool[len(events)]
            e0.record()
            obs = e.rollout(n, obs_every=min(args.obs_every, n))
            e1.record()
            events.append((e0, e1, n))
            e.gather_observations(obs)
            done += n

    def barrier() -> None:
        if distributed:
            dist.barrier(device_ids=[device]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(env, args.warmup)
    else:
        env.gather_observations(env.observe_device())          # the communicator is set up outside the timed region in any case
    barrier()
    events: list = []
    t0 = time.perf_counter()
    run_timed(env, args.steps, events)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], device=f"cuda:{device}", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    flags = sim.engine_flags()                                  # waits for the stream; bit 3 (hand-over timed out) has already raised in barrier()

    # companion figure for N > 1 (all ranks take part): the SAME per-GPU work as the N = 1 run (per-GPU batch = the global batch of
    # BASELINE's config), i.e. the weak-scaling point next to the strong-scaling headline.  Never `value`.
    weak_companion = None
    if distributed and not args.weak and ws > 1:
        wenv = make_env(args.global_batch * ws)
        wenv._comm, wenv._collective = env._comm, env._collective      # one RCCL communicator per process: the companion gathers on the headline run's
        run(wenv, max(args.warmup, chunk))
        barrier()
        tw = time.perf_counter()
        run(wenv, args.steps)
        barrier()
        welapsed = torch.tensor([time.perf_counter() - tw], device=f"cuda:{device}", dtype=torch.float64)
        dist.all_reduce(welapsed, op=dist.ReduceOp.MAX)
        weak_companion = {"value": args.global_batch * ws * args.steps / float(welapsed.item()), "unit": "env-steps/s", "scaling": "weak",
                          "per_gpu_batch": args.global_batch, "global_batch": args.global_batch * ws, "ms_per_step": float(welapsed.item()) * 1e3 / args.steps}
        del wenv
    in_region = len(events)
    # the roofline's launch duration is an average over >= 5 launches: when the timed region was fewer (a short --steps run is ONE
    # fused launch), more launches of the same length follow it here - they count for `roofline` only, never for `value`
    while len(events) < 5:
        run_timed(env, events[0][2], events)
    torch.cuda.synchronize()
    counters = env.data.counters()
    kernel_ms = [e0.elapsed_time(e1) for e0, e1, _ in events]
    steps_per_launch = [n for _, _, n in events]
    all_ms = [round(float(t), 3) for t in kernel_ms]
    med_ms = float(np.median(kernel_ms))
    slow = [k for k, t in enumerate(kernel_ms) if t > 3.0 * med_ms]      # never filtered out: reported (round 2 saw one 38 ms reading among 2 ms launches on one box)
    if slow and rank == 0:
        print(f"bench.py: WARNING launch(es) {slow} read more than 3x the median launch time: {all_ms}", file=sys.stderr)
    ranks_seen = dist.get_world_size() if distributed else 1
    shards = [[env0, count]]
    if distributed:
        mine = torch.tensor([env0, count], device=f"cuda:{device}" if args.backend == "nccl" else "cpu", dtype=torch.int64)
        got = [torch.zeros_like(mine) for _ in range(ws)]
        dist.all_gather(got, mine)
        shards = [[int(g[0]), int(g[1])] for g in got]

    if rank == 0:
        value = global_batch * args.steps / elapsed
        # algorithmic bytes per env-step (fp32): state read+write, ctrl, warm-start read+write, + obs when gathered
        bytes_step = 4 * ((nq + nv) * 2 + nu + 2 * nv)
        bytes_obs = 4 * obs_dim
        avg_ms = float(np.mean(kernel_ms))                        # plain mean over every timed launch
        avg_steps = float(np.mean(steps_per_launch))
        obs_rows = max(1.0, avg_steps // min(args.obs_every, max(1, int(avg_steps))))
        launch_bytes = count * (bytes_step * avg_steps + bytes_obs * obs_rows)   # obs rows per env per launch: steps // obs_every
        achieved = launch_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC profiles (separate --pmc passes, profiles/traffic.json): only a record measured
        # with the SAME model, batch and steps per launch as the launches timed here is quoted; anything else is null
        traffic = traffic_source = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof) and ws == 1 and len(set(steps_per_launch)) == 1:
            recs = json.load(open(prof))
            for rec in recs if isinstance(recs, list) else [recs]:
                if (rec.get("model") == args.model and rec.get("global_batch") == global_batch
                        and rec.get("launch_steps", rec.get("chunk")) == steps_per_launch[0]):
                    traffic, traffic_source = rec.get("traffic_bytes_per_launch"), rec.get("source")
        shape = (f"{nlaunch} fused launch(es) of {chunk} steps per rank, an observation row written in-kernel every {min(args.obs_every, chunk)} steps, "
                 f"ONE all-gather of the [{max(1, chunk // min(args.obs_every, chunk))}, batch, {obs_dim}] ring per launch")
        out = {
            "metric": "env-steps/sec (whole node), humanoid batch=4096 random-ctrl rollout" if args.model == "humanoid"
                      else f"env-steps/sec (whole node), {args.model} random-ctrl rollout",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": ws,
            "ranks_seen": ranks_seen,
            "shards_env0_count": shards,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "wall_ms_per_1000_step_rollout": elapsed * 1e3 / args.steps * 1000.0,
            "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (qpos0 start, Philox uniform random ctrl over ctrlrange, seed 0)",
            "config": {"workload": f"examples/{args.model} random-ctrl rollout (BASELINE.json configs[2])" if args.model == "humanoid"
                       else f"examples/{args.model} random-ctrl rollout",
                       "launch_and_communication_shape": shape,
                       "global_batch": global_batch, "per_gpu_batch": count, "rollout_steps": args.steps,
                       "fused_steps_per_launch": chunk, "obs_every": args.obs_every, "obs_dim": obs_dim, "parallelism": f"env-shard x{ws}",
                       "sharding": "Env.from_xml_path(..., batch=GLOBAL, shard=True) + Env.rollout / gather_observations", "gather_collective": env.gather_collective,
                       "lanes_per_env": sim.lanes, "lds_bytes_per_env": sim.lds_bytes_per_env,
                       "nefcmax": sim.nefcmax, "nconmax": sim.nconmax, "specialized_kernel": bool(sim.specialized),
                       "work_schedule": sim.schedule_info()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": ("mjb_k_step_spec (k_step<float,float,%d> with the model's sizes/offsets folded in)" if sim.specialized else "mjb::k_step<float,float,%d>") % sim.lanes,
                         "traffic_source": traffic_source, "launch_steps": avg_steps, "launches_timed": len(kernel_ms), "launches_slower_than_3x_median": len(slow), "launch_ms_all": all_ms, "launches_in_timed_region": in_region,
                         "avg_launch_ms": avg_ms, "median_launch_ms": med_ms, "algorithmic_bytes_per_env_step": bytes_step, "obs_bytes_per_env": bytes_obs,
                         "note": "fused step is VALU/LDS-latency bound by construction (SURVEY.md §8d): see flop_roofline / issue_roofline and DESIGN.md §5"},
            "solver": {"mean_nefc_last_step": float(counters["nefc"].mean()), "max_nefc_last_step": int(counters["nefc"].max()),
                       "mean_newton_iters_last_step": float(counters["solver_niter"].mean()),
                       "dropped_contacts": int(counters["con_dropped"].sum()), "dropped_rows": int(counters["efc_dropped"].sum()),
                       "bad_state_resets": int(counters["warn_badqpos"].sum() + counters["warn_badqvel"].sum() + counters["warn_badqacc"].sum()),
                       "engine_flags": flags, "engine_flags_meaning": "bit 0 contacts dropped, 1 rows dropped, 2 bad-state reset, 3 ticket hand-over timed out (= the run fails)"},
        }
        env_steps_per_s_kernel = count * avg_steps / (avg_ms * 1e-3)          # this rank's kernel-time rate
        # the yardstick SURVEY.md §8(d) asks for: ALGORITHMIC flops per env-step (the float64 oracle, instrumented build: scripts/flop_count.py)
        # x the rate of the dominant kernel, against the fp32 vector peak; beside it what the hardware counters say the kernel executes
        fj, pj = os.path.join(ROOT, "profiles", "r03_flops_per_env_step.json"), os.path.join(ROOT, "profiles", "r03_phase_table.json")
        if args.model == "humanoid" and os.path.exists(fj):
            fl = json.load(open(fj))
            fr = {"bound": "fp32-vector", "flops_per_env_step": fl["flops_per_env_step"], "achieved": fl["flops_per_env_step"] * env_steps_per_s_kernel / 1e12,
                  "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "source": "profiles/r03_flops_per_env_step.json (oracle/mjo.c compiled with a counting double; 256 envs x 1000 steps of this workload)"}
            fr["frac"] = fr["achieved"] / fr["peak"]
            if sim.specialized and os.path.exists(pj):
                wk = json.load(open(pj))["whole_kernel"]
                ex = wk["executed_valu_flops"] + wk["executed_mfma_flops"]
                fr.update({"executed_fp32_flops_per_env_step": ex, "executed_achieved": ex * env_steps_per_s_kernel / 1e12, "useful_fraction_of_executed": fl["flops_per_env_step"] / ex,
                           "mean_active_lanes_per_valu_instruction": wk["mean_active_lanes"], "valu_wave_instructions_per_env_step": wk["valu_insts"],
                           "fp32_arithmetic_share_of_valu_instructions": wk["fp32_insts"] / wk["valu_insts"],
                           "executed_source": "profiles/r03_phase_table.json (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 x SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU lanes + 512 x MFMA MOPS, separate --pmc passes)"})
            out["flop_roofline"] = fr
        # what actually bounds this kernel is instruction issue.  VALU wave-instructions per env-step from the committed PMC
        # pass of the same kernel; peak = one wave64 VALU instruction per 2 cycles per SIMD-32 (MI355X_MICROARCH.md), 1024 SIMDs, 2.4 GHz
        if args.model == "humanoid" and sim.specialized and os.path.exists(pj):
            per_step = float(json.load(open(pj))["whole_kernel"]["valu_insts"])
            peak_issue = 1024 * 2.4e9 / 2.0
            rate = per_step * env_steps_per_s_kernel
            out["issue_roofline"] = {"bound": "valu-issue", "achieved": rate, "peak": peak_issue, "unit": "wave-instructions/s", "frac": rate / peak_issue,
                                     "valu_wave_instructions_per_env_step": per_step, "source": "profiles/r03_phase_table.json (SQ_INSTS_VALU, separate --pmc pass)",
                                     "note": "one wave alone issues a VALU instruction every 4 cycles, so two resident waves per SIMD can reach 1.0 only with no waits at all"}
        if weak_companion is not None:
            out["weak_scaling_companion"] = weak_companion
        if ws == 1 and not args.no_other_configs:
            # continuity with rounds 1-2 (ten 100-step launches, an all-gather each): same workload, the older launch shape
            pool.extend(_event_pool(torch, 4))
            base = len(events)
            events_c = list(events)                                 # the pool index continues behind the launches above
            tcs = time.perf_counter()
            run_timed(env, 300, events_c, per_launch=100)
            torch.cuda.synchronize()
            tce = time.perf_counter() - tcs
            out["continuity_100_step_launches"] = {"value": count * 300 / tce, "unit": "env-steps/s", "launch_ms": [round(float(a.elapsed_time(b)), 3) for a, b, _ in events_c[base:]],
                                                   "shape": "3 launches of 100 steps, an observation row + all-gather per launch (the bench line's shape of rounds 1-2)"}
            out["other_configs"] = other_configs(device)
        if ws == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(os.path.join(ROOT, xml), scale)
        if ws == 1 and not args.no_host_loop:
            out["host_loop"] = host_loop(os.path.join(ROOT, xml), scale, device, global_batch)
        print(json.dumps(out), flush=True)
    if distributed:
        barrier()
        dist.destroy_process_group()
    if flags & 8:
        raise SystemExit(3)


def other_configs(device: int) -> dict:
    """Short, bounded legs for the other GPU configurations of BASELINE.json (2: cart-pole rollout B = 1024; 4: cart-pole batched
    mjd_transitionFD B = 512; 5: drone2 B = 2048 with the sites + bodies observation written EVERY step), <= ~2 s of GPU time in all.
    Same engine, same public API; the rocprof summaries of these kernels are under profiles/ (r03_other_configs_*)."""
    import numpy as np
    import torch

    from mujoco_template_amd import Env, ObservationSpec, RandomCtrlController
    from mujoco_template_amd.mjcf import compile_xml_path
    from oracle import mjo                                       # CPU port beside config 4 (checker-as-baseline, like cpu_baseline)

    out: dict = {}
    pool = _event_pool(torch, 8)

    def timed_rollouts(env, nsteps: int, obs_every: int, reps: int):
        env.data.sim.use_torch_stream()
        env.rollout(nsteps, obs_every=obs_every, gather=True)
        torch.cuda.synchronize()
        ms = []
        t = time.perf_counter()
        for k in range(reps):
            e0, e1 = pool[k]
            e0.record()
            obs = env.rollout(nsteps, obs_every=obs_every)
            e1.record()
            env.gather_observations(obs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        ms = [float(pool[k][0].elapsed_time(pool[k][1])) for k in range(reps)]
        return dt, ms

    # config 2: cart-pole, random ctrl (|ctrl| <= 1), B = 1024, 1000-step fused launches, an observation row every 100 steps
    env = Env.from_xml_path(os.path.join(ROOT, "models/cartpole.xml"), obs_spec=ObservationSpec(as_dict=False), controller=RandomCtrlController(seed=0, scale=0.005),
                            batch=1024, shard=True, dtype="float32", device=device)
    dt, ms = timed_rollouts(env, 1000, 100, 3)
    sim = env.data.sim
    bytes_step = 4 * ((env.model.nq + env.model.nv) * 2 + env.model.nu + 2 * env.model.nv)
    out["config2_cartpole_rollout_B1024"] = {"value": 1024 * 3000 / dt, "unit": "env-steps/s", "launch_ms": [round(x, 3) for x in ms], "steps_per_launch": 1000, "lanes_per_env": sim.lanes,
                                             "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": bytes_step, "achieved": 1024 * 1000 * bytes_step / (np.mean(ms) * 1e-3) / 1e9,
                                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 1024 * 1000 * bytes_step / (np.mean(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                             "note": "1024 environments x 8 lanes = 128 waves: a small fraction of the chip; launch latency of one wave's 1000 steps"}
    # config 4: cart-pole, batched mjd_transitionFD (eps 1e-6, centred), B = 512, states from the rollout above
    env4 = Env.from_xml_path(os.path.join(ROOT, "models/cartpole.xml"), obs_spec=ObservationSpec(as_dict=False), controller=RandomCtrlController(seed=0, scale=0.005),
                             batch=512, dtype="float32", device=device)
    env4.rollout(100)
    s4 = env4.data.sim
    s4.transition_fd(1e-6, True)
    t = time.perf_counter()
    reps = 50
    for _ in range(reps):
        A, B = s4.transition_fd(1e-6, True)
    dt4 = time.perf_counter() - t
    nvv, nuu = env4.model.nv, env4.model.nu
    fd_bytes = 4 * (env4.model.nq + 2 * nvv + nuu) + 8 * (2 * nvv) * (2 * nvv + nuu)       # state in (fp32), (A, B) out (float64)
    om = mjo.OracleModel(compile_xml_path(os.path.join(ROOT, "models/cartpole.xml")))
    od = mjo.OracleData(om)
    od.rollout_random(100, seed=0, env=0, scale=0.005)
    od.transition_fd(1e-6, True)
    tc = time.perf_counter()
    nc = 2000
    for _ in range(nc):
        od.transition_fd(1e-6, True)
    dtc = time.perf_counter() - tc
    out["config4_cartpole_transition_fd_B512"] = {"value": 512 * reps / dt4, "unit": "linearisations/s", "ms_per_call": dt4 / reps * 1e3, "A_shape": list(A.shape), "B_shape": list(B.shape),
                                                  "kernels": "k_fd (float64, 1 + 2(2nv+nu) = 11 perturbed steps per environment) + k_fd_combine + one pinned copy of (A, B), host-synchronous call",
                                                  "roofline": {"bound": "hbm", "algorithmic_bytes_per_linearisation": fd_bytes, "achieved": 512 * fd_bytes / (dt4 / reps) / 1e9, "peak": HBM_PEAK_GBS,
                                                               "unit": "GB/s", "frac": 512 * fd_bytes / (dt4 / reps) / 1e9 / HBM_PEAK_GBS,
                                                               "note": "whole-call time (launches + copy + sync), not the kernel alone: a 512-environment call is launch-latency bound; kernel time in profiles/r03_other_configs_summary.txt"},
                                                  "cpu_port": {"value": nc / dtc, "unit": "linearisations/s", "cores": 1, "kind": "port", "sample": f"{nc} calls of the oracle's mjd_transitionFD restatement on one state"}}
    # config 5: drone2, B = 2048, ObservationSpec(sites_pos, bodies_pos, as_dict=False) written in-kernel EVERY step (31 floats per environment)
    spec = ObservationSpec(sites_pos=("imu", "thrust1", "thrust2", "thrust3", "thrust4"), bodies_pos=("x2",), as_dict=False)
    env5 = Env.from_xml_path(os.path.join(ROOT, "models/drone2/scene.xml"), obs_spec=spec, controller=RandomCtrlController(seed=0, scale=0.3), batch=2048, shard=True,
                             dtype="float32", device=device, keyframe="hover")
    q = np.array(env5.data.qpos); q[1024:, 2] = 0.1                       # half hovering, half dropped from z = 0.1
    env5.data.qpos[...] = q
    dt5, ms5 = timed_rollouts(env5, 200, 1, 3)
    b5 = 4 * ((env5.model.nq + env5.model.nv) * 2 + env5.model.nu + 2 * env5.model.nv) + 4 * env5.extractor.obs_dim
    out["config5_drone2_obs_every_step_B2048"] = {"value": 2048 * 600 / dt5, "unit": "env-steps/s", "launch_ms": [round(x, 3) for x in ms5], "steps_per_launch": 200, "obs_dim": env5.extractor.obs_dim,
                                                  "obs_ring_bytes_per_launch": 200 * 2048 * env5.extractor.obs_dim * 4, "lanes_per_env": env5.data.sim.lanes,
                                                  "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": b5, "achieved": 2048 * 200 * b5 / (np.mean(ms5) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                                               "unit": "GB/s", "frac": 2048 * 200 * b5 / (np.mean(ms5) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                                  "includes": "the in-kernel observation ring (every step) and its gather (identity at one rank)"}
    return out


class _HostRandomCtrl:
    """A controller that lives in Python (no ``device_ctrl_mode``): the shape of every controller in the reference's examples —
    read ``data``, write ``data.ctrl`` in place, once per step (reference control.py:26-32)."""

    def __init__(self, scale: float, seed: int = 0, needs_linearization: bool = False, needs_jacobians: tuple = ()):
        import numpy as np

        from mujoco_template_amd import ControllerCapabilities, ControlSpace

        # needs_linearization: Env.step hands the controller's step a fresh discrete (A, B) every step (reference env.py:186-205 ->
        # linearization.py:123-135 -> mjd_transitionFD), the shape of gain-scheduled / iLQR-style controllers
        self.capabilities = ControllerCapabilities(control_space=ControlSpace.TORQUE, needs_linearization=needs_linearization,
                                                   needs_jacobians=tuple(needs_jacobians))
        self.scale, self.rng = scale, np.random.default_rng(seed)

    def prepare(self, model, data) -> None:
        import numpy as np

        lo = np.where(model.actuator_ctrllimited, model.actuator_ctrlrange[:, 0], -1.0)
        hi = np.where(model.actuator_ctrllimited, model.actuator_ctrlrange[:, 1], 1.0)
        self.mid, self.half = 0.5 * (lo + hi), 0.5 * (hi - lo) * self.scale

    def __call__(self, model, data, t: float) -> None:
        data.ctrl[...] = self.mid + self.half * self.rng.uniform(-1.0, 1.0, size=data.ctrl.shape)


def host_loop(xml_path: str, scale: float, device: int, batch: int) -> dict:
    """The reference's actual usage shape on this engine (reference runtime.py:631-663): ``Env.passive`` with a
    PYTHON controller in the loop — one ``Env.step`` per step: controller call, edited mirrors up, one mj_step launch, the
    pinned state block back (``mjb_step_host``), observation extraction.  Bounded samples, batch 1 and the bench batch."""
    from mujoco_template_amd import Env, ObservationSpec

    out: dict = {"unit": "env-steps/s", "controller": "python (numpy uniform ctrl written in place every step)",
                 "path": "Env.passive -> Env.step -> mjb_step_host (packed H2D of edited fields + k_step(1) + packed D2H)"}
    for b, nsteps in ((1, 400), (batch, 60)):
        env = Env.from_xml_path(xml_path, obs_spec=ObservationSpec(as_dict=False), controller=_HostRandomCtrl(scale), batch=b,
                                dtype="float32", device=device)
        for return_obs in (True, False):
            for _ in env.passive(max_steps=20, return_obs=return_obs):
                pass
            t = time.perf_counter()
            for _ in env.passive(max_steps=nsteps, return_obs=return_obs):
                pass
            dt = time.perf_counter() - t
            out[f"batch{b}_return_obs_{return_obs}"] = {"value": b * nsteps / dt, "us_per_step": dt / nsteps * 1e6, "steps": nsteps}
        del env
    # the same loop with a controller that asks for the linearisation every step (batch 1): k_fd + k_fd_combine + k_step per Env.step
    env = Env.from_xml_path(xml_path, obs_spec=ObservationSpec(as_dict=False), controller=_HostRandomCtrl(scale, needs_linearization=True), batch=1,
                            dtype="float32", device=device)
    for _ in env.passive(max_steps=10, return_obs=False):
        pass
    t = time.perf_counter()
    n = 0
    for res in env.passive(max_steps=100, return_obs=False):
        n += 1
    dt = time.perf_counter() - t
    A = res.info.get("A")
    out["batch1_needs_linearization"] = {"value": n / dt, "us_per_step": dt / n * 1e6, "steps": n,
                                         "A_shape": list(getattr(A, "shape", ())) if A is not None else None,
                                         "what": "Env.step = centred finite-difference (A, B) of the step (float64, all columns side by side on the device) + controller + step"}
    del env
    # ... and one that asks for two Jacobians every step (operational-space shape: reference env.py:186-205 -> jacobians.py:26-83)
    if "humanoid" in os.path.basename(xml_path):
        env = Env.from_xml_path(xml_path, obs_spec=ObservationSpec(as_dict=False), batch=1, dtype="float32", device=device,
                                controller=_HostRandomCtrl(scale, needs_jacobians=("subtreecom:torso", "bodycom:foot_left")))
        for _ in env.passive(max_steps=10, return_obs=False):
            pass
        t = time.perf_counter()
        n = 0
        for res in env.passive(max_steps=200, return_obs=False):
            n += 1
        dt = time.perf_counter() - t
        out["batch1_needs_jacobians"] = {"value": n / dt, "us_per_step": dt / n * 1e6, "steps": n, "requests": ["subtreecom:torso", "bodycom:foot_left"],
                                         "keys": sorted(k for k in res.info if "jac" in k.lower())}
        del env
    return out


def _oracle_passive_loop(om, mjo, scale: float, nsteps: int, return_obs: bool) -> float:
    """The reference's loop shape on the CPU port: ONE interpreter-level physics call per step at batch 1, controller and
    observation assembled in Python (reference runtime.py:631-663 / env.py:186-230).  Returns steps per second."""
    import numpy as np

    od = mjo.OracleData(om)
    cm = om.compiled
    rng = np.random.default_rng(0)
    lo = np.where(cm.actuator_ctrllimited, cm.actuator_ctrlrange[:, 0], -1.0)
    hi = np.where(cm.actuator_ctrllimited, cm.actuator_ctrlrange[:, 1], 1.0)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo) * scale
    t = time.perf_counter()
    for _ in range(nsteps):
        od.ctrl[:] = mid + half * rng.uniform(-1.0, 1.0, size=cm.nu)
        od.step()
        if return_obs:
            np.concatenate([od.qpos, od.qvel])
    return nsteps / (time.perf_counter() - t)


def cpu_baseline(xml_path: str, scale: float) -> dict:
    """Bounded sample of the same workload on the host cores with the float64 oracle (kind = "port")."""
    from mujoco_template_amd.mjcf import compile_xml_path
    from oracle import mjo

    om = mjo.OracleModel(compile_xml_path(xml_path))
    cores = min(os.cpu_count() or 1, 16)
    mjo.rollout_batch(om, cores, 20, seed=0, scale=scale, nthreads=cores)          # warm-up
    nenv, nstep = 64 * cores, 2000            # ~2 M env-steps: a few seconds of wall time on 16 cores (tens of core-seconds)
    t = time.perf_counter()
    mjo.rollout_batch(om, nenv, nstep, seed=0, scale=scale, nthreads=cores)
    dt = time.perf_counter() - t
    t1 = time.perf_counter()
    mjo.rollout_batch(om, 16, nstep, seed=0, scale=scale, nthreads=1)
    dt1 = time.perf_counter() - t1
    _oracle_passive_loop(om, mjo, scale, 200, True)           # warm-up
    loop = {f"return_obs_{ro}": _oracle_passive_loop(om, mjo, scale, 20000, ro) for ro in (True, False)}
    # ... and with the linearisation every step (mjd_transitionFD, centred, eps 1e-6, as the reference's Env.step does for such controllers)
    od = mjo.OracleData(om)
    od.transition_fd(1e-6, True)
    tl = time.perf_counter()
    nl = 40
    for _ in range(nl):
        od.transition_fd(1e-6, True)
        od.step()
    loop["needs_linearization"] = nl / (time.perf_counter() - tl)
    if om.compiled.nbody > 10:                # humanoid: the two Jacobians of the host_loop leg (subtree COM of body 1, body COM of body 10) + step
        tj = time.perf_counter()
        nj = 5000
        for _ in range(nj):
            od.jac(3, 1)
            od.jac(2, 10)
            od.step()
        loop["needs_jacobians"] = nj / (time.perf_counter() - tj)
    return {"value": nenv * nstep / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "single_core_value": 16 * nstep / dt1,
            "loop_faithful": {"unit": "env-steps/s", "cores": 1, **loop,
                              "what": "the reference's loop shape (runtime.py:631-663): batch 1, one Python-level physics call per step, "
                                      "Python controller + observation, 20000 steps; physics = the float64 C oracle"},
            "sample": f"{nenv} envs x {nstep} steps of the same random-ctrl rollout, float64 C oracle, one env per OpenMP task; "
                      "the reference's own CPU loop (mujoco wheel) is not runnable here"}


if __name__ == "__main__":
    main()
