"""Per-phase table of k_step from the PMC set of scripts/run_phase_pmc_r03.sh (gpurun_out/<tag>/rep<k>_{kt,pmcA,pmcB}):
phase k's own counts = (pass with phase k repeated) - (pass with nothing repeated), per env-step.  Writes
profiles/<out>_phase_table.{txt,json}; the whole-kernel rows feed bench.py's `flop_roofline`.

usage: python scripts/summarize_phase_pmc.py gpurun_out/r03_phase r03 [--envsteps-per-launch 409600] [--launches 10]
"""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
EPS = float(sys.argv[sys.argv.index("--envsteps-per-launch") + 1]) if "--envsteps-per-launch" in sys.argv else 409600.0
K = int(sys.argv[sys.argv.index("--launches") + 1]) if "--launches" in sys.argv else 10
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = {0: "kinematics (A1)", 1: "com_pos, cinert, cdof, tendons (A2/A3)", 3: "collision (A5)", 2: "crb -> M (A4)", 4: "constraint rows (A6)",
          5: "velocity stage: cvel, RNE bias, passive (A7)", 6: "actuation + M^-1 f (A8/A9)", 8: "Euler: (M + hD)^-1 solve (A11)"}
# pieces of the Newton solver and of the step loop (REP_* repeat points of mjb_device.hpp; present when the set was run with them)
SOLVER = {32: "  solver: Cholesky of H, factor + solve (all factorisations of the step)", 33: "  solver: solves that reuse the factor", 34: "  solver: [M; J] x search",
          35: "  solver: line search", 36: "  solver: warm start (two cost evaluations)", 37: "  solver: gradient (J^T f)", 38: "  step loop: random ctrl (Philox)"}
SOLVER_ORACLE = {32: ["solver: Cholesky + solve (A10)"], 33: [], 34: ["solver: M v, J v (A10)"], 35: ["solver: line search (A10)"], 36: ["solver: costs / warm start (A10)"],
                 37: ["solver: gradient + Hessian assembly (A10)"], 38: []}
# the oracle's phases that correspond to each device phase (profiles/r03_flops_per_env_step.json)
ORACLE = {0: ["kinematics (A1)"], 1: ["com_pos (A2)", "tendon/transmission (A3)"], 3: ["collision (A5)"], 2: ["crb + factor M (A4)"], 4: ["constraint rows (A6)"],
          5: ["com_vel (A7)", "passive (A7)", "rne bias (A7)"], 6: ["actuation (A8)", "M^-1 f (A9)"], 8: ["integrator (A11)"]}


def counters(d, kernel):
    agg = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        keep = set(ids[-K:])
        for r in rows:
            if int(r["Dispatch_Id"]) in keep:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]) / len(keep)
    return dict(agg)


def duration_ms(d, kernel):
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
        if len(durs) >= K:
            return sum(durs[-K:]) / K
    return float("nan")


def one(rep):
    c = {}
    for p in ("pmcA", "pmcB"):
        c.update(counters(os.path.join(src, f"rep{rep}_{p}"), "k_step"))
    c["ms"] = duration_ms(os.path.join(src, f"rep{rep}_kt"), "k_step")
    return c


base = one(-1)
# unit check: a full-lane torch kernel (x.mul_): thread-cycles per active-VALU count with all 64 lanes on
full = {}
for p in ("pmcA",):
    full.update(counters(os.path.join(src, f"rep-1_{p}"), "MulFunctor"))
if not full.get("SQ_ACTIVE_INST_VALU"):
    full = counters(os.path.join(src, "rep-1_pmcA"), "vectorized_elementwise_kernel")
unit = full["SQ_THREAD_CYCLES_VALU"] / full["SQ_ACTIVE_INST_VALU"] if full.get("SQ_ACTIVE_INST_VALU") else float("nan")


def derive(c):
    """per env-step quantities of one set of counters (whole kernel or a phase's difference)"""
    valu = c.get("SQ_INSTS_VALU", 0.0) / EPS
    lanes = 64.0 * (c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_ACTIVE_INST_VALU"]) / unit if c.get("SQ_ACTIVE_INST_VALU") else float("nan")
    fp = {k: c.get("SQ_INSTS_VALU_" + k, 0.0) / EPS for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT")}
    mfma_mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / EPS
    fp_insts = fp["ADD_F32"] + fp["MUL_F32"] + fp["FMA_F32"] + fp["TRANS_F32"]
    # executed fp32 lane-flops: wave instructions x mean active lanes (the kernel-wide mean is applied to every class), FMA = 2;
    # MFMA: 512 flops per MOP unit (rocprofiler's MfmaFlopsF32 definition), all of the 32x32 tile counted whether padded or not
    flops_valu = (fp["ADD_F32"] + fp["MUL_F32"] + fp["TRANS_F32"] + 2.0 * fp["FMA_F32"]) * lanes
    flops_mfma = mfma_mops * 512.0
    return {"ms_per_launch": c.get("ms"), "valu_insts": valu, "salu_insts": c.get("SQ_INSTS_SALU", 0.0) / EPS, "lds_insts": c.get("SQ_INSTS_LDS", 0.0) / EPS,
            "mean_active_lanes": lanes, "fp32_insts": fp_insts, "fma_f32": fp["FMA_F32"], "add_f32": fp["ADD_F32"], "mul_f32": fp["MUL_F32"], "trans_f32": fp["TRANS_F32"],
            "int32_insts": fp["INT32"], "cvt_insts": fp["CVT"], "mfma_mops_f32": mfma_mops, "executed_valu_flops": flops_valu, "executed_mfma_flops": flops_mfma,
            "wave_cycles": 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / EPS, "issue_cycles_any": 4.0 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / EPS,
            "wait_any_cycles": 4.0 * c.get("SQ_WAIT_ANY", 0.0) / EPS, "dep_stall_cycles": 4.0 * c.get("SQ_WAIT_INST_ANY", 0.0) / EPS}


oracle = {}
fj = os.path.join(ROOT, "profiles", "r03_flops_per_env_step.json")
if os.path.exists(fj):
    oj = json.load(open(fj))
    oracle = {r["phase"]: r["flops"] for r in oj["phases"]}
    oracle_total = oj["flops_per_env_step"]
else:
    oracle_total = float("nan")
whole = derive(base)
rows, acc = [], collections.defaultdict(float)
for rep, name in PHASES.items():
    c = one(rep)
    diff = {k: c.get(k, 0.0) - base.get(k, 0.0) for k in set(c) | set(base) if k != "ms"}
    diff["ms"] = c["ms"] - base["ms"]
    r = derive(diff)
    r["phase"] = name
    r["algorithmic_flops"] = sum(oracle.get(n, 0.0) for n in ORACLE[rep])
    rows.append(r)
    for k, v in r.items():
        if isinstance(v, float) and k != "mean_active_lanes":
            acc[k] += v
sub = []
for rep, name in SOLVER.items():
    if not os.path.isdir(os.path.join(src, f"rep{rep}_pmcA")):
        continue
    c = one(rep)
    diff = {k: c.get(k, 0.0) - base.get(k, 0.0) for k in set(c) | set(base) if k != "ms"}
    diff["ms"] = c["ms"] - base["ms"]
    r = derive(diff)
    r["phase"] = name
    r["algorithmic_flops"] = sum(oracle.get(n, 0.0) for n in SOLVER_ORACLE[rep])
    sub.append(r)
rest = {k: whole[k] - acc[k] for k in whole if isinstance(whole[k], float) and k != "mean_active_lanes"}
# active lanes of the remainder from the thread-cycle balance
tc_rest = base.get("SQ_THREAD_CYCLES_VALU", 0.0) - sum((one(rep).get("SQ_THREAD_CYCLES_VALU", 0.0) - base.get("SQ_THREAD_CYCLES_VALU", 0.0)) for rep in PHASES)
ai_rest = base.get("SQ_ACTIVE_INST_VALU", 0.0) - sum((one(rep).get("SQ_ACTIVE_INST_VALU", 0.0) - base.get("SQ_ACTIVE_INST_VALU", 0.0)) for rep in PHASES)
rest["mean_active_lanes"] = 64.0 * (tc_rest / ai_rest) / unit if ai_rest else float("nan")
rest["phase"] = "Newton solver (A10) + step bookkeeping, ctrl, I/O, hand-over (remainder)"
rest["algorithmic_flops"] = sum(v for k, v in oracle.items() if k.startswith("solver"))
rows.append(rest)
out = [f"# per-phase hardware counters of mjb_k_step_spec (humanoid, B = 4096, 100-step launches, last {K} launches; per ENV-STEP).",
       "# phase k = (PMC pass with phase k run twice, -DMJB_PHASE_REPEAT) - (pass with nothing repeated); remainder = whole kernel - the phases above.",
       f"# unit check: full-lane torch kernel SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = {unit:.3f} (taken as 64 active lanes)",
       f"# algorithmic flops: float64 oracle, instrumented build (profiles/r03_flops_per_env_step.json): {oracle_total:.0f} per env-step",
       "",
       f"{'phase':78s} {'us/launch':>9s} {'share':>6s} {'VALU':>7s} {'lanes':>6s} {'fp32 in':>8s} {'MFMA':>6s} {'exec flop':>10s} {'algo flop':>10s} {'useful':>7s}"]
tot_ms = whole["ms_per_launch"]
for r in rows + sub + [dict(whole, phase="WHOLE KERNEL", algorithmic_flops=oracle_total)]:
    ex = r["executed_valu_flops"] + r["executed_mfma_flops"]
    out.append(f"{r['phase']:78s} {1e3 * r['ms_per_launch']:9.1f} {100 * r['ms_per_launch'] / tot_ms:5.1f}% {r['valu_insts']:7.0f} {r['mean_active_lanes']:6.1f} {r['fp32_insts']:8.0f} "
               f"{r['mfma_mops_f32']:6.1f} {ex:10.0f} {r['algorithmic_flops']:10.0f} {r['algorithmic_flops'] / ex if ex else float('nan'):7.2f}")
out += ["", "(the indented rows are PART of the remainder row above them: pieces of the solver / step loop measured the same way)", "columns: us/launch = kernel-trace duration difference (409 600 env-steps per launch); VALU = wave-level VALU instructions; lanes = mean active lanes per VALU",
        "instruction; fp32 in = ADD + MUL + FMA + TRANS fp32 wave instructions; MFMA = SQ_INSTS_VALU_MFMA_MOPS_F32; exec flop = fp32 lane-flops actually executed",
        "(VALU classes x active lanes, FMA = 2, + 512 per MFMA MOP, padding included); algo flop = the oracle's count for the same phase; useful = algo / exec."]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_phase_table.txt"), "w").write("\n".join(out) + "\n")
json.dump({"envsteps_per_launch": EPS, "launches": K, "full_lane_unit": unit, "whole_kernel": whole, "phases": rows, "solver_pieces": sub, "raw_whole_kernel_counters": base,
           "algorithmic_flops_per_env_step": oracle_total}, open(os.path.join(ROOT, "profiles", f"{tag}_phase_table.json"), "w"), indent=1)
print("\n".join(out))
