#!/bin/bash
# Evidence set for profiles/: the bench line, a rocprofv3 kernel trace and the PMC passes (each in its OWN run, never
# combined with a trace domain).  Run on the GPU box from the repo root:  bash scripts/run_profile_set.sh <tag>
# then locally:  python scripts/summarize_profile.py <tag> gpurun_out/<tag>_kt gpurun_out/<tag>_pmc{1..6} --envsteps-per-launch 409600
set -e
tag=${1:-r01x}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 3000 $out/${tag}_bench.json
B="python bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -- $B > $out/${tag}_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/${tag}_pmc1 -- $B > $out/${tag}_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/${tag}_pmc2 -- $B > $out/${tag}_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc3 -- $B > $out/${tag}_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc4 -- $B > $out/${tag}_pmc4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc5 -- $B > $out/${tag}_pmc5.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $out/${tag}_pmc6 -- $B > $out/${tag}_pmc6.log 2>&1
echo profile-set-done
