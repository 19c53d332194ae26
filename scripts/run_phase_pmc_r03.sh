#!/bin/bash
# Per-phase hardware-counter table of k_step (round 3): run on the GPU box from the repo root:  bash scripts/run_phase_pmc_r03.sh [tag]
# One kernel-trace pass + two PMC passes per variant (none / each idempotent phase repeated); PMC never combined with a trace domain.
set -e
tag=${1:-r03_phase}
export TMPDIR=/tmp
out=gpurun_out/${tag}
mkdir -p $out
W="python3 scripts/gpu_phase_pmc.py"
for rep in ${REPS:--1 0 1 2 3 4 5 6 8}; do
  export MJB_REPEAT_PHASE=$rep
  n=rep${rep}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${n}_kt -- $W > $out/${n}_kt.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/${n}_pmcA -- $W > $out/${n}_pmcA.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY --output-format csv -d $out/${n}_pmcB -- $W > $out/${n}_pmcB.log 2>&1
  echo "$n done"
done
find $out -name "*.db" -delete
du -sh $out
echo phase-pmc-done
