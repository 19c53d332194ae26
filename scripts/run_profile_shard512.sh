#!/bin/bash
# Evidence for the N = 8 strong-scaling shard (512 humanoids per GPU -> the two-wave kernel mjb_k_step2_spec): kernel trace + SQ passes of
# the bench with that batch, 100-step launches.  Run on the GPU box from the repo root:  bash scripts/run_profile_shard512.sh <tag>
set -e
tag=${1:-r02s512}
export TMPDIR=/tmp
out=gpurun_out/${tag}
mkdir -p $out
C="python3 bench.py --no-cpu-baseline --no-host-loop --global-batch 512 --chunk 100"
rp() { d=$1; shift; rocprofv3 "$@" > $out/$d.log 2>&1; echo "$d done"; }
rp ckt   --kernel-trace --stats --output-format csv -d $out/ckt -- $C
rp pmc1  --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/pmc1 -- $C
rp pmc2  --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc2 -- $C
rp pmc5  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc5 -- $C
find $out -name "*.db" -delete
du -sh $out
echo profile-set-done
