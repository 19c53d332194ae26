#!/bin/bash
# Round-2 evidence set for profiles/ (run on the GPU box from the repo root:  bash scripts/run_profile_set_r02.sh <tag>).
# Every rocprofv3 run has the program directly after `--`; PMC passes are separate runs and never combined with a trace domain.
set -e
tag=${1:-r02}
export TMPDIR=/tmp
out=gpurun_out/${tag}
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_shape.json 2> $out/bench_driver_shape.err
B="python3 bench.py --no-cpu-baseline --no-host-loop"
# the SQ_* counters are 32-bit per shader engine: over a 100 ms launch (1000 fused steps) they wrap, so the instruction-mix / wait /
# MFMA passes run the same rollout in 100-step launches; the HBM passes (KiB units) keep the bench line's own launch length
C="python3 bench.py --no-cpu-baseline --no-host-loop --chunk 100"
D="python3 bench.py --no-cpu-baseline --no-host-loop --steps 20 --warmup 5"
F="python3 scripts/prof_fd.py"
rp() { d=$1; shift; rocprofv3 "$@" > $out/$d.log 2>&1; echo "$d done"; }
rp kt    --kernel-trace --stats --output-format csv -d $out/kt -- $B
rp ckt   --kernel-trace --stats --output-format csv -d $out/ckt -- $C
rp pmc1  --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/pmc1 -- $C
rp pmc2  --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc2 -- $C
rp pmc3  --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- $B
rp pmc4  --pmc WRITE_SIZE --output-format csv -d $out/pmc4 -- $B
rp pmc5  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc5 -- $C
rp dkt   --kernel-trace --stats --output-format csv -d $out/dkt -- $D
rp dpmc3 --pmc FETCH_SIZE --output-format csv -d $out/dpmc3 -- $D
rp dpmc4 --pmc WRITE_SIZE --output-format csv -d $out/dpmc4 -- $D
rp fkt   --kernel-trace --stats --output-format csv -d $out/fkt -- $F
rp fpmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/fpmc1 -- $F
rp fpmc2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/fpmc2 -- $F
rp fpmc3 --pmc FETCH_SIZE --output-format csv -d $out/fpmc3 -- $F
rp fpmc4 --pmc WRITE_SIZE --output-format csv -d $out/fpmc4 -- $F
rp fpmc5 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/fpmc5 -- $F
# keep the merge small: the per-dispatch CSVs are what the summariser reads
find $out -name "*.db" -delete
du -sh $out
echo profile-set-done
