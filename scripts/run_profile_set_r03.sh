#!/bin/bash
# Round-3 evidence set for profiles/ (run on the GPU box from the repo root:  bash scripts/run_profile_set_r03.sh <tag>).
# Every rocprofv3 run has the program directly after `--`; PMC passes are separate runs and never combined with a trace domain.
set -e
tag=${1:-r03}
export TMPDIR=/tmp
out=gpurun_out/${tag}
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_shape.json 2> $out/bench_driver_shape.err
B="python3 bench.py --no-cpu-baseline --no-host-loop --no-other-configs"
C="python3 bench.py --no-cpu-baseline --no-host-loop --no-other-configs --chunk 100"
D="python3 bench.py --no-cpu-baseline --no-host-loop --no-other-configs --steps 20 --warmup 5"
O="python3 bench.py --no-cpu-baseline --no-host-loop --steps 20 --warmup 5"
S5="python3 bench.py --no-cpu-baseline --no-host-loop --no-other-configs --global-batch 512 --chunk 100"
S10="python3 bench.py --no-cpu-baseline --no-host-loop --no-other-configs --global-batch 1024 --chunk 100"
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
rp okt   --kernel-trace --stats --output-format csv -d $out/okt -- $O
rp s5kt  --kernel-trace --stats --output-format csv -d $out/s5kt -- $S5
rp s5pmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/s5pmc1 -- $S5
rp s5pmc2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/s5pmc2 -- $S5
rp s10kt --kernel-trace --stats --output-format csv -d $out/s10kt -- $S10
rp s10pmc2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/s10pmc2 -- $S10
find $out -name "*.db" -delete
du -sh $out
echo profile-set-done
