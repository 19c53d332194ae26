# Probe compiler flags on the per-model specialised kernel (MJB_SPEC_FLAGS is appended to the hipcc --genco line, which
# already carries -ffp-contract=on and -mllvm -amdgpu-sched-strategy=iterative-ilp).  Round-1 findings on the humanoid:
# iterative-ilp +4 % over the default scheduler (max-ilp / max-memory-clause / iterative-minreg slower, iterative-maxocc +3 %),
# -fno-unroll-loops -8 %, a larger -unroll-threshold no change.
for fl in "" "-mllvm -enable-post-misched=false" "-mllvm -amdgpu-enable-power-sched=true" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-mllvm -amdgpu-use-aa-in-codegen=true -mllvm -enable-aa-sched-mi"; do
  echo "== flags: '$fl'"
  MJB_SPEC_FLAGS="$fl" python scripts/gpu_sweep.py 2>&1 | grep -v amdgpu | grep -E "B=512 caps|B=4096 caps 64/24|arn" | head -2
done
