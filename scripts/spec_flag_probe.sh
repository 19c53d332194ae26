# Probe compiler flags on the per-model specialised kernel (MJB_SPEC_FLAGS is appended to the hipcc --genco line).
for fl in "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -unroll-threshold=1200" "-mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -unroll-threshold=3000" "-mllvm -amdgpu-sched-strategy=iterative-minreg" "-mllvm -amdgpu-sched-strategy=iterative-maxocc"; do
  echo "== flags: '$fl'"
  MJB_SPEC_FLAGS="$fl" python scripts/gpu_sweep.py 2>&1 | grep -v amdgpu | grep -E "B=512 caps|B=4096 caps 64/24|arn" | head -2
done
MJB_SPEC_FLAGS="-mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -unroll-threshold=1200" python -m pytest tests -q -m gpu -k "specialised or full_size or fp32_teacher" 2>&1 | tail -2
