// Micro-benchmark of mjb::mfma_factor32 (the fp32 MFMA Cholesky of the step kernel) in isolation:
// cycles per call (s_memtime) for mode 0 (factor + solve) and mode 1 (Hessian assembly + factor + solve),
// at 1 wave per CU and at 2 waves per SIMD.  Ablation knobs: -DMJB_MICRO_NOMFMA / NOSTORE / NONR / NOFWD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mujoco_template_amd/csrc/mjb_device.hpp"
struct MiniRef { float timestep; const float* dof_damping; };
__global__ __launch_bounds__(64, 2) void kb(int n, int nefc, int mode, int iters, const float* Min, const float* Jin, const float* dwin, float* xout, unsigned long long* cyc, int pad_lds) {
  extern __shared__ float lds[];
  float *M = lds, *W = M + n * n, *dinv = W + 528, *J = dinv + 40, *dw = J + 64 * n, *x = dw + 64;
  int lane = threadIdx.x;
  for (int i = lane; i < n * n; i += 64) M[i] = Min[i];
  for (int i = lane; i < nefc * n; i += 64) J[i] = Jin[i];
  for (int i = lane; i < 64; i += 64) dw[i] = i < nefc ? dwin[i] : 0.f;
  __syncthreads();
  MiniRef m{0.002f, dwin};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long pf[3] = {0, 0, 0};
  for (int it = 0; it < iters; it++) {
    if (lane < n) x[lane] = 1.0f + 0.01f * lane;
    __syncthreads();
#ifdef MICRO_CHOL
    mjb::mfma_factor32<MiniRef>(m, M, W, dinv, J, dw, nefc, mode, n, lane, x, pf);
#else
    mjb::mfma_sweep_solve32<MiniRef>(m, M, dinv, J, dw, nefc, mode, n, lane, x, pf);
#endif
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = pf[0]; cyc[2] = pf[1]; cyc[3] = pf[2]; }
  if (blockIdx.x == 0 && lane < n) xout[lane] = x[lane];
}
int main(int argc, char** argv) {
  int n = 27, nefc = 24, iters = 200;
  std::vector<float> M(n * n, 0.f), J(64 * n), dw(64);
  srand(1);
  std::vector<float> A(n * n);
  for (auto& v : A) v = (rand() % 2001 - 1000) / 1000.0f;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { float s = i == j ? 2.0f : 0.f; for (int k = 0; k < n; k++) s += A[i * n + k] * A[j * n + k] * 0.1f; M[i * n + j] = s; }
  for (auto& v : J) v = (rand() % 2001 - 1000) / 1000.0f;
  for (int i = 0; i < 64; i++) dw[i] = (i % 2) ? 3.0f : 0.0f;       // half the rows active
  float *dM, *dJ, *dd, *dx; unsigned long long* dc;
  hipMalloc(&dM, M.size() * 4); hipMalloc(&dJ, J.size() * 4); hipMalloc(&dd, 256); hipMalloc(&dx, 256); hipMalloc(&dc, 64);
  hipMemcpy(dM, M.data(), M.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dJ, J.data(), J.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, dw.data(), 256, hipMemcpyHostToDevice);
  size_t lds = 20448;
  for (int blocks : {256, 2048}) for (int mode : {0, 1}) {
    hipLaunchKernelGGL(kb, dim3(blocks), dim3(64), lds, 0, n, nefc, mode, iters, dM, dJ, dd, dx, dc, 0);
    hipDeviceSynchronize();
    unsigned long long c[4]; float x[32];
    hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost); hipMemcpy(x, dx, 128, hipMemcpyDeviceToHost);
    // residual of the solve on the host
    double res = 0;
    for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) { double h = M[i * n + j]; if (mode == 1) for (int r = 0; r < nefc; r++) h += dw[r] * J[r * n + i] * J[r * n + j]; s += h * x[j]; } res = fmax(res, fabs(s - (1.0 + 0.01 * i))); }
    printf("blocks %4d mode %d: %7.0f cycles/call  (load/assembly %6.0f  panel %6.0f  store+back %6.0f)  solve residual %.2e\n", blocks, mode, (double)c[0] / iters, (double)c[1] / iters, (double)c[2] / iters, (double)c[3] / iters, res);
  }
  return 0;
}
