// Micro-test: layout of v_mfma_f32_32x32x2_f32 operands/accumulator and v_permlane32_swap on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* A /*32x2 row-major*/, const float* B /*2x32*/, float* D /*32x32*/, int* swapout) {
  int l = threadIdx.x;
  float a = A[(l % 32) * 2 + (l / 32)];
  float b = B[(l / 32) * 32 + (l % 32)];
  f16v acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 16; i++) {
    int row = 8 * (i / 4) + 4 * (l / 32) + (i % 4), col = l % 32;
    D[row * 32 + col] = acc[i];
  }
  // permlane32_swap: exchange a value between lane l and lane l^32 ?
  int v = l;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  swapout[l] = r[0]; swapout[64 + l] = r[1];
#else
  swapout[l] = -1; swapout[64 + l] = -1;
#endif
}
int main() {
  std::vector<float> A(64), B(64), D(1024), R(1024, 0.f);
  for (int i = 0; i < 64; i++) { A[i] = 1.0f + i * 0.37f; B[i] = -2.0f + i * 0.11f; }
  for (int r = 0; r < 32; r++) for (int c = 0; c < 32; c++) for (int k = 0; k < 2; k++) R[r * 32 + c] += A[r * 2 + k] * B[k * 32 + c];
  float *dA, *dB, *dD; int* dS;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096); hipMalloc(&dS, 512);
  hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  int S[128]; hipMemcpy(S, dS, 512, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; i++) err = fmax(err, fabs(D[i] - R[i]));
  printf("mfma 32x32x2 f32 layout max err %.3e\n", err);
  printf("permlane32_swap r0[0..3]=%d %d %d %d r0[32..35]=%d %d %d %d | r1[0..3]=%d %d %d %d r1[32..35]=%d %d %d %d\n", S[0], S[1], S[2], S[3], S[32], S[33], S[34], S[35], S[64], S[65], S[66], S[67], S[96], S[97], S[98], S[99]);
  return 0;
}
