// mjb_device.hpp — the batched mj_step pipeline as group-cooperative device code.
//
// One *group* of G lanes (G = 64: one CDNA4 wavefront; smaller G packs several
// environments into one wavefront for small models) advances one environment.
// All per-environment intermediates live in that group's LDS slice (layout:
// mjb::Lay); model constants are read through the vector cache (identical
// addresses for every wave -> L1/L2 hits).  Control flow is group-uniform;
// lanes split the work by index (bodies of one tree level, dofs, geom pairs,
// constraint rows, matrix entries) and meet at gsync().
//
// Replaces, per environment, the third-party calls behind the reference's
//   ModelHandle.step/forward  (reference mujoco_template/model.py:53-57)
// following MuJoCo's documented pipeline [MJ-KNOWLEDGE].
//
// The same source compiles in a host emulation (MJB_HOST_EMU: one std::thread
// per lane, real barriers) that only the CPU test-suite uses to check the
// kernel logic without a GPU; the product library never contains that build.
#pragma once
#include "mjb_types.hpp"

#ifdef MJB_HOST_EMU
#include <cmath>
#include "mjb_hostemu.hpp"
#define MJB_DEV static inline
#define MJB_DEVM inline
#else
#include <hip/hip_runtime.h>
#define MJB_DEV __device__ __forceinline__
#define MJB_DEVM __device__ __forceinline__
#endif

// Wave-level intrinsics behind macros: the device build expands them to the gfx950 builtins (unchanged code), the test-only host
// emulation (mjb_hostemu.hpp) to collectives of its lane threads - so the CPU suite runs the MFMA / readlane solver code itself.
#ifdef MJB_HOST_EMU
#define MJB_MFMA(a, b, acc) emu_mfma((a), (b), (acc))
#define MJB_BALLOT(p) emu_ballot(p)
#define MJB_RSQF(x) (1.0f / sqrtf(x))
#define MJB_RCPF(x) (1.0f / (x))
#define MJB_MEMTIME() 0ull
#define MJB_OPAQUE1(a) ((void)0)
#define MJB_OPAQUE2(a, b) ((void)0)
#define MJB_OPAQUE9(a, b, c, d, e, f, g, h, i) ((void)0)
#else
#define MJB_MFMA(a, b, acc) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)
#define MJB_BALLOT(p) __ballot(p)
#define MJB_RSQF(x) __builtin_amdgcn_rsqf(x)
#define MJB_RCPF(x) __builtin_amdgcn_rcpf(x)
#define MJB_MEMTIME() __builtin_amdgcn_s_memtime()
#define MJB_OPAQUE1(a) asm volatile("" : "+v"(a))
#define MJB_OPAQUE2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define MJB_OPAQUE9(a, b, c, d, e, f, g, h, i) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i))
#endif

namespace mjb {

#ifndef MJB_HOST_EMU
typedef float mjb_f16v __attribute__((ext_vector_type(16)));     // the 32x32 fp32 MFMA accumulator: 16 registers per lane
#endif

// ---------------------------------------------------------------------------
// group primitives
// ---------------------------------------------------------------------------
#ifndef MJB_HOST_EMU
// Every kernel runs ONE wavefront per workgroup and an environment never spans waves, so a group sync only has to order
// the wave's own LDS traffic.  DS instructions of a wave execute in issue order: a wavefront-scope fence (a compiler
// ordering point, no s_waitcnt vmcnt(0)/s_barrier) is enough, and outstanding global/constant loads stay in flight.
template <int G> MJB_DEV void gsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP lane permutations inside a 16-lane row (no LDS traffic): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141, row_mirror = 0x140.  After the four steps every lane of a row holds the row total.
template <int CTRL> MJB_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> MJB_DEV float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL> MJB_DEV double dpp_f(double v) {
  return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
MJB_DEV int rdlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
MJB_DEV float rdlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
MJB_DEV double rdlane_f(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// The four row totals of a wavefront combined with the two wave-level DPP broadcasts of gfx9 (row_bcast:15 into rows 1 and 3, then
// row_bcast:31 into rows 2 and 3: lane 63 ends with (r0 + r1) + (r2 + r3)) and ONE v_readlane - instead of four v_readlane and three
// scalar-operand adds.  Same association as the readlane form, so the sums are bitwise the same.  As inline assembly: the compiler
// does not fold a partial row mask into the add (it emits a zero, a v_mov_dpp and the add); the s_nop 1 are the two wait states a
// DPP read needs after the VALU write of its source.  float only (the float64 / integer forms keep the readlanes: not on the hot path).
MJB_DEV float wave_rows_sum(float v) {
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return rdlane_f(v, 63);
}
MJB_DEV double wave_rows_sum(double v) { return (rdlane_f(v, 0) + rdlane_f(v, 16)) + (rdlane_f(v, 32) + rdlane_f(v, 48)); }
MJB_DEV int wave_rows_sum(int v) { return (rdlane_i(v, 0) + rdlane_i(v, 16)) + (rdlane_i(v, 32) + rdlane_i(v, 48)); }
template <typename T, int G> MJB_DEV T gsum(T v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  if (G >= 8) v += dpp_f<0x141>(v);
  if (G >= 16) v += dpp_f<0x140>(v);
  if (G == 64) v = wave_rows_sum(v);
  return v;
}
template <int G> MJB_DEV int gsumi(int v) {
  v += dpp_i<0xB1>(v);
  v += dpp_i<0x4E>(v);
  if (G >= 8) v += dpp_i<0x141>(v);
  if (G >= 16) v += dpp_i<0x140>(v);
  if (G == 64) v = wave_rows_sum(v);
  return v;
}
template <int G> MJB_DEV int gmaxi(int v) {
  int t;
  t = dpp_i<0xB1>(v); v = t > v ? t : v;
  t = dpp_i<0x4E>(v); v = t > v ? t : v;
  if (G >= 8) { t = dpp_i<0x141>(v); v = t > v ? t : v; }
  if (G >= 16) { t = dpp_i<0x140>(v); v = t > v ? t : v; }
  if (G == 64) {
    int a0 = rdlane_i(v, 0), a1 = rdlane_i(v, 16), a2 = rdlane_i(v, 32), a3 = rdlane_i(v, 48);
    a0 = a0 > a1 ? a0 : a1; a2 = a2 > a3 ? a2 : a3; v = a0 > a2 ? a0 : a2;
  }
  return v;
}
// true if any lane of the group holds true (one wavefront per environment: a ballot)
template <int G> MJB_DEV bool gany(bool v) {
  if (G == 64) return __ballot(v) != 0ull;
  return gsumi<G>((int)v) != 0;
}
// exclusive prefix sum over the group for SMALL counts (0 <= v < 8): three ballots + population counts, no LDS crossbar
template <int G> MJB_DEV int gscan_small(int v, int lane, int& total) {
  const int wl = (int)(threadIdx.x & 63);
  const unsigned long long gm = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << (wl - lane));
  const unsigned long long lt = gm & ((1ull << wl) - 1ull);
  int pre = 0, tot = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const unsigned long long mk = __ballot((v >> b) & 1);
    pre += __popcll(mk & lt) << b;
    tot += __popcll(mk & gm) << b;
  }
  total = tot;
  return pre;
}
template <int G> MJB_DEV int gscan_excl(int v, int lane, int& total) {
  int x = v;
#pragma unroll
  for (int o = 1; o < G; o <<= 1) { int y = __shfl_up(x, o, G); if (lane >= o) x += y; }
  total = __shfl(x, G - 1, G);
  return x - v;
}
// value of lane `src` of the group; src must be group-uniform.  One wavefront per environment: v_readlane (no LDS hop).
template <typename T, int G> MJB_DEV T gshfl(T v, int src) {
  if (G == 64) return rdlane_f(v, __builtin_amdgcn_readfirstlane(src));
  return __shfl(v, src, G);
}
#endif

template <typename T> struct Num;
template <> struct Num<float> {
  MJB_DEVM static float minval() { return 1e-15f; }
};
template <> struct Num<double> {
  MJB_DEVM static double minval() { return 1e-15; }
};
// x[k] for a group-uniform k, n <= 64: with one wavefront per environment the vector lives in one VGPR and is
// broadcast with v_readlane (scalar lane select) instead of an LDS read; otherwise it is read from LDS.
template <typename T, int G> struct VecBcast {
  const T* x;
  T r0;
  MJB_DEVM VecBcast(const T* x_, int n, int lane) : x(x_), r0(0) {
#ifndef MJB_HOST_EMU
    if (G == 64 && lane < n) r0 = x_[lane];
#endif
  }
  MJB_DEVM T get(int k) const {
#ifndef MJB_HOST_EMU
    if (G == 64) return rdlane_f(r0, k);
#endif
    return x[k];
  }
};

// sum_k a[k*stride] * x(k), k < n, manually unrolled by 8 (independent LDS reads in flight; v_readlane is a
// convergent operation, so the compiler will not unroll such loops by itself)
template <typename T, typename X> MJB_DEV T dot_lds(const T* a, int stride, const X& x, int n) {
  T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    const T* p = a + k * stride;
    T a0 = p[0], a1 = p[stride], a2 = p[2 * stride], a3 = p[3 * stride], a4 = p[4 * stride], a5 = p[5 * stride], a6 = p[6 * stride], a7 = p[7 * stride];
    s0 += a0 * x.get(k); s1 += a1 * x.get(k + 1); s2 += a2 * x.get(k + 2); s3 += a3 * x.get(k + 3);
    s0 += a4 * x.get(k + 4); s1 += a5 * x.get(k + 5); s2 += a6 * x.get(k + 6); s3 += a7 * x.get(k + 7);
  }
  for (; k < n; k++) s0 += a[k * stride] * x.get(k);
  return (s0 + s1) + (s2 + s3);
}
template <typename T> struct VecLds { const T* x; MJB_DEVM T get(int k) const { return x[k]; } };

#define MJB_MINIMP ((T)0.0001)
#define MJB_MAXIMP ((T)0.9999)
#define MJB_PI ((T)3.14159265358979323846)

template <typename T> MJB_DEV T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> MJB_DEV T t_min(T a, T b) { return a < b ? a : b; }
template <typename T> MJB_DEV T t_abs(T a) { return a < 0 ? -a : a; }
MJB_DEV float t_sqrt(float a) { return sqrtf(a); }
MJB_DEV double t_sqrt(double a) { return sqrt(a); }
#ifndef MJB_HOST_EMU
MJB_DEV float t_rsqrt(float a) { float y = __builtin_amdgcn_rsqf(a); return y * (1.5f - 0.5f * a * y * y); }
#else
MJB_DEV float t_rsqrt(float a) { return 1.0f / sqrtf(a); }
#endif
#ifndef MJB_HOST_EMU
// v_rsq_f64 (about 2^-26 relative) + two Newton steps on y <- y (1.5 - 0.5 a y^2): full double precision in ~10 instructions
// instead of the library sqrt + division on the pivot chain of the float64 factorisations (finite differences, validation path)
MJB_DEV double t_rsqrt(double a) {
  double y = __builtin_amdgcn_rsq(a);
  const double ha = 0.5 * a;
  y = y * (1.5 - ha * y * y);
  y = y * (1.5 - ha * y * y);
  return y;
}
#else
MJB_DEV double t_rsqrt(double a) { return 1.0 / sqrt(a); }
#endif
MJB_DEV float t_sin(float a) { return sinf(a); }
MJB_DEV double t_sin(double a) { return sin(a); }
MJB_DEV float t_cos(float a) { return cosf(a); }
MJB_DEV double t_cos(double a) { return cos(a); }
MJB_DEV float t_atan2(float a, float b) { return atan2f(a, b); }
MJB_DEV double t_atan2(double a, double b) { return atan2(a, b); }
MJB_DEV float t_pow(float a, float b) { return powf(a, b); }
MJB_DEV double t_pow(double a, double b) { return pow(a, b); }

// ---------------------------------------------------------------------------
// 3-vector / quaternion / spatial helpers (register resident)
// ---------------------------------------------------------------------------
template <typename T> MJB_DEV T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T> MJB_DEV void cross3(T* r, const T* a, const T* b) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
// fp32: 1 / |a| as v_rsq_f32 + one Newton step (t_rsqrt: full single precision, ~5 instructions) instead of the correctly rounded
// sqrt followed by a correctly rounded division (~22): normalisations are all over the position stage (every body, geom, contact
// frame).  float64 keeps sqrt and division (the validation path is held to 1e-12 against the oracle).
template <typename T> MJB_DEV T normalize3(T* a) {
  const T n2 = dot3(a, a);
#ifdef MJB_R2_KINEMATICS
  if constexpr (false) {
#else
  if constexpr (sizeof(T) == 4) {
#endif
    if (n2 < Num<T>::minval() * Num<T>::minval()) { a[0] = 1; a[1] = 0; a[2] = 0; return t_sqrt(n2); }
    const T s = t_rsqrt(n2);
    a[0] *= s; a[1] *= s; a[2] *= s;
    return n2 * s;
  } else {
    T n = t_sqrt(n2);
    if (n < Num<T>::minval()) { a[0] = 1; a[1] = 0; a[2] = 0; } else { T s = 1 / n; a[0] *= s; a[1] *= s; a[2] *= s; }
    return n;
  }
}
template <typename T> MJB_DEV void mulmatvec3(T* r, const T* m, const T* v) {
  T x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> MJB_DEV void mulmatTvec3(T* r, const T* m, const T* v) {
  T x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2], z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> MJB_DEV void quat_mul(T* r, const T* a, const T* b) {
  T w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  T x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  T y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  T z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
template <typename T> MJB_DEV void quat_normalize(T* q) {
  const T n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
#ifdef MJB_R2_KINEMATICS
  if constexpr (false) {
#else
  if constexpr (sizeof(T) == 4) {
#endif
    if (n2 < Num<T>::minval() * Num<T>::minval()) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    const T s = t_rsqrt(n2);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  } else {
    T n = t_sqrt(n2);
    if (n < Num<T>::minval()) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { T s = 1 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
  }
}
template <typename T> MJB_DEV void quat2mat(T* m, const T* q) {
  T w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
// sin/cos of a half joint angle.  fp32: minimax polynomials on |x| <= pi/2 (|error| < 1.5e-7, i.e. fp32 rounding level);
// joint angles beyond +-pi fall back to the library functions.  fp64: library functions.
MJB_DEV void half_sincos(float x, float& s, float& c) {
  // joint angles beyond +-pi: x - k pi in three exact pieces of pi (Cody-Waite), sign (-1)^k - a handful of instructions instead of
  // the library's sinf + cosf (which inline to ~460 instructions at every call site for a branch almost never taken).  k = 0 below
  // pi/2, so the common case is bitwise what it was; up to |x| ~ 1e4 the reduced argument is good to fp32 rounding.
  float sgn = 1.0f;
  if (x > 1.5708f || x < -1.5708f) {
    const float k = rintf(x * 0.318309886f);
    x = fmaf(-k, 3.140625f, x);
    x = fmaf(-k, 9.67502593994140625e-4f, x);
    x = fmaf(-k, 1.50995799097837643e-7f, x);
    sgn = ((int)k & 1) ? -1.0f : 1.0f;
  }
  float x2 = x * x;
  s = x * (1.0f + x2 * (-1.6666667e-1f + x2 * (8.3333310e-3f + x2 * (-1.9840874e-4f + x2 * (2.7525562e-6f + x2 * -2.3889859e-8f)))));
  c = 1.0f + x2 * (-0.5f + x2 * (4.1666668e-2f + x2 * (-1.3888889e-3f + x2 * (2.4801587e-5f + x2 * (-2.7557314e-7f + x2 * 2.0875723e-9f)))));
  s *= sgn; c *= sgn;
}
MJB_DEV void half_sincos(double x, double& s, double& c) { s = sin(x); c = cos(x); }
template <typename T> MJB_DEV void axisangle2quat(T* q, const T* axis, T angle) {
  T s, cs;
  half_sincos(angle * (T)0.5, s, cs);
  q[0] = cs; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
template <typename T> MJB_DEV void quat_integrate(T* q, const T* w, T h) {
  T ax[3] = {w[0], w[1], w[2]};
  T n = normalize3(ax);
  if (n < Num<T>::minval()) return;
  T qr[4], out[4];
  axisangle2quat(qr, ax, h * n);
  quat_mul(out, q, qr);
  quat_normalize(out);
  q[0] = out[0]; q[1] = out[1]; q[2] = out[2]; q[3] = out[3];
}
template <typename T> MJB_DEV void quat_sub(T* res, const T* qa, const T* qb) {
  T qn[4] = {qa[0], -qa[1], -qa[2], -qa[3]}, qd[4];
  quat_mul(qd, qn, qb);
  T ax[3] = {qd[1], qd[2], qd[3]};
  T s = t_sqrt(dot3(ax, ax));
  if (s < Num<T>::minval()) { res[0] = res[1] = res[2] = 0; return; }
  T ang = 2 * t_atan2(s, qd[0]);
  if (ang > MJB_PI) ang -= 2 * MJB_PI;
  T k = ang / s;
  res[0] = ax[0] * k; res[1] = ax[1] * k; res[2] = ax[2] * k;
}
template <typename T> MJB_DEV void inert_com(T* res, const T* inert, const T* mat, const T* dif, T mass) {
  T tmp[9];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int j = 0; j < 3; j++) tmp[3 * k + j] = inert[k] * mat[3 * j + k];
  res[0] = mat[0] * tmp[0] + mat[1] * tmp[3] + mat[2] * tmp[6];
  res[1] = mat[3] * tmp[1] + mat[4] * tmp[4] + mat[5] * tmp[7];
  res[2] = mat[6] * tmp[2] + mat[7] * tmp[5] + mat[8] * tmp[8];
  res[3] = mat[0] * tmp[1] + mat[1] * tmp[4] + mat[2] * tmp[7];
  res[4] = mat[0] * tmp[2] + mat[1] * tmp[5] + mat[2] * tmp[8];
  res[5] = mat[3] * tmp[2] + mat[4] * tmp[5] + mat[5] * tmp[8];
  res[0] += mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] += mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] += mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] -= mass * dif[0] * dif[1];
  res[4] -= mass * dif[0] * dif[2];
  res[5] -= mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2]; res[9] = mass;
}
template <typename T> MJB_DEV void mul_inert_vec(T* res, const T* i, const T* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
template <typename T> MJB_DEV void cross_motion(T* res, const T* vel, const T* v) {
  res[0] = -vel[2] * v[1] + vel[1] * v[2];
  res[1] = vel[2] * v[0] - vel[0] * v[2];
  res[2] = -vel[1] * v[0] + vel[0] * v[1];
  res[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  res[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  res[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
template <typename T> MJB_DEV void cross_force(T* res, const T* vel, const T* f) {
  res[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  res[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  res[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  res[3] = -vel[2] * f[4] + vel[1] * f[5];
  res[4] = vel[2] * f[3] - vel[0] * f[5];
  res[5] = -vel[1] * f[3] + vel[0] * f[4];
}

MJB_DEV int tri_at(int r, int c) { return r * (r + 1) / 2 + c; }   // packed lower-triangular index (c <= r)

// triangle enumeration idx -> (r, c), 0 <= c <= r, idx = r(r+1)/2 + c: host-built table (r << 16 | c)
template <typename TP> MJB_DEV void tri_rc(TP tab, int idx, int& r, int& c) {
  int v = tab[idx];
  r = v >> 16; c = v & 0xffff;
}

// ---------------------------------------------------------------------------
// dense Cholesky in LDS (lower triangle, in place) + solve.  dinv receives 1/L[j][j].
// ---------------------------------------------------------------------------
template <typename T, int G, typename TP> MJB_DEV void chol_factor(T* A, T* dinv, int n, int lane, TP tri) {
  // A: packed lower triangle (tri_at), factored in place
  for (int j = 0; j < n; j++) {
    gsync<G>();
    T ajj = A[tri_at(j, j)];
    T ljj = t_sqrt(t_max(ajj, Num<T>::minval()));
    T inv = 1 / ljj;
    gsync<G>();
    for (int i = j + 1 + lane; i < n; i += G) A[tri_at(i, j)] *= inv;
    if (lane == 0) { A[tri_at(j, j)] = ljj; dinv[j] = inv; }
    gsync<G>();
    int t = n - 1 - j, np = t * (t + 1) / 2;
    for (int idx = lane; idx < np; idx += G) {
      int r, c;
      tri_rc(tri, idx, r, c);
      int i = j + 1 + r, k = j + 1 + c;
      A[tri_at(i, k)] -= A[tri_at(i, j)] * A[tri_at(k, j)];
    }
  }
  gsync<G>();
}

// x (LDS vector, length n) <- (L L^T)^-1 x.   Uses one lane per row when G >= n.
template <typename T, int G> MJB_DEV void chol_solve(const T* L, const T* dinv, T* x, int n, int lane) {
  gsync<G>();
  if (G >= n) {
    T r = lane < n ? x[lane] : (T)0;
    for (int j = 0; j < n; j++) {                       // forward: L y = b
      T xj = gshfl<T, G>(r, j) * dinv[j];
      if (lane == j) r = xj;
      else if (lane > j && lane < n) r -= L[tri_at(lane, j)] * xj;
    }
    for (int j = n - 1; j >= 0; j--) {                  // backward: L^T x = y
      T xj = gshfl<T, G>(r, j) * dinv[j];
      if (lane == j) r = xj;
      else if (lane < j) r -= L[tri_at(j, lane)] * xj;
    }
    if (lane < n) x[lane] = r;
  } else {
    for (int j = 0; j < n; j++) {
      T xj = x[j] * dinv[j];
      gsync<G>();
      if (lane == 0) x[j] = xj;
      for (int i = j + 1 + lane; i < n; i += G) x[i] -= L[tri_at(i, j)] * xj;
      gsync<G>();
    }
    for (int j = n - 1; j >= 0; j--) {
      T xj = x[j] * dinv[j];
      gsync<G>();
      if (lane == 0) x[j] = xj;
      for (int i = lane; i < j; i += G) x[i] -= L[tri_at(j, i)] * xj;
      gsync<G>();
    }
  }
  gsync<G>();
}

// ---------------------------------------------------------------------------
// Register-tiled Cholesky for one wavefront per environment (G == 64).  The 64 lanes form an 8x8 grid; lane
// (lr, lc) keeps the entries A[8a+lr][8b+lc], a,b < NB, in registers for the whole factorisation.  Per column
// only the pivot and the scaled column cross lanes (through LDS): 2 syncs and no table lookups per column.
//   mode 0: A = M      mode 1: A = M + J^T diag(dw) J (dw = D on active rows)      mode 2: A = M + h diag(damping)
// Result: L (lower) in W, 1/diag(L) in dinv — the layout chol_solve expects.
// ---------------------------------------------------------------------------
template <typename T, int NB, typename MRef>
MJB_DEV void tile_factor(MRef m, const T* M, T* W, T* dinv, T* col, const T* J, const T* dw, int nefc, int mode, int n, int lane) {
  const int lr = lane >> 3, lc = lane & 7;
  T e[NB][NB];
#pragma unroll
  for (int a = 0; a < NB; a++)
#pragma unroll
    for (int b = 0; b < NB; b++) {
      int r = 8 * a + lr, cc = 8 * b + lc;
      e[a][b] = (r < n && cc <= r) ? M[r * n + cc] : (T)0;
    }
  if (mode == 1) {
    for (int row = 0; row < nefc; row++) {
      T d = dw[row];
      if (d == 0) continue;
      T jr[NB], jc[NB];
#pragma unroll
      for (int a = 0; a < NB; a++) { int r = 8 * a + lr; jr[a] = r < n ? d * J[row * n + r] : (T)0; }
#pragma unroll
      for (int b = 0; b < NB; b++) { int cc = 8 * b + lc; jc[b] = cc < n ? J[row * n + cc] : (T)0; }
#pragma unroll
      for (int a = 0; a < NB; a++)
#pragma unroll
        for (int b = 0; b < NB; b++) e[a][b] += jr[a] * jc[b];
    }
  } else if (mode == 2) {
    if (lr == lc) {
#pragma unroll
      for (int a = 0; a < NB; a++) { int r = 8 * a + lr; if (r < n) e[a][a] += m.timestep * m.dof_damping[r]; }
    }
  }
  for (int j = 0; j < n; j++) {
    const int bj = j >> 3, lj = j & 7;
    if (lr == lj && lc == lj) {
#pragma unroll
      for (int b = 0; b < NB; b++) if (b == bj) col[n] = e[b][b];
    }
    gsync<64>();
    const T ljj = t_sqrt(t_max(col[n], Num<T>::minval()));
    const T inv = 1 / ljj;
    if (lc == lj) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
        if (b != bj) continue;
#pragma unroll
        for (int a = 0; a < NB; a++) {
          int r = 8 * a + lr;
          if (r > j && r < n) { T v = e[a][b] * inv; e[a][b] = v; col[r] = v; }
          else if (r == j) e[a][b] = ljj;
        }
      }
    }
    if (lane == 0) dinv[j] = inv;
    gsync<64>();
    T rv[NB], cv[NB];
#pragma unroll
    for (int a = 0; a < NB; a++) { int r = 8 * a + lr; rv[a] = (r > j && r < n) ? col[r] : (T)0; }
#pragma unroll
    for (int b = 0; b < NB; b++) { int cc = 8 * b + lc; cv[b] = (cc > j && cc < n) ? col[cc] : (T)0; }
#pragma unroll
    for (int a = 0; a < NB; a++)
#pragma unroll
      for (int b = 0; b < NB; b++) e[a][b] -= rv[a] * cv[b];
  }
#pragma unroll
  for (int a = 0; a < NB; a++)
#pragma unroll
    for (int b = 0; b < NB; b++) {
      int r = 8 * a + lr, cc = 8 * b + lc;
      if (r < n && cc <= r) W[tri_at(r, cc)] = e[a][b];
    }
  gsync<64>();
}

// ---------------------------------------------------------------------------
// environment context: LDS slice + model, passed to every phase
// ---------------------------------------------------------------------------
// Re-materialise the model / layout pointers at the start of every phase: the (invariant) loads through
// them then stay local to the phase instead of being hoisted to the kernel entry, where ~190 live SGPRs
// were spilled to VGPR lanes (17 % of the instruction stream was v_readlane / v_writelane).
// Per-model specialisation (mjb_spec_source / mjb_spec_load): the generated translation unit defines these two macros as
// lists of __builtin_assume(field == value) over the structural sizes of the model and the LDS layout offsets; sizes,
// trip counts and offsets then fold to constants in every phase (same source, same arithmetic).  Empty in the generic build.
#ifndef MJB_SPEC_ASSUME
#define MJB_SPEC_ASSUME(m)
#endif
#ifndef MJB_SPEC_ASSUME_LAY
#define MJB_SPEC_ASSUME_LAY(L)
#endif
// Baked model (specialised translation units, mjb_model_spec_source): MJB_SPEC_BAKED names a `__constant__` image of the model's
// DevModel<float> whose table pointers point at `__constant__` arrays of the SAME translation unit, so a table read is ONE access at
// a link-time address (no pointer hop through the device copy of DevModel) and reads with compile-time indices fold to literals.
// The run-time options (disableactuator, iterations, tolerance) are never baked: MJB_OPT reads them from the device copy.
#if defined(MJB_SPEC_BAKED) && !defined(MJB_HOST_EMU)
#define MJB_MODEL_OF(p) MJB_SPEC_BAKED
#else
#define MJB_MODEL_OF(p) (*(p))
#endif
#define MJB_OPT(c, f) ((*(c).mp).f)
#ifdef MJB_HOST_EMU
#define MJB_ENV(c) ModelRef<T> m = *(c).mp; LayRef L = *(c).lp
#else
#define MJB_ENV(c)                                     \
  auto mp_ = (c).mp; auto lp_ = (c).lp;                \
  asm volatile("" : "+s"(mp_), "+s"(lp_));             \
  ModelRef<T> m = MJB_MODEL_OF(mp_); LayRef L = *lp_;  \
  MJB_SPEC_ASSUME(m) MJB_SPEC_ASSUME_LAY(L)
#endif
#if defined(MJB_PROFILE) && !defined(MJB_HOST_EMU)
#define MJB_STAMP(c, k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); (c).pacc[k] += t_ - (c).pt; (c).pt = t_; } while (0)
#else
#define MJB_STAMP(c, k) ((void)0)
#endif
template <typename T> using ModelRef = const DevModel<T> MJB_CONST&;
typedef const Lay MJB_CONST& LayRef;

template <typename T> struct Ctx {
  bool skip_dynamics = false;   // inverse-dynamics mode: forward() stops before actuation / constraint solve
  const DevModel<T> MJB_CONST* mp;
  const Lay MJB_CONST* lp;
  T* w;      // T region of this environment's LDS slice
  int* wi;   // int region
  int lane;
  int ncon, nefc, niter, con_dropped, efc_dropped;
#if defined(MJB_PHASE_REPEAT) && !defined(MJB_HOST_EMU)
  int rep = -1;                 // index (PH_*) of the phase that runs twice (diagnostic build)
#endif
#if defined(MJB_PROFILE) && !defined(MJB_HOST_EMU)
  unsigned long long pacc[PH_N] = {};
  unsigned long long pt = 0;
#endif
  MJB_DEVM Ctx(const DevModel<T> MJB_CONST* m_, const Lay MJB_CONST* L_, T* w_, int* wi_, int lane_) : mp(m_), lp(L_), w(w_), wi(wi_), lane(lane_), ncon(0), nefc(0), niter(0), con_dropped(0), efc_dropped(0) {}
};

// Diagnostic build -DMJB_PHASE_REPEAT: how many times an idempotent piece of the solver runs (ids 32.. name the pieces; 1 in the product)
#if defined(MJB_PHASE_REPEAT) && !defined(MJB_HOST_EMU)
#define MJB_REP_N(c, id) ((c).rep == (id) ? 2 : 1)
#else
#define MJB_REP_N(c, id) 1
#endif
enum { REP_CHOL = 32, REP_REUSE = 33, REP_MV = 34, REP_LS = 35, REP_WARM = 36, REP_GRAD = 37, REP_CTRL = 38 };

// ---------------------------------------------------------------------------
// Row-per-lane Cholesky in registers (G == 64, n <= 32): lane i keeps row i (padded to 32 with identity) in 32
// VGPRs; a column step is one v_readlane + one FMA per trailing column — no LDS, no sync, no branches.
// Entries above the diagonal hold garbage that is never read.  Modes as tile_factor.
// ---------------------------------------------------------------------------
template <typename T, typename MRef>
MJB_DEV void reg_factor32(MRef m, const T* M, T* W, T* dinv, const T* J, const T* dw, int nefc, int mode, int n, int lane, T* x) {
  T a[32];
  const bool own = lane < n;
#pragma unroll
  for (int k = 0; k < 32; k++) a[k] = (own && k < n) ? M[lane * n + k] : ((k == lane) ? (T)1 : (T)0);
  if (mode == 1) {
    for (int row = 0; row < nefc; row++) {
      T d = dw[row];
      if (d == 0) continue;
      T jrow = own ? J[row * n + lane] : (T)0;
      T s = d * jrow;
#pragma unroll
      for (int k = 0; k < 32; k++) a[k] += s * rdlane_f(jrow, k);
    }
  } else if (mode == 2) {
    T dd = own ? m.timestep * m.dof_damping[lane] : (T)0;
#pragma unroll
    for (int k = 0; k < 32; k++) if (k == lane) a[k] += dd;
  }
  T myinv = 1;
  T r = (x && own) ? x[lane] : (T)0;          // fused forward substitution L y = b (b = x on entry)
#pragma unroll
  for (int j = 0; j < 32; j++) {
    T ajj = t_max(rdlane_f(a[j], j), Num<T>::minval());
    T inv = t_rsqrt(ajj);                       // 1/L[j][j]; fp32: v_rsq_f32 + one Newton step
    T ljj = ajj * inv;
    a[j] = lane == j ? ljj : a[j] * inv;
    if (lane == j) myinv = inv;
    T yj = rdlane_f(r, j) * inv;
    r = lane > j ? r - a[j] * yj : (lane == j ? yj : r);
#pragma unroll
    for (int k = j + 1; k < 32; k++) a[k] -= a[j] * rdlane_f(a[j], k);
  }
  if (own) {
#pragma unroll
    for (int k = 0; k < 32; k++) if (k <= lane) W[tri_at(lane, k)] = a[k];
    dinv[lane] = myinv;
  }
  gsync<64>();
  if (x) {                                      // backward substitution L^T x = y: row j of L comes back from LDS
    T lrow[32];
#pragma unroll
    for (int j = 0; j < 32; j++) lrow[j] = (j < n && lane < j) ? W[tri_at(j, lane)] : (T)0;
#pragma unroll
    for (int j = 31; j >= 0; j--) {
      T xj = rdlane_f(r, j) * rdlane_f(myinv, j);
      r = lane == j ? xj : r - lrow[j] * xj;
    }
    if (own) x[lane] = r;
    gsync<64>();
  }
}

// ---------------------------------------------------------------------------
// MFMA Cholesky (fp32, G == 64, n <= 32): the symmetric 32x32 matrix lives in the accumulator layout of
// v_mfma_f32_32x32x2_f32 (lane l: column l%32, 16 rows 8q + 4(l/32) + t).  By symmetry "row j across lanes" IS
// column j of the factor, so a panel of two columns costs a handful of VALU ops and the rank-2 trailing update
// of the whole matrix is ONE MFMA (exact fp32 FMAs).  The Hessian M + J^T D J is assembled the same way:
// one MFMA per pair of constraint rows.  Forward substitution is fused; L goes to LDS (packed) for the
// backward substitution and for reuse when the active set does not change.
// ---------------------------------------------------------------------------
#ifndef MJB_PANEL4
#define MJB_PANEL4 0             // experiment (round 3, off): FOUR pivots per panel in the MFMA sweep inverse instead of two.  Measured in
#endif                           // scripts/micro/factor_bench (profiles/r03_panel4_microbench.log): panel loop 4682 -> 4374 cycles alone, 5210 -> 4595 with two
                                 // waves per SIMD, same residual - about 1 % of a step.  Not adopted: a wave issues one VALU instruction per >= 4 cycles
                                 // whatever it does, and the 4x4 scalar elimination (10 v_readlane, 4 rcp, 20 dependent FMAs) costs as many instructions
                                 // per pivot as two 2x2 ones; halving the MFMA round trips alone buys little, and the rounding of every solve would change.
#ifndef MJB_SWEEP_EXCLUDE
#define MJB_SWEEP_EXCLUDE 1      // factor_W mode that keeps the Cholesky path (1 = Hessian); -1: sweep everywhere
#endif
#ifndef MJB_HOST_EMU
MJB_DEV float half_bcast(float v, int half) {     // value of the given 32-lane half, column-aligned, in all 64 lanes
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(half == 0 ? p[0] : p[1]);
}
// v_permlane32_swap: a' = [a.lo | b.lo], b' = [a.hi | b.hi] (lo = lanes 0..31, hi = lanes 32..63)
MJB_DEV void half_swap(float a, float b, float& ao, float& bo) {
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  ao = __uint_as_float(p[0]); bo = __uint_as_float(p[1]);
}
// both 32-lane halves of v, each column-aligned in all 64 lanes, from ONE v_permlane32_swap (lo = lanes 0..31, hi = lanes 32..63)
MJB_DEV void half_bcast2(float v, float& lo, float& hi) {
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  lo = __uint_as_float(p[0]); hi = __uint_as_float(p[1]);
}
// v(lane) + v(lane ^ 32): the sum over the two 32-lane halves, in all 64 lanes
MJB_DEV float half_sum(float v) {
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
MJB_DEV double half_sum(double v) {
  auto ph = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  auto pl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  return __hiloint2double((int)ph[0], (int)pl[0]) + __hiloint2double((int)ph[1], (int)pl[1]);
}
#endif
// Backward substitution L^T x = y, 8 rows at a time.  rs holds x pre-scaled by the lane's own 1/L_cc, so the serial chain
// per row is one v_readlane + one FMA; the next block's rows are loaded while the current chain runs.
MJB_DEV void mfma_back_load8(float (&dst)[8], const float* Wc, int blk, int n) {
  if (8 * blk + 8 <= n) {                                       // every row exists: immediate offsets
#pragma unroll
    for (int t = 0; t < 8; t++) dst[t] = Wc[tri_at(8 * blk + t, 0)];     // L[j][c]; lanes c >= j read past the row: masked later
  } else {
#pragma unroll
    for (int t = 0; t < 8; t++) { int j = 8 * blk + t, jj = j < n ? j : n - 1; dst[t] = Wc[tri_at(jj, 0)]; }   // uniform clamp
  }
}
MJB_DEV void mfma_back_chain8(float (&lrow)[8], int blk, int n, int c_, float myinv, float& rs) {
  int c = c_;
  MJB_OPAQUE9(c, lrow[0], lrow[1], lrow[2], lrow[3], lrow[4], lrow[5], lrow[6], lrow[7]);
#pragma unroll
  for (int t = 0; t < 8; t++) { const int j = 8 * blk + t; lrow[t] = (c < j && j < n) ? lrow[t] * myinv : 0.0f; }
#pragma unroll
  for (int t = 7; t >= 0; t--) rs -= lrow[t] * rdlane_f(rs, 8 * blk + t);   // rows >= n: lrow == 0 and x_j == 0
}

// L^T x = y from the packed factor in LDS (lane c -> x_c, replicated in both halves); y in r, 1/L_cc in myinv
MJB_DEV float mfma_backward32(const float* W, int n, int c_, float myinv, float r) {
  float rs = r * myinv;
  const float* Wc = W + c_;
  float cur[8], nxt[8];
  const int top = (n - 1) >> 3;
  mfma_back_load8(cur, Wc, top, n);
#pragma unroll
  for (int blk = 3; blk >= 0; blk--) {
    if (blk > top) continue;
    if (blk > 0) mfma_back_load8(nxt, Wc, blk - 1, n);
    mfma_back_chain8(cur, blk, n, c_, myinv, rs);
#pragma unroll
    for (int t = 0; t < 8; t++) cur[t] = nxt[t];
  }
  return rs;
}
// Solve with the factor that is already in LDS (Newton iterations whose active set did not change): forward substitution
// in the same style - lane c keeps r_c / L_cc, so column j costs one v_readlane + one FMA; L[c][j] comes from the lane's own
// row (row base + immediate j), 8 columns per batch of loads - then the backward substitution above.
MJB_DEV void mfma_solve32(const float* W, const float* dinv, float* x, int n, int lane) {
  const int h = lane >> 5, c_ = lane & 31;
  const float myinv = c_ < n ? dinv[c_] : 1.0f;
  float rs = (c_ < n ? x[c_] : 0.0f) * myinv;
  const float* Wr = W + tri_at(c_ < n ? c_ : 0, 0);
  const int top = (n - 1) >> 3;
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {
    if (blk > top) continue;
    float lrow[8];
    int c = c_;
    MJB_OPAQUE1(c);
#pragma unroll
    for (int t = 0; t < 8; t++) lrow[t] = Wr[8 * blk + t];      // L[c][j]; j >= c reads past the row: masked below
#pragma unroll
    for (int t = 0; t < 8; t++) { const int j = 8 * blk + t; lrow[t] = (j < c && c < n) ? lrow[t] * myinv : 0.0f; }
#pragma unroll
    for (int t = 0; t < 8; t++) rs -= lrow[t] * rdlane_f(rs, 8 * blk + t);
  }
  // rs_c = y_c (forward-substituted, already divided by L_cc): the backward pass wants y_c and multiplies by 1/L_cc itself
  float r = mfma_backward32(W, n, c_, myinv, rs);
  if (h == 0 && c_ < n) x[c_] = r;
  gsync<64>();
}
// ---- round 3: the accumulator in ELIMINATION ORDER -------------------------------------------------------------------------------
// The hardware fixes which accumulator row a (register, half) pair holds: register 4q+t, half h <-> row 8q+4h+t.  Which MATRIX index
// sits on accumulator row / column rho is ours to choose, as long as rows and columns use the same map (a symmetric relabelling).
// With  pos(rho) = 8q + 2t + h  register i holds matrix rows 2i (lanes 0..31) and 2i+1 (lanes 32..63): the two pivot columns of panel
// jb ARE register jb, already stacked the way the rank-2 MFMA wants its operands - no half broadcasts of two registers, no select.
// Lane (h, c) stands for matrix column pos(c); everything the lanes index (rows of M, J, the packed factor) goes through pos(c) once,
// outside the panel loop.  The factor that lands in LDS is the ordinary Cholesky factor in natural order (only lanes were relabelled).
MJB_DEVM constexpr int acc_pos(int c) { return (c & 24) | ((c & 3) << 1) | ((c >> 2) & 1); }       // matrix index of accumulator row / column c
MJB_DEVM constexpr int acc_lane(int p) { return (p & 24) | ((p & 1) << 2) | ((p >> 1) & 3); }      // its inverse
// Second change: the right-hand side rides along as row / column n of the matrix (n < 32).  Cholesky of [A b; b^T 1] has y^T = (L^-1 b)^T
// as row n of its factor, so the forward substitution IS the trailing update (the same MFMA) - no per-panel v_readlane / FMA / select
// chain for it.  y is stored like any row of the factor (to ybuf) by the lane that stands for column n.
// Third: nothing is masked on the way into the MFMA.  Columns already eliminated hold rounding residue instead of zeros; it only ever
// feeds dead rows / columns (operand lane rho touches accumulator row rho, operand lane gamma column gamma).  Only the stores to LDS
// are predicated.  The factor L is bitwise the one of the round-2 form (same pivots, same FMAs on the live entries).
template <typename MRef>
MJB_DEV void mfma_factor32(MRef m, const float* M, float* W, float* dinv, float* ybuf, const float* J, const float* dw, int nefc, int mode, int n, int lane, float* x, unsigned long long* pf = nullptr) {
  const int h = lane >> 5, c_ = lane & 31, pc = acc_pos(c_);
  unsigned long long tq0 = pf ? MJB_MEMTIME() : 0;
  const bool aug = x != nullptr && n < 32;                     // room for the right-hand side as column n
  const bool cin = pc < n, isb = aug && pc == n;
  mjb_f16v acc;
  {
    // acc[i] of lane (h, c) = A[2i+h][pos(c)] = A[pos(c)][2i+h]: every lane reads along its own row of M (the lane of column n along
    // the right-hand side), one address register and immediate offsets.  Rows / columns beyond are padded with identity.
    const float* Mc = (isb ? x : M + (cin ? pc : 0) * n) + h;
    const float xv = (aug && cin) ? x[pc] : 0.0f;
    const bool colok = cin || isb;
    // the lane's diagonal entry (row 2i + h == pos(c)) sits in ONE register, idg: + h D there (mode 2; identity on the padding), or the
    // identity padding alone.  The damping is loaded ONCE, in front of the loop (inside it the compiler re-loaded it per register).
    const int dq = pc - h, idiag = (dq & 1) ? -1 : dq >> 1;
    const float hd = mode == 2 && cin ? m.timestep * m.dof_damping[pc] : 1.0f;
    int idg = (mode == 2 || !cin) ? idiag : -1;
    MJB_OPAQUE1(idg);                             // keep the sixteen compares here (hoisted out of the step loop they live as spilled lane masks)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r0 = 2 * i;                                    // this register holds row r0 + h
      const float mv = Mc[r0];                                 // may read past the row when r0 + h >= n: masked
      float v;
      if (r0 + 1 < n) v = colok ? mv : 0.0f;                   // both rows exist: one shared lane mask
      else {
        v = (colok && r0 + h < n) ? mv : 0.0f;
        if (aug && r0 + h == n && cin) v = xv;                 // row n = the right-hand side
      }
      if (mode == 2 || r0 + 1 >= n) v += i == idg ? hd : 0.0f;
      acc[i] = v;
    }
    if (mode == 1) {
      // Hessian M + J^T D J: one rank-2 MFMA per PAIR OF ACTIVE ROWS (D != 0), the next pair's J loads in flight
      // while the current MFMA runs.
      const int cm = cin ? pc : 0;
      for (int base = 0; base < nefc; base += 64) {
        const int rix = base + lane;
        const float dl = rix < nefc ? dw[rix] : 0.0f;
        unsigned long long act = MJB_BALLOT(dl != 0.0f);
        float jcur = 0.0f, dcur = 0.0f;
        bool have = false;
        while (act) {
          int ra = __builtin_ctzll(act); act &= act - 1;
          int rb = ra; float dB = 0.0f;
          if (act) { rb = __builtin_ctzll(act); act &= act - 1; dB = rdlane_f(dl, rb); }
          float dA = rdlane_f(dl, ra);
          int rr = h == 0 ? ra : rb;
          float jn = J[(base + rr) * n + cm];
          float dn = h == 0 ? dA : dB;
          if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
          jcur = cin ? jn : 0.0f; dcur = dn; have = true;
        }
        if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
      }
    }
  }
  float* const wrowp = (isb ? ybuf : W + tri_at(cin ? pc : 0, 0)) + h;     // lane (h, c) stores entry (pos(c), j0 + h): the column rides in the immediate offset
  // lane (h, c) stores the entries (pos(c), j0 + h) with pos(c) >= j0 + h, up to the last stored row: ONE compare per panel against
  // qs = pos(c) - h (lanes beyond the last stored row never pass)
  int qs = pc <= (aug ? n : n - 1) ? pc - h : -64;
  int dofs = 0;
  MJB_OPAQUE1(dofs);                                          // dinv's address in a register for the whole loop (not re-materialised per panel)
  float* const dinvp = dinv + dofs;
  unsigned long long tq1 = pf ? MJB_MEMTIME() : 0;
#pragma unroll
  for (int jb = 0; jb < 16; jb++) {
    const int j0 = 2 * jb, j1 = j0 + 1, l0 = acc_lane(j0), l1 = l0 + 4;
    if (j0 >= n) continue;                                    // padded (identity) columns: nothing to eliminate (uniform skip)
    MJB_OPAQUE1(qs);                              // keep the per-panel lane compares in the loop (cheaper than hoisted, spilled masks)
    // S: columns j0 (lanes 0..31) and j1 (lanes 32..63) of the trailing matrix; the 2x2 pivot block [a b; b d] is eliminated in
    // scalars so that the serial chain per panel is rsq -> fma -> rsq
    const float S = acc[jb];
    const float a = __builtin_fmaxf(rdlane_f(S, l0), Num<float>::minval()), b = rdlane_f(S, l1), d = rdlane_f(S, 32 + l1);
    const float inv0 = MJB_RSQF(a), bia = b * inv0 * inv0;      // v_rsq_f32: 1 ulp, no refinement step on the chain
    const float d1 = __builtin_fmaxf(d - b * bia, Num<float>::minval());
    const float inv1 = j1 < n ? MJB_RSQF(d1) : 0.0f;           // n odd: the last panel has one real pivot
    // lanes 0..31: L[.][j0] = v0 inv0;  lanes 32..63: L[.][j1] = (v1 - bia v0) inv1.  The half swap against a zero register hands the
    // upper lanes both columns (z0 = [0 | v0], z1 = [0 | v1]) and clobbers S - rows j0, j1 are dead from here on - so the panel needs
    // one swap and no copy of S; the lower lanes get their product before the swap, and (0 - bia 0) inv1 + x = x leaves it exact.
    const float avlo = S * (h == 0 ? inv0 : 0.0f);
    float z0, z1;
    half_swap(0.0f, S, z0, z1);
    acc[jb] = z1;
    const float av = (z1 - bia * z0) * inv1 + avlo;
    if (qs >= j0) wrowp[j0] = av;                               // packed factor (and y) to LDS for the backward substitution / later reuse
    if (lane == 0) { dinvp[j0] = inv0; dinvp[j1] = inv1; }      // 1 / L_jj (uniform values: one lane, one ds_write2)
    acc = MJB_MFMA(-av, av, acc);                               // rank-2 trailing update of the whole matrix
  }
  unsigned long long tq2 = pf ? MJB_MEMTIME() : 0;
  gsync<64>();
  if (x) {
    if (aug) {                                                // backward substitution L^T x = y from the packed factor in LDS
      const float r = c_ < n ? ybuf[c_] : 0.0f, myinv = c_ < n ? dinv[c_] : 1.0f;
      float rs = mfma_backward32(W, n, c_, myinv, r);
      if (h == 0 && c_ < n) x[c_] = rs;
      gsync<64>();
    } else mfma_solve32(W, dinv, x, n, lane);                  // n == 32: no spare column, both substitutions from LDS
  }
  if (pf) { unsigned long long tq3 = MJB_MEMTIME(); pf[0] += tq1 - tq0; pf[1] += tq2 - tq1; pf[2] += tq3 - tq2; }
}
// the round-2 form (natural lane order, fused forward substitution in registers): kept for A/B runs (-DMJB_R2_FACTOR)
template <typename MRef>
MJB_DEV void mfma_factor32_r2(MRef m, const float* M, float* W, float* dinv, const float* J, const float* dw, int nefc, int mode, int n, int lane, float* x, unsigned long long* pf = nullptr) {
  const int h = lane >> 5, c_ = lane & 31;
  unsigned long long tq0 = pf ? MJB_MEMTIME() : 0;
  mjb_f16v acc;
  {
    // acc[4q+t] of lane (h, c) = A[8q+4h+t][c] = A[c][8q+4h+t] (symmetric): every lane reads along its own row of M,
    // so the 16 LDS reads use one address register and immediate offsets.  Rows/columns >= n are padded with identity.
    const int c = c_;
    const bool cin = c < n;
    const float* Mc = M + (cin ? c : 0) * n + 4 * h;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int k = 8 * (i >> 2) + (i & 3);                    // this register holds row k + 4 h
      const float mv = Mc[k];                                  // may read past row c when the row is >= n: masked
      // Round 3: one SHARED lane mask (c < n) for the registers whose rows exist in both halves (k + 4 < n: a uniform test, a
      // compile-time one in the specialised kernel); only the last registers need the per-row test, and only they can hold the
      // identity padding.  (The round-2 form built two lane masks per register - 32 SGPR pairs the compiler hoisted out of the step
      // loop and kept spilled in VGPR lanes: two v_readlane + a hazard nop per use.)  Same values bit for bit.
      float v;
      if (k + 4 < n) v = cin ? mv : 0.0f;
      else v = (cin && k + 4 * h < n) ? mv : 0.0f;
      if (mode == 2) { if (k + 4 * h == c) v += cin ? m.timestep * m.dof_damping[c] : 1.0f; }      // + h D on the diagonal (identity on the padding)
      else if (k + 4 >= n) { if (k + 4 * h == c && !cin) v += 1.0f; }                              // identity padding of rows / columns >= n
      acc[i] = v;
    }
    if (mode == 1) {
      // Hessian M + J^T D J: one rank-2 MFMA per PAIR OF ACTIVE ROWS (D != 0), the next pair's J loads in flight
      // while the current MFMA runs.
      const int cm = c < n ? c : 0;
      for (int base = 0; base < nefc; base += 64) {
        const int rix = base + lane;
        const float dl = rix < nefc ? dw[rix] : 0.0f;
        unsigned long long act = MJB_BALLOT(dl != 0.0f);
        float jcur = 0.0f, dcur = 0.0f;
        bool have = false;
        while (act) {
          int ra = __builtin_ctzll(act); act &= act - 1;
          int rb = ra; float dB = 0.0f;
          if (act) { rb = __builtin_ctzll(act); act &= act - 1; dB = rdlane_f(dl, rb); }
          float dA = rdlane_f(dl, ra);
          int rr = h == 0 ? ra : rb;
          float jn = J[(base + rr) * n + cm];
          float dn = h == 0 ? dA : dB;
          if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
          jcur = c < n ? jn : 0.0f; dcur = dn; have = true;
        }
        if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
      }
    }
  }
  float r = (x && c_ < n) ? x[c_] : 0.0f;                     // RHS replicated in both halves (lane -> row c)
  float myinv = 1.0f;
  const int wrow = tri_at(c_, 0), cvalid = (h == 0 && c_ < n) ? 0 : -1;
  float* const wrowp = W + wrow;
  float* const dumpp = dinv + n;                              // one spare word behind dinv swallows the masked stores
  unsigned long long tq1 = pf ? MJB_MEMTIME() : 0;
#pragma unroll
  for (int jb = 0; jb < 16; jb++) {
    const int j0 = 2 * jb, j1 = j0 + 1, hj = (j0 >> 2) & 1, ij = 4 * (j0 >> 3) + (j0 & 3);
    if (j0 >= n) continue;                                    // padded (identity) columns: nothing to eliminate (uniform skip)
    int c = c_;
    MJB_OPAQUE1(c);                               // keep the per-column lane compares in the loop (cheaper than spilled masks)
    // columns j0, j1 of the trailing matrix, one entry per lane; the 2x2 pivot block [a b; b d] is eliminated in scalars
    // so that the serial chain per panel is rsq -> fma -> rsq
    float v0 = half_bcast(acc[ij], hj), v1r = half_bcast(acc[ij + 1], hj);
    float a = __builtin_fmaxf(rdlane_f(v0, j0), Num<float>::minval()), b = rdlane_f(v0, j1), d = rdlane_f(v1r, j1);
    float inv0 = MJB_RSQF(a), bia = b * inv0 * inv0;            // v_rsq_f32: 1 ulp, no refinement step on the chain
    float d1 = __builtin_fmaxf(d - b * bia, Num<float>::minval());
    float inv1 = MJB_RSQF(d1);
    float v1 = v1r - bia * v0;
    // lane j0 of v0 IS the pivot a (lane j1 of v1 is d1), so the diagonal needs no case of its own: one compare + select per column
    // instead of two (bitwise the same unless a pivot fell below the 1e-15 clamp, which an SPD Hessian >= M never does)
    float L0 = c >= j0 ? v0 * inv0 : 0.0f;
    float L1 = c >= j1 ? v1 * inv1 : 0.0f;
    myinv = c == j0 ? inv0 : (c == j1 ? inv1 : myinv);
    // fused forward substitution
    float y0 = rdlane_f(r, j0) * inv0;
    float y1 = (rdlane_f(r, j1) - b * inv0 * y0) * inv1;
    r = c > j1 ? r - L0 * y0 - L1 * y1 : (c == j1 ? y1 : (c == j0 ? y0 : r));
    // packed factor to LDS for the backward substitution / later reuse; masked lanes write the dump word
    int cc = c | cvalid;
    float* p0 = cc >= j0 ? wrowp : dumpp - j0;                // select between two lane-constant pointers: the column index
    float* p1 = cc >= j1 ? wrowp : dumpp - j1;                // rides in the DS instruction's immediate offset
    p0[j0] = L0;
    p1[j1] = L1;
    // rank-2 trailing update of the whole matrix: acc -= [L0 L1] [L0 L1]^T
    float av = h == 0 ? L0 : L1;
    acc = MJB_MFMA(-av, av, acc);
  }
  unsigned long long tq2 = pf ? MJB_MEMTIME() : 0;
  if (h == 0 && c_ < n) dinv[c_] = myinv;
  gsync<64>();
  if (x) {                                                    // backward substitution L^T x = y from the packed factor in LDS
    float rs = mfma_backward32(W, n, c_, myinv, r);
    if (h == 0 && c_ < n) x[c_] = rs;
    gsync<64>();
  }
  if (pf) { unsigned long long tq3 = MJB_MEMTIME(); pf[0] += tq1 - tq0; pf[1] += tq2 - tq1; pf[2] += tq3 - tq2; }
}
// x <- A^-1 x for A = M (mode 0), M + J^T D_active J (mode 1, D in dw) or M + h diag(damping) (mode 2), n <= 32, fp32.
// The symmetric matrix lives in one 32x32 MFMA accumulator (lane (h,c): column c, rows 8q+4h+t) and is inverted in place
// by the symmetric SWEEP operator, two pivots per v_mfma_f32_32x32x2_f32:
//     B_ij <- B_ij - U_i P^-1 U_j^T,   B_iP <- U_i P^-1,   B_PP <- -P^-1        (U = columns j0,j1;  P = their 2x2 pivot block)
// With U' = U - [e_j0 e_j1] the MFMA operands  A_op = -U' P^-1,  B_op = U'^T  produce the pivot rows/columns as well; only the
// two pivot diagonal entries need a "-2" fix-up.  After all pivots acc = -A^-1 and the solve is a 16-FMA mat-vec per lane:
// no factor in LDS, no forward/backward substitution chains.  The matrix stays symmetric, so "column j as a lane vector"
// is one register (+ a half swap) at every step.
// Two halves, so that the two-wave step kernel (k_step2) can invert M + h D while the other wave still solves the constraints:
// mfma_sweep_invert32 leaves -A^-1 in the accumulator, mfma_sweep_apply32 is the mat-vec.  mfma_sweep_solve32 = both.
#ifndef MJB_R2_FACTOR
// Round 3: the accumulator in elimination order (acc_pos above): the two pivot columns of panel jb are register jb, stacked
// [u0 | u1] - which is the B operand U'^T itself once the two pivot lanes have had their 1 subtracted (and the register that goes on
// living is that minus the same indicator again: the "-2" of the pivot diagonals).  One half swap hands every lane both columns for
// the A operand.  acc = -A^-1 comes out in the same layout: register i, lane (h, c) = -A^-1[2i+h][pos(c)].
template <typename MRef>
MJB_DEV mjb_f16v mfma_sweep_invert32(MRef m, const float* M, const float* J, const float* dw, int nefc, int mode, int n, int lane, unsigned long long* pf = nullptr) {
  const int h = lane >> 5, c_ = lane & 31, pc = acc_pos(c_);
  unsigned long long tq0 = pf ? MJB_MEMTIME() : 0;
  const bool cin = pc < n;
  mjb_f16v acc;
  {
    const float* Mc = M + (cin ? pc : 0) * n + h;
    // the lane's diagonal entry (row 2i + h == pos(c)) sits in ONE register, idg: + h D there (mode 2; identity on the padding), or the
    // identity padding alone.  The damping is loaded ONCE, in front of the loop: inside it the compiler re-loaded it per register -
    // sixteen dependent global loads in a row, which made the Euler solve the slowest 470 instructions of the step.
    const int dq = pc - h, idiag = (dq & 1) ? -1 : dq >> 1;
    const float hd = mode == 2 && cin ? m.timestep * m.dof_damping[pc] : 1.0f;
    int idg = (mode == 2 || !cin) ? idiag : -1;
    MJB_OPAQUE1(idg);                             // keep the sixteen compares here (hoisted out of the step loop they live as spilled lane masks)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r0 = 2 * i;                                    // this register holds row r0 + h
      const float mv = Mc[r0];                                 // may read past the row when r0 + h >= n: masked
      float v;
      if (r0 + 1 < n) v = cin ? mv : 0.0f;                     // both rows exist: one shared lane mask
      else v = (cin && r0 + h < n) ? mv : 0.0f;
      if (mode == 2 || r0 + 1 >= n) v += i == idg ? hd : 0.0f;
      acc[i] = v;
    }
    if (mode == 1) {
      const int cm = cin ? pc : 0;
      for (int base = 0; base < nefc; base += 64) {
        const int rix = base + lane;
        const float dl = rix < nefc ? dw[rix] : 0.0f;
        unsigned long long act = MJB_BALLOT(dl != 0.0f);
        float jcur = 0.0f, dcur = 0.0f;
        bool have = false;
        while (act) {
          int ra = __builtin_ctzll(act); act &= act - 1;
          int rb = ra; float dB = 0.0f;
          if (act) { rb = __builtin_ctzll(act); act &= act - 1; dB = rdlane_f(dl, rb); }
          float dA = rdlane_f(dl, ra);
          int rr = h == 0 ? ra : rb;
          float jn = J[(base + rr) * n + cm];
          float dn = h == 0 ? dA : dB;
          if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
          jcur = cin ? jn : 0.0f; dcur = dn; have = true;
        }
        if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
      }
    }
  }
  unsigned long long tq1 = pf ? MJB_MEMTIME() : 0;
  int qs = pc - h;                                            // == j0 exactly on the two pivot lanes of a panel: (lower, column j0), (upper, column j1)
#pragma unroll
  for (int jb = 0; jb < 16; jb++) {
    const int j0 = 2 * jb, l0 = acc_lane(j0), l1 = l0 + 4;
    if (j0 >= n) continue;                                    // padded (identity) columns: nothing to eliminate (uniform skip)
    MJB_OPAQUE1(qs);                              // keep the per-panel lane compare in the loop (cheaper than hoisted, spilled masks)
    const float S = acc[jb];
    const float a = __builtin_fmaxf(rdlane_f(S, l0), Num<float>::minval()), b = rdlane_f(S, l1), d = rdlane_f(S, 32 + l1);
    const float det = __builtin_fmaxf(a * d - b * b, a * Num<float>::minval());
    const float rdet = MJB_RCPF(det);
    const float n00 = -d * rdet, n01 = b * rdet, n11 = -a * rdet;     // -P^-1: the A operand is -U' P^-1, the signs ride in the products
    const float e = qs == j0 ? 1.0f : 0.0f;
    const float bop = S - e;                                    // U'^T: [u0 - e_j0 | u1 - e_j1]
    acc[jb] = bop - e;                                          // the -2 of the two pivot diagonals
    float u0p, u1p;
    half_bcast2(bop, u0p, u1p);
    const float q0 = h == 0 ? n00 : n01, q1 = h == 0 ? n01 : n11;
    const float aop = u0p * q0 + u1p * q1;
    acc = MJB_MFMA(aop, bop, acc);
  }
  if (pf) { unsigned long long tq2 = MJB_MEMTIME(); pf[0] += tq1 - tq0; pf[1] += tq2 - tq1; }
  return acc;
}
// x <- A^-1 x with acc = -A^-1 from mfma_sweep_invert32 (bpad: 32 words of LDS for the zero-padded right-hand side)
MJB_DEV void mfma_sweep_apply32(const mjb_f16v& acc, float* bpad, int n, int lane, float* x, unsigned long long* pf = nullptr) {
  const int h = lane >> 5, c_ = lane & 31, pc = acc_pos(c_);
  unsigned long long tq2 = pf ? MJB_MEMTIME() : 0;
  if (h == 0) bpad[c_] = c_ < n ? x[c_] : 0.0f;               // right-hand side, zero-padded to 32
  gsync<64>();
  {
    const float* bp = bpad + h;                               // register i of half h multiplies b[2i + h]
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) { s0 += acc[t] * bp[2 * t]; s1 += acc[4 + t] * bp[8 + 2 * t]; s2 += acc[8 + t] * bp[16 + 2 * t]; s3 += acc[12 + t] * bp[24 + 2 * t]; }
    float s = (s0 + s1) + (s2 + s3);
    float tot = half_sum(s);
    if (h == 0 && pc < n) x[pc] = -tot;                       // acc = -A^-1
    gsync<64>();
  }
  if (pf) pf[2] += MJB_MEMTIME() - tq2;
}
#else
template <typename MRef>
MJB_DEV mjb_f16v mfma_sweep_invert32(MRef m, const float* M, const float* J, const float* dw, int nefc, int mode, int n, int lane, unsigned long long* pf = nullptr) {
  const int h = lane >> 5, c_ = lane & 31;
  unsigned long long tq0 = pf ? MJB_MEMTIME() : 0;
  mjb_f16v acc;
  {
    // acc[4q+t] of lane (h, c) = A[8q+4h+t][c] = A[c][8q+4h+t] (symmetric): every lane reads along its own row of M,
    // so the 16 LDS reads use one address register and immediate offsets.  Rows/columns >= n are padded with identity.
    const int c = c_;
    const bool cin = c < n;
    const float* Mc = M + (cin ? c : 0) * n + 4 * h;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int k = 8 * (i >> 2) + (i & 3);                    // this register holds row k + 4 h
      const float mv = Mc[k];                                  // may read past row c when the row is >= n: masked
      // Round 3: one SHARED lane mask (c < n) for the registers whose rows exist in both halves (k + 4 < n: a uniform test, a
      // compile-time one in the specialised kernel); only the last registers need the per-row test, and only they can hold the
      // identity padding.  (The round-2 form built two lane masks per register - 32 SGPR pairs the compiler hoisted out of the step
      // loop and kept spilled in VGPR lanes: two v_readlane + a hazard nop per use.)  Same values bit for bit.
      float v;
      if (k + 4 < n) v = cin ? mv : 0.0f;
      else v = (cin && k + 4 * h < n) ? mv : 0.0f;
      if (mode == 2) { if (k + 4 * h == c) v += cin ? m.timestep * m.dof_damping[c] : 1.0f; }      // + h D on the diagonal (identity on the padding)
      else if (k + 4 >= n) { if (k + 4 * h == c && !cin) v += 1.0f; }                              // identity padding of rows / columns >= n
      acc[i] = v;
    }
    if (mode == 1) {
      // Hessian M + J^T D J: one rank-2 MFMA per PAIR OF ACTIVE ROWS (D != 0), the next pair's J loads in flight
      // while the current MFMA runs.
      const int cm = c < n ? c : 0;
      for (int base = 0; base < nefc; base += 64) {
        const int rix = base + lane;
        const float dl = rix < nefc ? dw[rix] : 0.0f;
        unsigned long long act = MJB_BALLOT(dl != 0.0f);
        float jcur = 0.0f, dcur = 0.0f;
        bool have = false;
        while (act) {
          int ra = __builtin_ctzll(act); act &= act - 1;
          int rb = ra; float dB = 0.0f;
          if (act) { rb = __builtin_ctzll(act); act &= act - 1; dB = rdlane_f(dl, rb); }
          float dA = rdlane_f(dl, ra);
          int rr = h == 0 ? ra : rb;
          float jn = J[(base + rr) * n + cm];
          float dn = h == 0 ? dA : dB;
          if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
          jcur = c < n ? jn : 0.0f; dcur = dn; have = true;
        }
        if (have) acc = MJB_MFMA(dcur * jcur, jcur, acc);
      }
    }
  }
  unsigned long long tq1 = pf ? MJB_MEMTIME() : 0;
#if MJB_PANEL4
  // FOUR pivots per panel (experiment, see MJB_PANEL4 above).  Columns j0..j0+3 are rows 8q+4h+t of ONE half, registers 4q..4q+3.
  // Block elimination  B <- B - U P^-1 U^T  with P = L D L^T (unit lower L, in scalars):  W = U' L^-T  (per lane: three dependent
  // FMAs),  Z = W D^-1,  B -= Z W^T  as two rank-2 MFMAs;  U' = U - [e_j0 .. e_j3] makes the same products deliver the pivot rows /
  // columns and -P^-1 up to a -2 on the four pivot diagonals (as in the two-pivot form).
#pragma unroll
  for (int jq = 0; jq < 8; jq++) {
    const int j0 = 4 * jq, hj = jq & 1, ij = 4 * (jq >> 1);
    if (j0 >= n) continue;                                    // padded (identity) columns: nothing to eliminate (uniform skip)
    int c = c_, ln = lane;
    MJB_OPAQUE2(c, ln);                     // keep the per-column lane compares in the loop (cheaper than spilled masks)
    const float u0 = half_bcast(acc[ij], hj), u1 = half_bcast(acc[ij + 1], hj), u2 = half_bcast(acc[ij + 2], hj), u3 = half_bcast(acc[ij + 3], hj);
    const float tiny = Num<float>::minval();
    const float a00 = __builtin_fmaxf(rdlane_f(u0, j0), tiny), a10 = rdlane_f(u0, j0 + 1), a20 = rdlane_f(u0, j0 + 2), a30 = rdlane_f(u0, j0 + 3);
    const float a11 = rdlane_f(u1, j0 + 1), a21 = rdlane_f(u1, j0 + 2), a31 = rdlane_f(u1, j0 + 3);
    const float a22 = rdlane_f(u2, j0 + 2), a32 = rdlane_f(u2, j0 + 3), a33 = rdlane_f(u3, j0 + 3);
    const float i0 = MJB_RCPF(a00);
    const float l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const float d1 = __builtin_fmaxf(a11 - l10 * a10, a11 * tiny), i1 = MJB_RCPF(d1);
    const float t21 = a21 - l20 * a10, t31 = a31 - l30 * a10;
    const float l21 = t21 * i1, l31 = t31 * i1;
    const float d2 = __builtin_fmaxf(a22 - l20 * a20 - l21 * t21, a22 * tiny), i2 = MJB_RCPF(d2);
    const float t32 = a32 - l30 * a20 - l31 * t21;
    const float l32 = t32 * i2;
    const float d3 = __builtin_fmaxf(a33 - l30 * a30 - l31 * t31 - l32 * t32, a33 * tiny), i3 = MJB_RCPF(d3);
    const float w0 = c == j0 ? u0 - 1.0f : u0;
    const float w1 = (c == j0 + 1 ? u1 - 1.0f : u1) - l10 * w0;
    const float w2 = (c == j0 + 2 ? u2 - 1.0f : u2) - l20 * w0 - l21 * w1;
    const float w3 = (c == j0 + 3 ? u3 - 1.0f : u3) - l30 * w0 - l31 * w1 - l32 * w2;
#pragma unroll
    for (int t = 0; t < 4; t++) acc[ij + t] -= ln == 32 * hj + j0 + t ? 2.0f : 0.0f;
    acc = MJB_MFMA(h == 0 ? -(w0 * i0) : -(w1 * i1), h == 0 ? w0 : w1, acc);
    acc = MJB_MFMA(h == 0 ? -(w2 * i2) : -(w3 * i3), h == 0 ? w2 : w3, acc);
  }
#else
#pragma unroll
  for (int jb = 0; jb < 16; jb++) {
    const int j0 = 2 * jb, j1 = j0 + 1, hj = (j0 >> 2) & 1, ij = 4 * (j0 >> 3) + (j0 & 3);
    if (j0 >= n) continue;                                    // padded (identity) columns: nothing to eliminate (uniform skip)
    int c = c_, ln = lane;
    MJB_OPAQUE2(c, ln);                     // keep the per-column lane compares in the loop (cheaper than spilled masks)
    float u0 = half_bcast(acc[ij], hj), u1 = half_bcast(acc[ij + 1], hj);
    float a = __builtin_fmaxf(rdlane_f(u0, j0), Num<float>::minval()), b = rdlane_f(u0, j1), d = rdlane_f(u1, j1);
    float det = __builtin_fmaxf(a * d - b * b, a * Num<float>::minval());
    float rdet = MJB_RCPF(det);
    float p00 = d * rdet, p01 = -b * rdet, p11 = a * rdet;
    float q0 = h == 0 ? p00 : p01, q1 = h == 0 ? p01 : p11;
    float u0p = c == j0 ? u0 - 1.0f : u0, u1p = c == j1 ? u1 - 1.0f : u1;
    float aop = -(u0p * q0 + u1p * q1);
    float bop = h == 0 ? u0p : u1p;
    acc[ij] -= ln == 32 * hj + j0 ? 2.0f : 0.0f;
    acc[ij + 1] -= ln == 32 * hj + j1 ? 2.0f : 0.0f;
    acc = MJB_MFMA(aop, bop, acc);
  }
#endif
  if (pf) { unsigned long long tq2 = MJB_MEMTIME(); pf[0] += tq1 - tq0; pf[1] += tq2 - tq1; }
  return acc;
}
// x <- A^-1 x with acc = -A^-1 from mfma_sweep_invert32 (bpad: 32 words of LDS for the zero-padded right-hand side)
MJB_DEV void mfma_sweep_apply32(const mjb_f16v& acc, float* bpad, int n, int lane, float* x, unsigned long long* pf = nullptr) {
  const int h = lane >> 5, c_ = lane & 31;
  unsigned long long tq2 = pf ? MJB_MEMTIME() : 0;
  if (h == 0) bpad[c_] = c_ < n ? x[c_] : 0.0f;               // right-hand side, zero-padded to 32
  gsync<64>();
  {
    const float* bp = bpad + 4 * h;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) { s0 += acc[t] * bp[t]; s1 += acc[4 + t] * bp[8 + t]; s2 += acc[8 + t] * bp[16 + t]; s3 += acc[12 + t] * bp[24 + t]; }
    float s = (s0 + s1) + (s2 + s3);
    float tot = half_sum(s);
    if (h == 0 && c_ < n) x[c_] = -tot;                       // acc = -A^-1
    gsync<64>();
  }
  if (pf) pf[2] += MJB_MEMTIME() - tq2;
}
#endif
template <typename MRef>
MJB_DEV void mfma_sweep_solve32(MRef m, const float* M, float* bpad, const float* J, const float* dw, int nefc, int mode, int n, int lane, float* x, unsigned long long* pf = nullptr) {
  const mjb_f16v acc = mfma_sweep_invert32<MRef>(m, M, J, dw, nefc, mode, n, lane, pf);
  mfma_sweep_apply32(acc, bpad, n, lane, x, pf);
}
template <typename MRef>
MJB_DEV void mfma_sweep_solve32(MRef, const double*, double*, const double*, const double*, int, int, int, int, double*, unsigned long long* = nullptr) {}
template <typename MRef>
MJB_DEV void mfma_factor32(MRef, const double*, double*, double*, double*, const double*, const double*, int, int, int, int, double*, unsigned long long* = nullptr) {}
template <typename MRef>
MJB_DEV void mfma_factor32_r2(MRef, const double*, double*, double*, const double*, const double*, int, int, int, int, double*, unsigned long long* = nullptr) {}
MJB_DEV void mfma_solve32(const double*, const double*, double*, int, int) {}

// W <- Cholesky factor of M (mode 0), M + J^T D_active J (mode 1, dw in efc_jv) or M + h diag(damping) (mode 2)
// If x != nullptr the system (factor) x = x is solved in the same pass (fused on the register path).
template <typename T, int G> MJB_DEV void factor_W_impl(Ctx<T>& c, int mode, T* x);
// true where factor_W() inverts-and-solves in registers (fp32, one wave per environment, nv <= 32) and leaves NO factor in W
template <typename T, int G> MJB_DEV bool fused_inverse_path(int nv) { return G == 64 && sizeof(T) == 4 && nv <= 32; }
template <typename T, int G> MJB_DEV void factor_W(Ctx<T>& c, int mode, T* x) {
#if defined(MJB_PROFILE) && !defined(MJB_HOST_EMU)
  unsigned long long t0_ = __builtin_amdgcn_s_memtime();
  factor_W_impl<T, G>(c, mode, x);
  c.pacc[PH_FAC_ALL] += __builtin_amdgcn_s_memtime() - t0_;   // informational: already inside the enclosing phase's stamp
#else
  factor_W_impl<T, G>(c, mode, x);
#endif
}
template <typename T, int G> MJB_DEV void factor_W_impl(Ctx<T>& c, int mode, T* x) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, nefc = c.nefc;
  T *M = w + L.M, *W = w + L.W, *J = w + L.efc_J, *dw = w + L.efc_jv;
  if (G == 64) {
    if (nv <= 32) {
      // M and M + hD are well conditioned: in-register sweep inverse.  The Hessian (contact stiffness on a few dofs) keeps
      // the backward-stable Cholesky, whose packed factor in W is reused while the active set does not change.
#if defined(MJB_PROFILE)
      unsigned long long* pf = c.pacc + PH_FAC_LOAD;
#else
      unsigned long long* pf = nullptr;
#endif
      if (sizeof(T) == 4 && mode != MJB_SWEEP_EXCLUDE) mfma_sweep_solve32<ModelRef<T>>(m, M, w + L.tmp, J, dw, nefc, mode, nv, lane, x, pf);
#ifdef MJB_R2_FACTOR
      else if (sizeof(T) == 4) mfma_factor32_r2<ModelRef<T>>(m, M, W, w + L.tmp, J, dw, nefc, mode, nv, lane, x, pf);
#else
      else if (sizeof(T) == 4) mfma_factor32<ModelRef<T>>(m, M, W, w + L.tmp, w + L.cholcol, J, dw, nefc, mode, nv, lane, x, pf);
#endif
      else reg_factor32<T, ModelRef<T>>(m, M, W, w + L.tmp, J, dw, nefc, mode, nv, lane, x);
      return;
    }
    tile_factor<T, 8, ModelRef<T>>(m, M, W, w + L.tmp, w + L.cholcol, J, dw, nefc, mode, nv, lane);
    if (x) chol_solve<T, G>(W, w + L.tmp, x, nv, lane);
    return;
  }
  int np = nv * (nv + 1) / 2;
  for (int idx = lane; idx < np; idx += G) {
    int i, k;
    tri_rc(m.tri_tab, idx, i, k);
    T h = M[i * nv + k];
    if (mode == 1) {
      for (int r = 0; r < nefc; r++) {
        T d = dw[r];
        if (d != 0) h += d * J[r * nv + i] * J[r * nv + k];
      }
    } else if (mode == 2 && i == k) h += m.timestep * m.dof_damping[i];
    W[tri_at(i, k)] = h;
  }
  gsync<G>();
  chol_factor<T, G>(W, w + L.tmp, nv, lane, m.tri_tab);
  if (x) chol_solve<T, G>(W, w + L.tmp, x, nv, lane);
}

// ---------------------------------------------------------------------------
// A1 kinematics: tree levels in order, bodies of a level across lanes
// ---------------------------------------------------------------------------
// arr[b] += arr[parent] down the tree (velocities, bias accelerations): depth and parent of the lane's body in registers
template <typename T, int G, int NC> MJB_DEV void tree_forward_sum_levels(Ctx<T>& c, T* arr) {
  MJB_ENV(c); const int lane = c.lane;
  const int b0 = 1 + lane;
  int dep0 = -1, par0 = 0;
  if (b0 < m.nbody) { dep0 = m.body_depth[b0]; par0 = m.body_parentid[b0]; }
  for (int lev = 2; lev <= m.nlevel; lev++) {                   // depth-1 bodies hang off the world body: nothing to add / added by the caller
    for (int b = b0; b < m.nbody; b += G) {
      const int dep = b == b0 ? dep0 : m.body_depth[b];
      if (dep != lev) continue;
      const int p = b == b0 ? par0 : m.body_parentid[b];
      T pa[NC], ch[NC];
#pragma unroll
      for (int q = 0; q < NC; q++) { pa[q] = arr[NC * p + q]; ch[q] = arr[NC * b + q]; }
#pragma unroll
      for (int q = 0; q < NC; q++) arr[NC * b + q] = ch[q] + pa[q];
    }
    gsync<G>();
  }
}

// Subtree sums (subtree COM, composite inertias, subtree forces): every body adds its NC-vector into its parent.
// Bodies are scheduled in "rounds" (host: deepest level first, one sibling rank per round) so that no two bodies of a
// round share a parent and every body is complete before it is added.  The lane's body, round and parent stay in
// registers: one round = LDS read-add-write, no index tables on the critical path.
template <typename T, int G, int NC> MJB_DEV void tree_backward_sum_rounds(Ctx<T>& c, T* arr, int nrounds) {
  MJB_ENV(c); const int lane = c.lane;
  const int b0 = 1 + lane;
  int rd0 = -1, p0 = 0;
  if (b0 < m.nbody) { rd0 = m.body_round[b0]; p0 = m.body_parentid[b0]; }
  for (int rd = 0; rd < nrounds; rd++) {
    if (rd0 == rd) {                                            // all loads first: the stores may alias them for the compiler
      T pa[NC], ch[NC];
#pragma unroll
      for (int k = 0; k < NC; k++) { pa[k] = arr[NC * p0 + k]; ch[k] = arr[NC * b0 + k]; }
#pragma unroll
      for (int k = 0; k < NC; k++) arr[NC * p0 + k] = pa[k] + ch[k];
    }
    for (int b = b0 + G; b < m.nbody; b += G) {                 // models with more bodies than lanes
      if (m.body_round[b] != rd) continue;
      int p = m.body_parentid[b];
      T pa[NC], ch[NC];
#pragma unroll
      for (int k = 0; k < NC; k++) { pa[k] = arr[NC * p + k]; ch[k] = arr[NC * b + k]; }
#pragma unroll
      for (int k = 0; k < NC; k++) arr[NC * p + k] = pa[k] + ch[k];
    }
    gsync<G>();
  }
}

// Flat forms of the two tree sums: ONE read phase, a sync, one write phase - instead of a dependent LDS round trip per tree
// level / per round.  Work item = (body, component), all 64 lanes busy.
//  * ancestor sum  x[b] += sum of x over the proper ancestors of b (the world body excluded): host table body_anc
//    [nbody, nlevel] (parent, grandparent, ...), all table entries and all LDS reads of an item are independent loads;
//  * subtree sum   x[b] += sum of x over the descendants of b: with the bodies in depth-first order (checked on the host:
//    dfs_ok) the descendants are the id range (b, b + nsub[b]], read with loads masked beyond nsub.
// Every item reads ORIGINAL values (the sync separates all reads from all writes), so the sums are formed in place.
// Up to MJB_TREE_PASSES * G items stay in registers; larger models / non-DFS body orders use the level / round loops.
#define MJB_TREE_PASSES 4
template <typename T, int G, int NC> MJB_DEV void tree_forward_sum(Ctx<T>& c, T* arr) {
  MJB_ENV(c); const int lane = c.lane;
  const int nitem = m.nbody * NC, nl = m.nlevel;
  if (nitem > MJB_TREE_PASSES * G) { tree_forward_sum_levels<T, G, NC>(c, arr); return; }
  T s[MJB_TREE_PASSES];
#pragma unroll
  for (int ps = 0; ps < MJB_TREE_PASSES; ps++) {
    const int e = lane + ps * G;
    s[ps] = 0;
    if (e < nitem) {
      const int b = e / NC, k = e - b * NC;
      int na = b > 0 ? m.body_depth[b] - 1 : 0;
      MJB_OPAQUE1(na);                            // lane constants: compared here (one v_cmp each), not hoisted out of the step loop as spilled lane masks
      T acc = arr[e];
      for (int u = 0; u < nl - 1; u++) {
        const int a = m.body_anc[b * nl + u];
        T v = arr[NC * a + k];
        acc += u < na ? v : (T)0;
      }
      s[ps] = acc;
    }
  }
  gsync<G>();
#pragma unroll
  for (int ps = 0; ps < MJB_TREE_PASSES; ps++) { const int e = lane + ps * G; if (e < nitem) arr[e] = s[ps]; }
  gsync<G>();
}
template <typename T, int G, int NC> MJB_DEV void tree_backward_sum(Ctx<T>& c, T* arr, int nrounds, int first_body) {
  MJB_ENV(c); const int lane = c.lane;
  const int nitem = m.nbody * NC, mx = m.max_nsub;
  if (!m.dfs_ok || nitem > MJB_TREE_PASSES * G) { tree_backward_sum_rounds<T, G, NC>(c, arr, nrounds); return; }
  T s[MJB_TREE_PASSES];
#pragma unroll
  for (int ps = 0; ps < MJB_TREE_PASSES; ps++) {
    if (ps * G >= nitem) { s[ps] = 0; continue; }             // (uniform) no item in this pass
    const int e = lane + ps * G;
    const bool on = e < nitem;
    const int b = on ? e / NC : 0;
    const int n = on && b >= first_body ? m.body_nsub[b] : 0;
    const T* base = arr + (on ? e : 0);
    T acc = base[0];
    // four masked loads in flight per step - for as long as ANY item of this pass has descendants left: the bodies are in depth-first
    // order, so the long subtrees (torso, waist) sit in the first pass and the later passes stop after one step (humanoid, 10
    // components: 4 + 1 + 1 steps instead of 3 x 4)
    for (int d0 = 0; d0 < mx && gany<G>(d0 < n); d0 += 4) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int dd = d0 + u + 1; v[u] = base[NC * (dd <= n ? dd : n)]; }
#pragma unroll
      for (int u = 0; u < 4; u++) acc += d0 + u + 1 <= n ? v[u] : (T)0;
    }
    s[ps] = acc;
  }
  gsync<G>();
#pragma unroll
  for (int ps = 0; ps < MJB_TREE_PASSES; ps++) { const int e = lane + ps * G; if (e < nitem) arr[e] = s[ps]; }
  gsync<G>();
}

// rotate v by the unit quaternion q (through the rotation matrix, like the per-body frames)
template <typename T> MJB_DEV void quat_rot(T* r, const T* q, const T* v) {
  T R[9];
  quat2mat(R, q);
  mulmatvec3(r, R, v);
}

// Three stages instead of one long per-level body loop (only a few lanes are active per tree level):
//   (1) all bodies in parallel: pose of the body RELATIVE TO ITS PARENT (joint chain applied in the parent frame), and
//       the joints' anchors / axes in the parent frame;
//   (2) per tree level: compose with the parent's world pose (one quaternion product + one rotation per body);
//   (3) all bodies / joints in parallel: xmat, inertial frames, world anchors and axes.
template <typename T, int G> MJB_DEV void kinematics(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane;
  T *xpos = w + L.xpos, *xquat = w + L.xquat, *xmat = w + L.xmat, *xipos = w + L.xipos, *ximat = w + L.ximat;
  T *qpos = w + L.qpos, *xanchor = w + L.xanchor, *xaxis = w + L.xaxis;
  if (lane == 0) {
    xpos[0] = xpos[1] = xpos[2] = 0; xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0;
    xipos[0] = xipos[1] = xipos[2] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) { xmat[k] = (k % 4 == 0) ? (T)1 : (T)0; ximat[k] = xmat[k]; }
  }
  // (1) local poses; xpos/xquat/xanchor/xaxis temporarily hold parent-frame values
  for (int b = 1 + lane; b < m.nbody; b += G) {
    const int jadr = m.body_irec[4 * b], jnum = m.body_irec[4 * b + 1], jt0 = m.body_irec[4 * b + 2];
    T pos[3], quat[4], R[9];
    if (jnum == 1 && jt0 == JNT_FREE) {
      int qa = m.jnt_irec[6 * jadr + 1];
      quat[0] = qpos[qa + 3]; quat[1] = qpos[qa + 4]; quat[2] = qpos[qa + 5]; quat[3] = qpos[qa + 6];
      quat_normalize(quat);
      qpos[qa + 3] = quat[0]; qpos[qa + 4] = quat[1]; qpos[qa + 5] = quat[2]; qpos[qa + 6] = quat[3];
      pos[0] = qpos[qa]; pos[1] = qpos[qa + 1]; pos[2] = qpos[qa + 2];
      xanchor[3 * jadr] = pos[0]; xanchor[3 * jadr + 1] = pos[1]; xanchor[3 * jadr + 2] = pos[2];
      xaxis[3 * jadr] = 0; xaxis[3 * jadr + 1] = 0; xaxis[3 * jadr + 2] = 1;
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) pos[k] = m.body_pos[3 * b + k];
#pragma unroll
      for (int k = 0; k < 4; k++) quat[k] = m.body_quat[4 * b + k];
      quat2mat(R, quat);                                       // R follows quat through the joints: formed once per change, not twice
      for (int j = jadr; j < jadr + jnum; j++) {
        T jp[3] = {m.jnt_rec[8 * j], m.jnt_rec[8 * j + 1], m.jnt_rec[8 * j + 2]};
        T ja[3] = {m.jnt_rec[8 * j + 3], m.jnt_rec[8 * j + 4], m.jnt_rec[8 * j + 5]};
        const int jtype = m.jnt_irec[6 * j], qa = m.jnt_irec[6 * j + 1];
        T val = qpos[qa] - m.jnt_rec[8 * j + 6];
        T anchor[3], axis[3];
        mulmatvec3(anchor, R, jp);
        anchor[0] += pos[0]; anchor[1] += pos[1]; anchor[2] += pos[2];
        mulmatvec3(axis, R, ja);
        xanchor[3 * j] = anchor[0]; xanchor[3 * j + 1] = anchor[1]; xanchor[3 * j + 2] = anchor[2];
        xaxis[3 * j] = axis[0]; xaxis[3 * j + 1] = axis[1]; xaxis[3 * j + 2] = axis[2];
        if (jtype == JNT_SLIDE) {
          pos[0] += axis[0] * val; pos[1] += axis[1] * val; pos[2] += axis[2] * val;
        } else {
          T ql[4], qn[4], v[3];
          axisangle2quat(ql, ja, val);
          quat_mul(qn, quat, ql);
          quat[0] = qn[0]; quat[1] = qn[1]; quat[2] = qn[2]; quat[3] = qn[3];
          quat2mat(R, quat);                                   // = quat_rot(v, quat, jp), keeping the matrix for the next joint
          mulmatvec3(v, R, jp);
          pos[0] = anchor[0] - v[0]; pos[1] = anchor[1] - v[1]; pos[2] = anchor[2] - v[2];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) xpos[3 * b + k] = pos[k];
#pragma unroll
    for (int k = 0; k < 4; k++) xquat[4 * b + k] = quat[k];
  }
  gsync<G>();
  // (2) compose down the tree, in place.
  // Round 3: POINTER JUMPING instead of one pass per tree level.  The per-level loop ran nlevel (humanoid: 7) dependent rounds with
  // one to four lanes active each (mean 7.3 active lanes over the whole phase, profiles/r03_phase_table.txt: the worst of the
  // kernel) - ~105 instructions per level whatever the number of bodies in it.  Rigid transforms compose associatively, so every
  // body can instead double the distance to the ancestor its pose is relative to: round r composes the pose of body b (relative
  // to its ancestor at distance 2^r) with the pose of THAT ancestor (itself relative to its own ancestor at distance 2^r); after
  // ceil(log2(nlevel)) rounds (3) every pose is relative to the world.  All bodies work in every round (16 lanes), every round
  // reads all its inputs, syncs, then writes.  The unit quaternion is re-normalised ONCE per body at the end (the level loop did
  // it per level, like mj_kinematics; the product of unit quaternions stays unit to rounding) - results differ from the
  // sequential composition in the last bits only (float64 instantiation vs the oracle: <= 1e-15).
#ifdef MJB_R2_KINEMATICS                                            // A/B timing against the round-2 form (scripts/gpu_perf_quick.py)
  if (false) {
#else
  if (m.nbody - 1 <= G) {
#endif
    const int b = 1 + lane, nl = m.nlevel;
    const bool own = b < m.nbody;
    const int dep = own ? m.body_depth[b] : 0;
    for (int d = 1; d < nl; d <<= 1) {
      const bool hop = own && d <= dep - 1;                        // an ancestor at distance d exists below the world body
      T pq[4], pp[3], lq[4], lp[3], q[4], t[3];
      if (hop) {
        const int a = m.body_anc[b * nl + d - 1];
#pragma unroll
        for (int k = 0; k < 4; k++) { pq[k] = xquat[4 * a + k]; lq[k] = xquat[4 * b + k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { pp[k] = xpos[3 * a + k]; lp[k] = xpos[3 * b + k]; }
        quat_mul(q, pq, lq);
        quat_rot(t, pq, lp);
      }
      gsync<G>();                                                   // every read of this round before any write
      if (hop) {
#pragma unroll
        for (int k = 0; k < 4; k++) xquat[4 * b + k] = q[k];
#pragma unroll
        for (int k = 0; k < 3; k++) xpos[3 * b + k] = pp[k] + t[k];
      }
      gsync<G>();
    }
    if (own) {
      T q[4];
#pragma unroll
      for (int k = 0; k < 4; k++) q[k] = xquat[4 * b + k];
      quat_normalize(q);
#pragma unroll
      for (int k = 0; k < 4; k++) xquat[4 * b + k] = q[k];
    }
    gsync<G>();
  } else {                                                          // more bodies than lanes: one pass per tree level
    const int b0 = 1 + lane;
    int dep0 = -1, par0 = 0;                                    // the lane's body: depth and parent stay in registers
    if (b0 < m.nbody) { dep0 = m.body_depth[b0]; par0 = m.body_parentid[b0]; }
    for (int lev = 1; lev <= m.nlevel; lev++) {
      for (int b = b0; b < m.nbody; b += G) {
        const int dep = b == b0 ? dep0 : m.body_depth[b];
        if (dep != lev) continue;
        const int p = b == b0 ? par0 : m.body_parentid[b];
        T pq[4], pp[3], lq[4], lp[3], q[4], t[3];
#pragma unroll
        for (int k = 0; k < 4; k++) { pq[k] = xquat[4 * p + k]; lq[k] = xquat[4 * b + k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { pp[k] = xpos[3 * p + k]; lp[k] = xpos[3 * b + k]; }
        quat_mul(q, pq, lq);
        quat_normalize(q);
        quat_rot(t, pq, lp);
#pragma unroll
        for (int k = 0; k < 4; k++) xquat[4 * b + k] = q[k];
#pragma unroll
        for (int k = 0; k < 3; k++) xpos[3 * b + k] = pp[k] + t[k];
      }
      gsync<G>();
    }
  }
  // (3) frames of the bodies and world anchors / axes of the joints (read the PARENT pose: already final)
  for (int b = 1 + lane; b < m.nbody; b += G) {
    T quat[4], R[9], t[3], q2[4], R2[9];
#pragma unroll
    for (int k = 0; k < 4; k++) quat[k] = xquat[4 * b + k];
    quat2mat(R, quat);
    T ip[3] = {m.body_ipos[3 * b], m.body_ipos[3 * b + 1], m.body_ipos[3 * b + 2]};
    T iq[4] = {m.body_iquat[4 * b], m.body_iquat[4 * b + 1], m.body_iquat[4 * b + 2], m.body_iquat[4 * b + 3]};
    mulmatvec3(t, R, ip);
    quat_mul(q2, quat, iq);
    quat2mat(R2, q2);
#pragma unroll
    for (int k = 0; k < 3; k++) xipos[3 * b + k] = xpos[3 * b + k] + t[k];
#pragma unroll
    for (int k = 0; k < 9; k++) { xmat[9 * b + k] = R[k]; ximat[9 * b + k] = R2[k]; }
  }
  for (int j = lane; j < m.njnt; j += G) {
    if (m.jnt_irec[6 * j] == JNT_FREE) continue;                // already world values (the parent is the world body)
    int p = m.jnt_irec[6 * j + 4];
    T pq[4], R[9], a[3], ax[3], la[3] = {xanchor[3 * j], xanchor[3 * j + 1], xanchor[3 * j + 2]}, lx[3] = {xaxis[3 * j], xaxis[3 * j + 1], xaxis[3 * j + 2]};
#pragma unroll
    for (int k = 0; k < 4; k++) pq[k] = xquat[4 * p + k];
    quat2mat(R, pq);
    mulmatvec3(a, R, la);
    mulmatvec3(ax, R, lx);
#pragma unroll
    for (int k = 0; k < 3; k++) { xanchor[3 * j + k] = xpos[3 * p + k] + a[k]; xaxis[3 * j + k] = ax[k]; }
  }
  gsync<G>();
  T *gx = w + L.geom_xpos, *gm = w + L.geom_xmat, *sx = w + L.site_xpos, *sm = w + L.site_xmat;
  for (int g = lane; g < m.ngeom + m.nsite; g += G) {
    bool is_geom = g < m.ngeom;
    int id = is_geom ? g : g - m.ngeom;
    int b = is_geom ? m.geom_bodyid[id] : m.site_bodyid[id];
    auto lp = is_geom ? m.geom_pos + 3 * id : m.site_pos + 3 * id;
    auto lq = is_geom ? m.geom_quat + 4 * id : m.site_quat + 4 * id;
    T p3[3] = {lp[0], lp[1], lp[2]}, q4[4] = {lq[0], lq[1], lq[2], lq[3]}, bm[9], bq[4], t[3], q[4], R[9];
#pragma unroll
    for (int k = 0; k < 9; k++) bm[k] = xmat[9 * b + k];
#pragma unroll
    for (int k = 0; k < 4; k++) bq[k] = xquat[4 * b + k];
    mulmatvec3(t, bm, p3);
    quat_mul(q, bq, q4);
    quat_normalize(q);
    quat2mat(R, q);
    T* op = is_geom ? gx + 3 * id : sx + 3 * id;
    T* om = is_geom ? gm + 9 * id : sm + 9 * id;
#pragma unroll
    for (int k = 0; k < 3; k++) op[k] = xpos[3 * b + k] + t[k];
#pragma unroll
    for (int k = 0; k < 9; k++) om[k] = R[k];
  }
  gsync<G>();
}

// ---------------------------------------------------------------------------
// A2 com-frame quantities: subtree_com, cinert, cdof
// ---------------------------------------------------------------------------
template <typename T, int G> MJB_DEV void com_pos(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane;
  T *sc = w + L.subtree_com, *xipos = w + L.xipos, *ximat = w + L.ximat, *xmat = w + L.xmat;
  for (int b = lane; b < m.nbody; b += G) {
    T ms = m.body_mass[b];
#pragma unroll
    for (int k = 0; k < 3; k++) sc[3 * b + k] = ms * xipos[3 * b + k];
  }
  gsync<G>();
  tree_backward_sum<T, G, 3>(c, sc, m.nround, 0);           // includes the world body's subtree (everything)
  for (int b = lane; b < m.nbody; b += G) {
    T sm = m.body_subtreemass[b];
    if (sm < Num<T>::minval()) { sc[3 * b] = xipos[3 * b]; sc[3 * b + 1] = xipos[3 * b + 1]; sc[3 * b + 2] = xipos[3 * b + 2]; }
    else { T inv = 1 / sm; sc[3 * b] *= inv; sc[3 * b + 1] *= inv; sc[3 * b + 2] *= inv; }
  }
  gsync<G>();
  T* cin = w + L.cinert;
  for (int b = lane; b < m.nbody; b += G) {
    if (b == 0) {
#pragma unroll
      for (int k = 0; k < 10; k++) cin[k] = 0;
      continue;
    }
    int r = m.body_rootid[b];
    T off[3], im[9], res[10];
    T inr[3] = {m.body_inertia[3 * b], m.body_inertia[3 * b + 1], m.body_inertia[3 * b + 2]};
#pragma unroll
    for (int k = 0; k < 3; k++) off[k] = xipos[3 * b + k] - sc[3 * r + k];
#pragma unroll
    for (int k = 0; k < 9; k++) im[k] = ximat[9 * b + k];
    inert_com(res, inr, im, off, m.body_mass[b]);
#pragma unroll
    for (int k = 0; k < 10; k++) cin[10 * b + k] = res[k];
  }
  T *cdof = w + L.cdof, *xanchor = w + L.xanchor, *xaxis = w + L.xaxis;
  for (int j = lane; j < m.njnt; j += G) {
    int b = m.jnt_irec[6 * j + 3], da = m.jnt_irec[6 * j + 2], r = m.jnt_irec[6 * j + 5], jt = m.jnt_irec[6 * j];
    T off[3];
#pragma unroll
    for (int k = 0; k < 3; k++) off[k] = sc[3 * r + k] - xanchor[3 * j + k];
    T* cd = cdof + 6 * da;
    if (jt == JNT_FREE) {
      for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int k = 0; k < 6; k++) cd[6 * i + k] = (k == 3 + i) ? (T)1 : (T)0;
        T ax[3] = {xmat[9 * b + i], xmat[9 * b + 3 + i], xmat[9 * b + 6 + i]}, cr[3];
        cross3(cr, ax, off);
        T* cc = cd + 6 * (3 + i);
        cc[0] = ax[0]; cc[1] = ax[1]; cc[2] = ax[2]; cc[3] = cr[0]; cc[4] = cr[1]; cc[5] = cr[2];
      }
    } else if (jt == JNT_SLIDE) {
      cd[0] = cd[1] = cd[2] = 0; cd[3] = xaxis[3 * j]; cd[4] = xaxis[3 * j + 1]; cd[5] = xaxis[3 * j + 2];
    } else {
      T ax[3] = {xaxis[3 * j], xaxis[3 * j + 1], xaxis[3 * j + 2]}, cr[3];
      cross3(cr, ax, off);
      cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2]; cd[3] = cr[0]; cd[4] = cr[1]; cd[5] = cr[2];
    }
  }
  // fixed tendons: length and Jacobian
  T *tl = w + L.ten_length, *tj = w + L.ten_J, *qpos = w + L.qpos;
  for (int t = lane; t < m.ntendon; t += G) {
    for (int i = 0; i < m.nv; i++) tj[t * m.nv + i] = 0;
    T len = 0;
    for (int wi = m.tendon_adr[t]; wi < m.tendon_adr[t] + m.tendon_num[t]; wi++) {
      int j = m.wrap_objid[wi];
      len += m.wrap_prm[wi] * qpos[m.jnt_qposadr[j]];
      tj[t * m.nv + m.jnt_dofadr[j]] = m.wrap_prm[wi];
    }
    tl[t] = len;
  }
  gsync<G>();
}

// Jacobian column of dof i for a world point attached to body b (zero if i does not move b)
template <typename T> MJB_DEV void jac_col(const Ctx<T>& c, int b, int i, const T* point, T* jp, T* jr) {
  ModelRef<T> m = MJB_MODEL_OF(c.mp);
  jp[0] = jp[1] = jp[2] = 0;
  if (jr) { jr[0] = jr[1] = jr[2] = 0; }
  if (!((m.body_dofmask[b] >> i) & 1ull)) return;
  const T* cd = c.w + c.lp->cdof + 6 * i;
  const T* sc = c.w + c.lp->subtree_com + 3 * m.body_rootid[b];
  T off[3] = {point[0] - sc[0], point[1] - sc[1], point[2] - sc[2]}, ang[3] = {cd[0], cd[1], cd[2]}, t[3];
  cross3(t, ang, off);
  jp[0] = cd[3] + t[0]; jp[1] = cd[4] + t[1]; jp[2] = cd[5] + t[2];
  if (jr) { jr[0] = ang[0]; jr[1] = ang[1]; jr[2] = ang[2]; }
}

// ---------------------------------------------------------------------------
// A4 composite rigid body -> dense M in LDS, Cholesky factor in W
// ---------------------------------------------------------------------------
template <typename T, int G> MJB_DEV void crb_factor(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv;
  T *crb = w + L.crb, *cin = w + L.cinert, *cdof = w + L.cdof, *buf = w + L.dofbuf, *M = w + L.M, *W = w + L.W;
  for (int i = lane; i < 10 * m.nbody; i += G) crb[i] = cin[i];
  gsync<G>();
  tree_backward_sum<T, G, 10>(c, crb, m.nround_inner, 1);
  for (int i = lane; i < nv; i += G) {
    T in[10], v[6], r[6];
    int b = m.dof_bodyid[i];
#pragma unroll
    for (int k = 0; k < 10; k++) in[k] = crb[10 * b + k];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = cdof[6 * i + k];
    mul_inert_vec(r, in, v);
#pragma unroll
    for (int k = 0; k < 6; k++) buf[6 * i + k] = r[k];
  }
  gsync<G>();
  // entries of M outside the (dof, ancestor-dof) pairs are structurally zero and never written: cleared once per launch (env_run)
  for (int idx = lane; idx < m.nmpair; idx += G) {
    int pr = m.mpair[idx], i = pr >> 8, j = pr & 0xff;      // j is i or one of its ancestor dofs
    T val = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) val += cdof[6 * j + k] * buf[6 * i + k];
    if (i == j) val += m.dof_armature[i];
    M[i * nv + j] = val; M[j * nv + i] = val;
  }
  gsync<G>();     // M stays unfactored here: it is factored together with the M^-1 solve in actuation_acceleration
}

// ---------------------------------------------------------------------------
// A5 collision: static pair list, narrow phase per lane, ordered compaction
// ---------------------------------------------------------------------------
template <typename T> struct RawCon { T dist, pos[3], n[3], yh[3]; };

template <typename T> MJB_DEV int nc_sphere_sphere(const T* p1, T r1, const T* p2, T r2, T margin, RawCon<T>& o) {
  T dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  T cd = t_sqrt(dot3(dif, dif)), dist = cd - r1 - r2;
  if (dist > margin) return 0;
  if (cd < Num<T>::minval()) { o.n[0] = 1; o.n[1] = 0; o.n[2] = 0; } else { T s = 1 / cd; o.n[0] = dif[0] * s; o.n[1] = dif[1] * s; o.n[2] = dif[2] * s; }
  o.dist = dist; o.yh[0] = o.yh[1] = o.yh[2] = 0;
  T k = r1 + (T)0.5 * dist;
  o.pos[0] = p1[0] + o.n[0] * k; o.pos[1] = p1[1] + o.n[1] * k; o.pos[2] = p1[2] + o.n[2] * k;
  return 1;
}
template <typename T> MJB_DEV int nc_plane_sphere(const T* pp, const T* n, const T* sp, T r, T margin, RawCon<T>& o) {
  T dif[3] = {sp[0] - pp[0], sp[1] - pp[1], sp[2] - pp[2]};
  T dist = dot3(dif, n) - r;
  if (dist > margin) return 0;
  o.dist = dist; o.n[0] = n[0]; o.n[1] = n[1]; o.n[2] = n[2]; o.yh[0] = o.yh[1] = o.yh[2] = 0;
  T k = -(r + (T)0.5 * dist);
  o.pos[0] = sp[0] + n[0] * k; o.pos[1] = sp[1] + n[1] * k; o.pos[2] = sp[2] + n[2] * k;
  return 1;
}

// Narrow phase + ordered emission for ONE candidate pair per lane (p valid where `valid`); contacts are appended after
// `ncon` in lane order, i.e. in static pair order as long as the callers feed the pairs in that order.
template <typename T, int G> MJB_DEV void collide_pass(Ctx<T>& c, int p, bool valid, int& ncon, int& dropped) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane;
  T *gx = w + L.geom_xpos, *gm = w + L.geom_xmat, *con = w + L.con;
  int* con_pair = c.wi + L.i_con_pair;
  {
    int n = 0;
    RawCon<T> rc[4];
    if (valid) {
      int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p], t1 = m.geom_type[g1], t2 = m.geom_type[g2];
      T margin = m.pair_margin[p];
      T p1[3] = {gx[3 * g1], gx[3 * g1 + 1], gx[3 * g1 + 2]}, p2[3] = {gx[3 * g2], gx[3 * g2 + 1], gx[3 * g2 + 2]};
      T s1[3] = {m.geom_size[3 * g1], m.geom_size[3 * g1 + 1], m.geom_size[3 * g1 + 2]};
      T s2[3] = {m.geom_size[3 * g2], m.geom_size[3 * g2 + 1], m.geom_size[3 * g2 + 2]};
      T z1[3] = {gm[9 * g1 + 2], gm[9 * g1 + 5], gm[9 * g1 + 8]}, z2[3] = {gm[9 * g2 + 2], gm[9 * g2 + 5], gm[9 * g2 + 8]};
      if (t1 == G_PLANE) {
        if (t2 == G_SPHERE) n = nc_plane_sphere(p1, z1, p2, s2[0], margin, rc[0]);
        else if (t2 == G_CAPSULE) {
#pragma unroll
          for (int s = 0; s < 2; s++) {
            T sg = s == 0 ? s2[1] : -s2[1];
            T e[3] = {p2[0] + z2[0] * sg, p2[1] + z2[1] * sg, p2[2] + z2[2] * sg};
            RawCon<T> tmp;
            if (nc_plane_sphere(p1, z1, e, s2[0], margin, tmp)) {
              tmp.yh[0] = z2[0]; tmp.yh[1] = z2[1]; tmp.yh[2] = z2[2];
              if (n == 0) rc[0] = tmp; else rc[1] = tmp;
              n++;
            }
          }
        } else if (t2 == G_BOX) {
          T bm[9];
#pragma unroll
          for (int k = 0; k < 9; k++) bm[k] = gm[9 * g2 + k];
          T dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
          T dist = dot3(dif, z1);
#pragma unroll
          for (int i = 0; i < 8; i++) {
            T v[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])}, corner[3];
            mulmatvec3(corner, bm, v);
            T ld = dot3(z1, corner);
            if (n < 4 && !(dist + ld > margin || ld > 0)) {
              RawCon<T> tmp;
              tmp.dist = dist + ld; tmp.n[0] = z1[0]; tmp.n[1] = z1[1]; tmp.n[2] = z1[2]; tmp.yh[0] = tmp.yh[1] = tmp.yh[2] = 0;
              T k = -(T)0.5 * tmp.dist;
#pragma unroll
              for (int a = 0; a < 3; a++) tmp.pos[a] = corner[a] + p2[a] + z1[a] * k;
              if (n == 0) rc[0] = tmp; else if (n == 1) rc[1] = tmp; else if (n == 2) rc[2] = tmp; else rc[3] = tmp;
              n++;
            }
          }
        } else if (t2 == G_ELLIPSOID) {
          T em[9], nn[3] = {-z1[0], -z1[1], -z1[2]}, dl[3], s[3], wpt[3];
#pragma unroll
          for (int k = 0; k < 9; k++) em[k] = gm[9 * g2 + k];
          mulmatTvec3(dl, em, nn);
          T den = t_sqrt(s2[0] * s2[0] * dl[0] * dl[0] + s2[1] * s2[1] * dl[1] * dl[1] + s2[2] * s2[2] * dl[2] * dl[2]);
          den = t_max(den, Num<T>::minval());
#pragma unroll
          for (int k = 0; k < 3; k++) s[k] = s2[k] * s2[k] * dl[k] / den;
          mulmatvec3(wpt, em, s);
          wpt[0] += p2[0]; wpt[1] += p2[1]; wpt[2] += p2[2];
          T dif[3] = {wpt[0] - p1[0], wpt[1] - p1[1], wpt[2] - p1[2]};
          T dist = dot3(dif, z1);
          if (!(dist > margin)) {
            rc[0].dist = dist; rc[0].n[0] = z1[0]; rc[0].n[1] = z1[1]; rc[0].n[2] = z1[2]; rc[0].yh[0] = rc[0].yh[1] = rc[0].yh[2] = 0;
            T k = -(T)0.5 * dist;
#pragma unroll
            for (int a = 0; a < 3; a++) rc[0].pos[a] = wpt[a] + z1[a] * k;
            n = 1;
          }
        }
      } else if (t1 == G_SPHERE && t2 == G_SPHERE) {
        n = nc_sphere_sphere(p1, s1[0], p2, s2[0], margin, rc[0]);
      } else if (t1 == G_SPHERE && t2 == G_CAPSULE) {
        T dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        T x = dot3(z2, dif);
        x = t_min(t_max(x, -s2[1]), s2[1]);
        T pt[3] = {p2[0] + z2[0] * x, p2[1] + z2[1] * x, p2[2] + z2[2] * x};
        n = nc_sphere_sphere(p1, s1[0], pt, s2[0], margin, rc[0]);
      } else if (t1 == G_CAPSULE && t2 == G_CAPSULE) {
        // mjraw_CapsuleCapsule: nearest points of the axis segments -> sphere-sphere; PARALLEL axes: the four end caps in the order
        // (+1, -1 of capsule 1, +1, -1 of capsule 2), every end that projects inside the other segment gives a contact, at most two.
        // "Parallel" is |det| < mjMINVAL in float64; fp32 cannot resolve det below ~1e-7 ma mc, so the fp32 kernel takes the
        // parallel branch for |det| < 1e-6 ma mc (axes within 1e-3 rad), where the general formula would divide noise by noise.
        T dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        T ma = dot3(z1, z1), mb = -dot3(z1, z2), mc = dot3(z2, z2), u = -dot3(z1, dif), v = dot3(z2, dif);
        T det = ma * mc - mb * mb, x1, x2;
        if (t_abs(det) >= (sizeof(T) == 4 ? (T)1e-6 * ma * mc : Num<T>::minval())) {
          x1 = (mc * u - mb * v) / det; x2 = (ma * v - mb * u) / det;
          if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb * s1[1]) / mc; }
          else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v + mb * s1[1]) / mc; }
          if (x2 > s2[1]) { x2 = s2[1]; x1 = t_min(t_max((u - mb * s2[1]) / ma, -s1[1]), s1[1]); }
          else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = t_min(t_max((u + mb * s2[1]) / ma, -s1[1]), s1[1]); }
          T v1[3] = {p1[0] + z1[0] * x1, p1[1] + z1[1] * x1, p1[2] + z1[2] * x1};
          T v2[3] = {p2[0] + z2[0] * x2, p2[1] + z2[1] * x2, p2[2] + z2[2] * x2};
          n = nc_sphere_sphere(v1, s1[0], v2, s2[0], margin, rc[0]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (n >= 2) continue;
            bool ok;
            if (k < 2) { x1 = k == 0 ? s1[1] : -s1[1]; x2 = (v - mb * x1) / mc; ok = x2 >= -s2[1] && x2 <= s2[1]; }
            else { x2 = k == 2 ? s2[1] : -s2[1]; x1 = (u - mb * x2) / ma; ok = x1 >= -s1[1] && x1 <= s1[1]; }
            if (!ok) continue;
            T v1[3] = {p1[0] + z1[0] * x1, p1[1] + z1[1] * x1, p1[2] + z1[2] * x1};
            T v2[3] = {p2[0] + z2[0] * x2, p2[1] + z2[1] * x2, p2[2] + z2[2] * x2};
            RawCon<T> tmp;
            if (nc_sphere_sphere(v1, s1[0], v2, s2[0], margin, tmp)) { if (n == 0) rc[0] = tmp; else rc[1] = tmp; n++; }
          }
        }
      }
    }
    int total, off = gscan_small<G>(n, lane, total);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k < n) {
        int slot = ncon + off + k;
        if (slot < m.ncon_max) {
          const RawCon<T>& r = rc[k];
          T f[9] = {r.n[0], r.n[1], r.n[2], r.yh[0], r.yh[1], r.yh[2], 0, 0, 0};
          normalize3(f);                                     // mju_makeFrame
          if (dot3(f + 3, f + 3) < (T)0.25) {
            f[3] = f[4] = f[5] = 0;
            if (f[1] < (T)0.5 && f[1] > (T)-0.5) f[4] = 1; else f[5] = 1;
          }
          T d = dot3(f, f + 3);
          f[3] -= d * f[0]; f[4] -= d * f[1]; f[5] -= d * f[2];
          normalize3(f + 3);
          T* o = con + slot * CON_STRIDE;
          o[0] = r.dist; o[1] = r.pos[0]; o[2] = r.pos[1]; o[3] = r.pos[2];
#pragma unroll
          for (int a = 0; a < 6; a++) o[4 + a] = f[a];
          o[10] = m.pair_friction[5 * p];
          con_pair[slot] = p;
        }
      }
    }
    int newn = ncon + total;
    if (newn > m.ncon_max) { dropped += newn - (ncon > m.ncon_max ? ncon : m.ncon_max); }
    ncon = newn;
  }
}

// A5 collision.  Broad phase first: every candidate pair of the static list is tested against its bounding radius
// (sphere-sphere, or centre-to-plane distance for plane pairs; host-derived pair_cull = r1 + r2 + margin, negative for plane
// pairs) and the survivors are compacted - in pair order - into a list of up to G entries (kept in the not yet used efc_J
// region); the expensive narrow phase then runs once per G SURVIVORS instead of once per G candidates (humanoid: 159
// candidates, typically 10-20 survivors).  A pair outside its bounding test can produce no contact within the margin, so
// the contact list is the one the plain loop would produce.
template <typename T, int G> MJB_DEV void collision(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane;
  T *gx = w + L.geom_xpos, *gm = w + L.geom_xmat;
  int ncon = 0, dropped = 0;
  if (m.nefc_max * m.nv < G) {                                  // no room for the survivor list: plain loop
    for (int base = 0; base < m.npair; base += G) collide_pass<T, G>(c, base + lane, base + lane < m.npair, ncon, dropped);
  } else {
    int* cand = (int*)(w + L.efc_J);
    int nl = 0;
    for (int base = 0; base < m.npair; base += G) {
      const int p = base + lane;
      bool keep = false;
      if (p < m.npair) {
        const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
        const T cull = m.pair_cull[p];
        T d[3] = {gx[3 * g2] - gx[3 * g1], gx[3 * g2 + 1] - gx[3 * g1 + 1], gx[3 * g2 + 2] - gx[3 * g1 + 2]};
        if (cull < 0) {                                         // g1 is a plane: signed distance of g2's centre along its normal
          T nz[3] = {gm[9 * g1 + 2], gm[9 * g1 + 5], gm[9 * g1 + 8]};
          keep = dot3(d, nz) <= -cull * (T)1.00001 + (T)1e-6;
        } else {
          T r = cull * (T)1.00001 + (T)1e-6;
          keep = dot3(d, d) <= r * r;
        }
      }
      int total, off = gscan_small<G>(keep ? 1 : 0, lane, total);
      if (nl + total > G) {                                     // flush the list before it overflows
        gsync<G>();
        collide_pass<T, G>(c, lane < nl ? cand[lane] : 0, lane < nl, ncon, dropped);
        gsync<G>();
        nl = 0;
      }
      if (keep) cand[nl + off] = p;
      nl += total;
    }
    gsync<G>();
    if (nl > 0) collide_pass<T, G>(c, lane < nl ? cand[lane] : 0, lane < nl, ncon, dropped);
  }
  c.con_dropped += dropped;                                   // over all steps of this call (the launch's counters add it up)
  c.ncon = ncon < m.ncon_max ? ncon : m.ncon_max;
  gsync<G>();
}

// ---------------------------------------------------------------------------
// A6 constraint rows: limits + contacts; impedance, R/D, Jacobian, aref
// ---------------------------------------------------------------------------
// impedance and regulariser of one row (the state-dependent half of row_params; K and B are host-folded model constants)
template <typename T> MJB_DEV void row_imp_R(T pos, T margin, const T* solimp, T diagApprox, T& imp, T& R) {
  T dmin = t_min(t_max(solimp[0], MJB_MINIMP), MJB_MAXIMP), dmax = t_min(t_max(solimp[1], MJB_MINIMP), MJB_MAXIMP);
  T width = t_max(solimp[2], (T)0), mid = t_min(t_max(solimp[3], MJB_MINIMP), MJB_MAXIMP), power = t_max(solimp[4], (T)1);
  if (dmin == dmax || width <= Num<T>::minval()) imp = (T)0.5 * (dmin + dmax);
  else {
    T x = t_abs(pos - margin) / width;
    if (x >= 1) imp = dmax;
    else if (x <= 0) imp = dmin;
    else {
      T y;
      if (power == 1) y = x;
#ifdef MJB_SPEC_SOLIMP_POWER_1_OR_2      // every solimp power of the baked model is 1 or 2 (checked when the kernel was specialised): no powf code
      else y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
#else
      else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
      else if (x <= mid) y = t_pow(x, power) / t_pow(mid, power - 1);
      else y = 1 - t_pow(1 - x, power) / t_pow(1 - mid, power - 1);
#endif
      imp = dmin + y * (dmax - dmin);
    }
  }
  R = t_max(Num<T>::minval(), (1 - imp) * diagApprox / imp);
}
template <typename T> MJB_DEV void row_params(ModelRef<T> m, T pos, T margin, const T* solref, const T* solimp, T diagApprox, T& K, T& B, T& imp, T& R) {
  T dmin = t_min(t_max(solimp[0], MJB_MINIMP), MJB_MAXIMP), dmax = t_min(t_max(solimp[1], MJB_MINIMP), MJB_MAXIMP);
  T width = t_max(solimp[2], (T)0), mid = t_min(t_max(solimp[3], MJB_MINIMP), MJB_MAXIMP), power = t_max(solimp[4], (T)1);
  if (dmin == dmax || width <= Num<T>::minval()) imp = (T)0.5 * (dmin + dmax);
  else {
    T x = t_abs(pos - margin) / width;
    if (x >= 1) imp = dmax;
    else if (x <= 0) imp = dmin;
    else {
      T y;
      if (power == 1) y = x;
#ifdef MJB_SPEC_SOLIMP_POWER_1_OR_2      // every solimp power of the baked model is 1 or 2 (checked when the kernel was specialised): no powf code
      else y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
#else
      else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
      else if (x <= mid) y = t_pow(x, power) / t_pow(mid, power - 1);
      else y = 1 - t_pow(1 - x, power) / t_pow(1 - mid, power - 1);
#endif
      imp = dmin + y * (dmax - dmin);
    }
  }
  if (solref[0] > 0) {
    T tc = t_max(solref[0], 2 * m.timestep), dr = solref[1];
    K = 1 / t_max(Num<T>::minval(), dmax * dmax * tc * tc * dr * dr);
    B = 2 / t_max(Num<T>::minval(), dmax * tc);
  } else {
    K = -solref[0] / t_max(Num<T>::minval(), dmax * dmax);
    B = -solref[1] / t_max(Num<T>::minval(), dmax);
  }
  R = t_max(Num<T>::minval(), (1 - imp) * diagApprox / imp);
}

template <typename T, int G> MJB_DEV void make_constraint(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, cap = m.nefc_max;
  T *qpos = w + L.qpos, *qvel = w + L.qvel, *J = w + L.efc_J, *epos = w + L.efc_pos, *eD = w + L.efc_D, *earef = w + L.efc_aref;
  T *eK = w + L.efc_jar, *eB = w + L.efc_jv, *eI = w + L.efc_force;       // K, B, imp scratch until aref is known
  T *emargin = w + L.efc_KBI;
  int *etype = c.wi + L.i_efc_type, *con_pair = c.wi + L.i_con_pair;   // etype: bits 0-7 type, 8 active-at-last-factor, 9.. object id
  T *con = w + L.con, *tl = w + L.ten_length, *tj = w + L.ten_J;
  int nefc = 0, dropped = 0;
  // joint limits, then tendon limits, in ONE lane-parallel pass over the njnt + ntendon objects (rows keep that order):
  // object `o`, side 0 (lower) then 1 (upper)
  {
    const int nobj = m.njnt + m.ntendon;
    for (int base = 0; base < nobj; base += G) {
      const int k = base + lane;
      const bool isj = k < m.njnt;
      const int o = isj ? k : k - m.njnt;
      int cnt = 0;
      T dist[2] = {0, 0}, margin = 0;
      bool act[2] = {false, false};
      T rec[12];
      if (k < nobj && m.lim_i[2 * k]) {                         // one record per limit object: no index hops
        const int vi = m.lim_i[2 * k + 1];
#pragma unroll
        for (int q = 0; q < 11; q++) rec[q] = m.lim_f[12 * k + q];
        T value = isj ? qpos[vi] : tl[vi];
        margin = rec[2];
        dist[0] = value - rec[0]; dist[1] = rec[1] - value;
        act[0] = dist[0] < margin; act[1] = dist[1] < margin;
        cnt = (int)act[0] + (int)act[1];
      }
      int total, off = gscan_small<G>(cnt, lane, total);
      int row = nefc + off;
      if (cnt > 0) {
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
          if (act[sd]) {
            if (row < cap) {
              T imp, R;
              row_imp_R(dist[sd], margin, rec + 6, rec[3], imp, R);
              etype[row] = (isj ? EFC_LIMIT_JOINT : EFC_LIMIT_TENDON) | ((o * 2 + sd) << 9);
              epos[row] = dist[sd]; emargin[row] = margin; eD[row] = 1 / R; eK[row] = rec[4]; eB[row] = rec[5]; eI[row] = imp;
            }
            row++;
          }
        }
      }
      int newn = nefc + total;
      if (newn > cap) dropped += newn - (nefc > cap ? nefc : cap);
      nefc = newn;
    }
  }
  if (nefc > cap) nefc = cap;
  int nlimit = nefc;
  for (int base = 0; base < c.ncon; base += G) {
    int ci = base + lane, rows = 0, p = 0;
    T dist = 0, mu = 0, incm = 0;
    if (ci < c.ncon) {
      p = con_pair[ci] & 0xffff;
      dist = con[ci * CON_STRIDE]; mu = con[ci * CON_STRIDE + 10];
      incm = m.pair_kb[4 * p];
      if (dist < incm) rows = m.pair_condim[p] == 1 ? 1 : 4;
    }
    int total, off = gscan_small<G>(rows, lane, total);
    int row = nefc + off;
    if (ci < c.ncon) {
      bool fits = rows > 0 && row + rows <= cap;
      con_pair[ci] = p | ((fits ? row + 1 : 0) << 16);     // bits 16..: first constraint row + 1 (0 = none)
      if (fits) {
        const T tran = m.pair_kb[4 * p + 1], K = m.pair_kb[4 * p + 2], B = m.pair_kb[4 * p + 3];
        T si[5] = {m.pair_solimp[5 * p], m.pair_solimp[5 * p + 1], m.pair_solimp[5 * p + 2], m.pair_solimp[5 * p + 3], m.pair_solimp[5 * p + 4]};
        T imp, R;
        if (rows == 1) row_imp_R(dist, incm, si, tran, imp, R);
        else {
          row_imp_R(dist, incm, si, tran + mu * mu * tran, imp, R);
          R = t_max(Num<T>::minval(), 2 * mu * mu * R);
        }
        for (int r = 0; r < rows; r++) {
          etype[row + r] = (rows == 1 ? EFC_CONTACT_FRICTIONLESS : EFC_CONTACT_PYRAMIDAL) | (ci << 9);
          epos[row + r] = dist; emargin[row + r] = incm; eD[row + r] = 1 / R; eK[row + r] = K; eB[row + r] = B; eI[row + r] = imp;
        }
      }
    }
    int newn = nefc + total;
    if (newn > cap) dropped += newn - (nefc > cap ? nefc : cap);
    nefc = newn;
  }
  // contacts that did not fit leave holes only at the tail: rows are contiguous because offsets are monotone
  if (nefc > cap) {
    // recompute the true count = last fitting contact's end
    int last = nlimit;
    for (int ci = lane; ci < c.ncon; ci += G) {
      int a = (con_pair[ci] >> 16) - 1;
      if (a >= 0) { int p = con_pair[ci] & 0xffff; int e = a + (m.pair_condim[p] == 1 ? 1 : 4); if (e > last) last = e; }
    }
    nefc = gmaxi<G>(last);
  }
  c.nefc = nefc; c.efc_dropped += dropped;
  gsync<G>();
  // Jacobian rows of limits
  {
    int tot = nlimit << m.nvshift;
    for (int idx = lane; idx < tot; idx += G) {
      int r = idx >> m.nvshift, i = idx & (m.nvp - 1);
      if (i >= nv) continue;
      int id_ = etype[r] >> 9, o = id_ >> 1, s = id_ & 1;
      T sign = s == 0 ? (T)1 : (T)-1, v;
      if ((etype[r] & 0xff) == EFC_LIMIT_JOINT) v = (i == m.jnt_dofadr[o]) ? sign : (T)0;
      else v = sign * tj[o * nv + i];
      J[r * nv + i] = v;
    }
  }
  // Jacobian rows of contacts: (contact, dof) across lanes
  {
    int tot = c.ncon << m.nvshift;
    for (int idx = lane; idx < tot; idx += G) {
      int ci = idx >> m.nvshift, i = idx & (m.nvp - 1);
      if (i >= nv) continue;
      int row = (con_pair[ci] >> 16) - 1;
      if (row < 0) continue;
      int p = con_pair[ci] & 0xffff;
      int b1 = m.pair_body[2 * p], b2 = m.pair_body[2 * p + 1];
      const T* cc = con + ci * CON_STRIDE;
      T pos[3] = {cc[1], cc[2], cc[3]}, j1[3], j2[3], dj[3];
      jac_col<T>(c, b1, i, pos, j1, (T*)0);
      jac_col<T>(c, b2, i, pos, j2, (T*)0);
      dj[0] = j2[0] - j1[0]; dj[1] = j2[1] - j1[1]; dj[2] = j2[2] - j1[2];
      T jn = cc[4] * dj[0] + cc[5] * dj[1] + cc[6] * dj[2];
      if (m.pair_condim[p] == 1) J[row * nv + i] = jn;
      else {
        T mu = cc[10], nrm[3] = {cc[4], cc[5], cc[6]}, t1[3] = {cc[7], cc[8], cc[9]}, t2[3];
        cross3(t2, nrm, t1);
        T jt1 = mu * dot3(t1, dj), jt2 = mu * dot3(t2, dj);
        J[row * nv + i] = jn + jt1; J[(row + 1) * nv + i] = jn - jt1;
        J[(row + 2) * nv + i] = jn + jt2; J[(row + 3) * nv + i] = jn - jt2;
      }
    }
  }
  gsync<G>();
  // reference acceleration: aref = -B (J qvel) - K imp (pos - margin)
  const VecLds<T> xv{qvel};
  for (int r = lane; r < nefc; r += G) {
    T v = dot_lds(J + r * nv, 1, xv, nv);
    earef[r] = -eB[r] * v - eK[r] * eI[r] * (epos[r] - emargin[r]);
  }
  gsync<G>();
}

// ---------------------------------------------------------------------------
// A7 velocity stage: cvel, cdof_dot, bias forces (RNE), passive forces
// ---------------------------------------------------------------------------
template <typename T, int G> MJB_DEV void vel_bias_passive(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv;
  T *cvel = w + L.cvel, *cacc = w + L.cacc, *cfrc = w + L.cfrc, *cdof = w + L.cdof, *cdd = w + L.cdof_dot, *cin = w + L.cinert;
  T *qvel = w + L.qvel, *qpos = w + L.qpos;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; k++) { cvel[k] = 0; cfrc[k] = 0; }
    cacc[0] = cacc[1] = cacc[2] = 0; cacc[3] = -m.gravity[0]; cacc[4] = -m.gravity[1]; cacc[5] = -m.gravity[2];
  }
  gsync<G>();
  // Spatial velocities / bias accelerations are sums over the ancestor dofs (everything is expressed at the subtree COM of
  // the root).  Per-body increments are formed in parallel, the per-level loops only add the parent's total:
  //   (1) cvel[b] <- sum of the body's own cdof*qvel                  (all bodies)
  //   (2) cvel[b] += cvel[parent]                                     (per level)
  //   (3) cdof_dot[j] = (velocity just before dof j) x cdof[j]        (all dofs)
  //   (4) cacc[b] <- sum of the body's own cdof_dot*qvel              (all bodies)
  //   (5) cacc[b] += cacc[parent]                                     (per level)
  //   (6) cfrc[b] = I cacc + cvel x* (I cvel)                         (all bodies)
  for (int b = 1 + lane; b < m.nbody; b += G) {
    int da = m.body_dofadr[b], dn = m.body_dofnum[b];
    T dv[6] = {0, 0, 0, 0, 0, 0};
    for (int j = da; j < da + dn; j++) {
      T qv = qvel[j];
#pragma unroll
      for (int q = 0; q < 6; q++) dv[q] += cdof[6 * j + q] * qv;
    }
#pragma unroll
    for (int q = 0; q < 6; q++) cvel[6 * b + q] = dv[q];
  }
  gsync<G>();
  tree_forward_sum<T, G, 6>(c, cvel);
  for (int j = lane; j < nv; j += G) {
    const int p = m.dof_irec[6 * j + 1], first = m.dof_irec[6 * j + 2], gs = m.dof_irec[6 * j + 3];   // one record: no index hops
    T tmp[6] = {0, 0, 0, 0, 0, 0};
    if (gs >= 0) {                                              // gs < 0: translational dof of a free joint, cdof_dot = 0
      // gs: the three rotations of a free joint share the velocity after its translations; every other dof starts at itself
      T cv[6], cd[6];
#pragma unroll
      for (int q = 0; q < 6; q++) { cv[q] = cvel[6 * p + q]; cd[q] = cdof[6 * j + q]; }
      for (int k = first; k < gs; k++) {
        T qv = qvel[k];
#pragma unroll
        for (int q = 0; q < 6; q++) cv[q] += cdof[6 * k + q] * qv;
      }
      cross_motion(tmp, cv, cd);
    }
#pragma unroll
    for (int q = 0; q < 6; q++) cdd[6 * j + q] = tmp[q];
  }
  gsync<G>();
  for (int b = 1 + lane; b < m.nbody; b += G) {
    int da = m.body_dofadr[b], dn = m.body_dofnum[b];
    T da6[6] = {0, 0, 0, 0, 0, 0};
    for (int j = da; j < da + dn; j++) {
      T qv = qvel[j];
#pragma unroll
      for (int q = 0; q < 6; q++) da6[q] += cdd[6 * j + q] * qv;
    }
    if (m.body_depth[b] == 1) {                                  // children of the world body start from -gravity
#pragma unroll
      for (int q = 0; q < 6; q++) da6[q] += cacc[q];
    }
#pragma unroll
    for (int q = 0; q < 6; q++) cacc[6 * b + q] = da6[q];
  }
  gsync<G>();
  tree_forward_sum<T, G, 6>(c, cacc);
  for (int b = 1 + lane; b < m.nbody; b += G) {
    T in[10], cv[6], ca[6], f[6], t1[6], t2[6];
#pragma unroll
    for (int k = 0; k < 10; k++) in[k] = cin[10 * b + k];
#pragma unroll
    for (int k = 0; k < 6; k++) { cv[k] = cvel[6 * b + k]; ca[k] = cacc[6 * b + k]; }
    mul_inert_vec(f, in, ca);
    mul_inert_vec(t1, in, cv);
    cross_force(t2, cv, t1);
#pragma unroll
    for (int k = 0; k < 6; k++) cfrc[6 * b + k] = f[k] + t2[k];
  }
  gsync<G>();
  tree_backward_sum<T, G, 6>(c, cfrc, m.nround_inner, 1);
  // fluid forces per body (inertia-box model) into bfrc = [torque; force] at xipos
  T* bfrc = w + L.bfrc;
  if (m.has_fluid) {
    T *xipos = w + L.xipos, *ximat = w + L.ximat, *sc = w + L.subtree_com;
    for (int b = lane; b < m.nbody; b += G) {
      T mass = m.body_mass[b];
#pragma unroll
      for (int k = 0; k < 6; k++) bfrc[6 * b + k] = 0;
      if (b == 0 || mass < Num<T>::minval()) continue;
      T I0 = m.body_inertia[3 * b], I1 = m.body_inertia[3 * b + 1], I2 = m.body_inertia[3 * b + 2];
      T box[3] = {t_sqrt(t_max(Num<T>::minval(), I1 + I2 - I0) / mass * 6), t_sqrt(t_max(Num<T>::minval(), I0 + I2 - I1) / mass * 6), t_sqrt(t_max(Num<T>::minval(), I0 + I1 - I2) / mass * 6)};
      int r = m.body_rootid[b];
      T cv[6], off[3], lin[3], t[3], im[9], lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 6; k++) cv[k] = cvel[6 * b + k];
#pragma unroll
      for (int k = 0; k < 3; k++) off[k] = xipos[3 * b + k] - sc[3 * r + k];
#pragma unroll
      for (int k = 0; k < 9; k++) im[k] = ximat[9 * b + k];
      cross3(t, cv, off);
      lin[0] = cv[3] + t[0]; lin[1] = cv[4] + t[1]; lin[2] = cv[5] + t[2];
      mulmatTvec3(lvel, im, cv);
      mulmatTvec3(lvel + 3, im, lin);
      if (m.viscosity > 0) {
        T diam = (box[0] + box[1] + box[2]) / 3;
#pragma unroll
        for (int k = 0; k < 3; k++) { lfrc[k] = -MJB_PI * diam * diam * diam * m.viscosity * lvel[k]; lfrc[3 + k] = -3 * MJB_PI * diam * m.viscosity * lvel[3 + k]; }
      }
      if (m.density > 0) {
        lfrc[3] -= (T)0.5 * m.density * box[1] * box[2] * t_abs(lvel[3]) * lvel[3];
        lfrc[4] -= (T)0.5 * m.density * box[0] * box[2] * t_abs(lvel[4]) * lvel[4];
        lfrc[5] -= (T)0.5 * m.density * box[0] * box[1] * t_abs(lvel[5]) * lvel[5];
        T b0 = box[0] * box[0], b1 = box[1] * box[1], b2 = box[2] * box[2];
        lfrc[0] -= m.density * box[0] * (b1 * b1 + b2 * b2) * t_abs(lvel[0]) * lvel[0] / 64;
        lfrc[1] -= m.density * box[1] * (b0 * b0 + b2 * b2) * t_abs(lvel[1]) * lvel[1] / 64;
        lfrc[2] -= m.density * box[2] * (b0 * b0 + b1 * b1) * t_abs(lvel[2]) * lvel[2] / 64;
      }
      T o1[3], o2[3];
      mulmatvec3(o1, im, lfrc);
      mulmatvec3(o2, im, lfrc + 3);
#pragma unroll
      for (int k = 0; k < 3; k++) { bfrc[6 * b + k] = o1[k]; bfrc[6 * b + 3 + k] = o2[k]; }
    }
    gsync<G>();
  }
  T *qb = w + L.qfrc_bias, *qp = w + L.qfrc_passive;
  for (int i = lane; i < nv; i += G) {
    const int b = m.dof_irec[6 * i], qa = m.dof_irec[6 * i + 4];
    T v = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) v += cdof[6 * i + k] * cfrc[6 * b + k];
    qb[i] = v;
    T pf = -m.dof_frec[4 * i] * qvel[i];
    if (qa >= 0) pf -= m.dof_frec[4 * i + 1] * (qpos[qa] - m.dof_frec[4 * i + 2]);      // hinge / slide spring
    if (m.has_fluid) {
      T* xipos = w + L.xipos;
      for (int bb = 1; bb < m.nbody; bb++) {
        if (!((m.body_dofmask[bb] >> i) & 1ull)) continue;
        T pt[3] = {xipos[3 * bb], xipos[3 * bb + 1], xipos[3 * bb + 2]}, jp[3], jr[3];
        jac_col<T>(c, bb, i, pt, jp, jr);
        pf += jp[0] * bfrc[6 * bb + 3] + jp[1] * bfrc[6 * bb + 4] + jp[2] * bfrc[6 * bb + 5] + jr[0] * bfrc[6 * bb] + jr[1] * bfrc[6 * bb + 1] + jr[2] * bfrc[6 * bb + 2];
      }
    }
    qp[i] = pf;
  }
  gsync<G>();
}

// ---------------------------------------------------------------------------
// A8/A9 actuation and unconstrained acceleration
// ---------------------------------------------------------------------------
// actuator forces -> qfrc_actuator, qfrc_smooth (and the right-hand side of the M^-1 solve in qacc_smooth)
template <typename T, int G> MJB_DEV void actuation_forces(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv;
  T *ctrl = w + L.ctrl, *af = w + L.act_force, *qpos = w + L.qpos, *qvel = w + L.qvel;
  for (int a = lane; a < m.nu; a += G) {
    int grp = m.actuator_group[a];
    T force = 0;
    if (!(grp >= 0 && grp < 31 && ((MJB_OPT(c, disableactuator) >> grp) & 1))) {
      T u = ctrl[a];
      if (m.actuator_ctrllimited[a]) u = t_min(t_max(u, m.actuator_ctrlrange[2 * a]), m.actuator_ctrlrange[2 * a + 1]);
      force = m.actuator_gainprm[3 * a] * u;
      if (m.actuator_biastype[a] == 1) {
        T len = 0, vel = 0;
        if (m.actuator_trntype[a] == TRN_JOINT) {
          int j = m.actuator_trnid[2 * a];
          len = m.actuator_gear[6 * a] * qpos[m.jnt_qposadr[j]];
          vel = m.actuator_gear[6 * a] * qvel[m.jnt_dofadr[j]];
        }
        force += m.actuator_biasprm[3 * a] + m.actuator_biasprm[3 * a + 1] * len + m.actuator_biasprm[3 * a + 2] * vel;
      }
      if (m.actuator_forcelimited[a]) force = t_min(t_max(force, m.actuator_forcerange[2 * a]), m.actuator_forcerange[2 * a + 1]);
    }
    af[a] = force;
  }
  gsync<G>();
  T *qa = w + L.qfrc_actuator, *qs = w + L.qfrc_smooth, *qas = w + L.qacc_smooth, *qb = w + L.qfrc_bias, *qp = w + L.qfrc_passive;
  T *sx = w + L.site_xpos, *sm = w + L.site_xmat;
  for (int i = lane; i < nv; i += G) {
    T s = 0;
    for (int k = m.dofact_adr[i]; k < m.dofact_adr[i + 1]; k++) { int a = m.dofact_act[k]; s += m.actuator_gear[6 * a] * af[a]; }
    for (int k = 0; k < m.nsiteact; k++) {
      int a = m.siteact[k], id = m.actuator_trnid[2 * a], b = m.site_bodyid[id];
      if (!((m.body_dofmask[b] >> i) & 1ull)) continue;
      T g[6], R[9], f[3], tq[3], pt[3] = {sx[3 * id], sx[3 * id + 1], sx[3 * id + 2]}, jp[3], jr[3];
#pragma unroll
      for (int q = 0; q < 6; q++) g[q] = m.actuator_gear[6 * a + q];
#pragma unroll
      for (int q = 0; q < 9; q++) R[q] = sm[9 * id + q];
      mulmatvec3(f, R, g);
      mulmatvec3(tq, R, g + 3);
      jac_col<T>(c, b, i, pt, jp, jr);
      s += (dot3(jp, f) + dot3(jr, tq)) * af[a];
    }
    qa[i] = s;
    T fs = qp[i] - qb[i] + s;
    qs[i] = fs; qas[i] = fs;
  }
  gsync<G>();
}
template <typename T, int G> MJB_DEV void actuation_acceleration(Ctx<T>& c) {
  actuation_forces<T, G>(c);
  factor_W<T, G>(c, 0, c.w + c.lp->qacc_smooth);
}

// ---------------------------------------------------------------------------
// A10 Newton solver on the primal problem (limit / frictionless / pyramidal rows)
// ---------------------------------------------------------------------------
// (J^T f)_i for dof i = lane & 31 on the split path (one wavefront per environment, nv <= 32): the rows are split between
// the two 32-lane halves and the partial sums exchanged with v_permlane32_swap - half the serial length of the dot.
template <typename T, int G> MJB_DEV bool jt_split(int nv) { return G == 64 && nv <= 32; }
template <typename T, int G> MJB_DEV T jt_dot(const T* J, const T* f, int nefc, int nv, int lane) {
  const int h = lane >> 5, i = lane & 31, n2 = (nefc + 1) >> 1;
  const int r0 = h ? n2 : 0, cnt = h ? nefc - n2 : n2;
  T p = i < nv ? dot_lds(J + r0 * nv + i, nv, VecLds<T>{f + r0}, cnt) : (T)0;
  return half_sum(p);
}

template <typename T, int G> MJB_DEV T solver_cost(Ctx<T>& c, const T* qacc, bool store) {
  // Ma = M qacc, jar = J qacc - aref; returns Gauss + constraint cost.  store=false leaves force untouched.
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, nefc = c.nefc;
  T *M = w + L.M, *J = w + L.efc_J, *Ma = w + L.Ma, *jar = w + L.efc_jar, *aref = w + L.efc_aref, *D = w + L.efc_D, *force = w + L.efc_force;
  T *qs = w + L.qfrc_smooth, *qas = w + L.qacc_smooth;
  T part = 0;
  const VecLds<T> xq{qacc};                                   // uniform-address LDS reads (merged into ds_read2_b64): fewer issue slots than one v_readlane per element
  // rows of M and rows of J in ONE pass over the stacked matrix [M; J] (nv + nefc rows usually fit the 64 lanes)
  for (int rho = lane; rho < nv + nefc; rho += G) {
    const bool ism = rho < nv;
    const int r = ism ? rho : rho - nv;
    T s = dot_lds(ism ? M + r * nv : J + r * nv, 1, xq, nv);
    if (ism) {
      Ma[r] = s;
      part += (T)0.5 * (s - qs[r]) * (qacc[r] - qas[r]);
    } else {
      s -= aref[r];
      jar[r] = s;
      if (s < 0) { part += (T)0.5 * D[r] * s * s; if (store) force[r] = -D[r] * s; }
      else if (store) force[r] = 0;
    }
  }
  T cost = gsum<T, G>(part);
  gsync<G>();
  return cost;
}

// Active set of one constraint row at its new residual sj: bit 8 of efc_type remembers the state the last factor was built with,
// dw (kept in the jv array, free between the line search and the next [M; J] x search) = D on active rows.  Returns whether the row
// changed state.  Called by the loops that PRODUCE the residual (warm start, update) - the row's lane has sj and D in registers there,
// so the scan costs no pass of its own over jar / D.
template <typename T> MJB_DEV bool mark_active_row(int* etype, T* dw, int r, T sj, T d) {
  const int act = sj < 0 ? 1 : 0, t = etype[r];
  etype[r] = (t & ~0x100) | (act << 8);
  dw[r] = act ? d : (T)0;
  return ((t >> 8) & 1) != act;
}
template <typename T, int G> MJB_DEV T newton_direction(Ctx<T>& c, bool rebuild_wanted, T gtol2) {
  // grad, H = M + J^T D_active J (lower, Cholesky in W), search = -H^-1 grad.  Returns |grad|^2; when that is
  // already below gtol2 the (expensive) factorisation is skipped — the caller stops iterating.
  // The factor is rebuilt only when the active set changed since the last build (rebuild_wanted: first direction of the solve, or a row
  // changed state in the update that produced the current point - mark_active_row).
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, nefc = c.nefc;
  T *M = w + L.M, *W = w + L.W, *J = w + L.efc_J, *Ma = w + L.Ma, *force = w + L.efc_force;
  T *grad = w + L.grad, *search = w + L.search, *qs = w + L.qfrc_smooth;
  T gpart = 0;
  for (int rp_ = MJB_REP_N(c, REP_GRAD); rp_ > 0; rp_--) {
  gpart = 0;
  if (jt_split<T, G>(nv)) {
    const int i = lane & 31;
    T jf = jt_dot<T, G>(J, force, nefc, nv, lane);
    if (lane < nv) { T g = Ma[i] - qs[i] - jf; grad[i] = g; search[i] = -g; gpart = g * g; }      // right-hand side -g: the solve returns the search direction itself
  } else {
    for (int i = lane; i < nv; i += G) {
      T g = Ma[i] - qs[i] - dot_lds(J + i, nv, VecLds<T>{force}, nefc);
      grad[i] = g; search[i] = -g;
      gpart += g * g;
    }
  }
  }
  T gn = gsum<T, G>(gpart);
  const bool rebuild = (MJB_SWEEP_EXCLUDE != 1 && fused_inverse_path<T, G>(nv)) || rebuild_wanted;   // the sweep path keeps no factor
  gsync<G>();
  if (gn < gtol2) return gn;
#if defined(MJB_PROFILE) && !defined(MJB_HOST_EMU)
  c.pacc[PH_CNT_DIR] += 1; if (rebuild) c.pacc[PH_CNT_FACT] += 1;
#endif
  for (int rp_ = MJB_REP_N(c, rebuild ? REP_CHOL : REP_REUSE); rp_ > 1; rp_--) {      // diagnostic build: the solve twice, right-hand side restored
    if (rebuild) factor_W<T, G>(c, 1, search);
    else if (fused_inverse_path<T, G>(nv)) { gsync<G>(); mfma_solve32(W, w + L.tmp, search, nv, lane); }
    for (int i = lane; i < nv; i += G) search[i] = -grad[i];
    gsync<G>();
  }
  if (rebuild) factor_W<T, G>(c, 1, search);
  else if (fused_inverse_path<T, G>(nv)) { gsync<G>(); mfma_solve32(W, w + L.tmp, search, nv, lane); }
  else chol_solve<T, G>(W, w + L.tmp, search, nv, lane);
  return gn;                                                   // (every solve ends with a group sync; H x = -g, so x is the direction: no negation pass)
}

template <typename T, int G> MJB_DEV void solve_constraints(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, nefc = c.nefc;
  T *qacc = w + L.qacc, *ws = w + L.qacc_ws, *qas = w + L.qacc_smooth, *qc = w + L.qfrc_constraint;
  c.niter = 0;
  if (nefc == 0) {
    for (int i = lane; i < nv; i += G) { T a = qas[i]; qacc[i] = a; ws[i] = a; qc[i] = 0; }
    gsync<G>();
    return;
  }
  T *M = w + L.M, *J = w + L.efc_J, *Ma = w + L.Ma, *jar = w + L.efc_jar, *jv = w + L.efc_jv, *D = w + L.efc_D, *force = w + L.efc_force;
  T *search = w + L.search, *Mv = w + L.Mv, *qs = w + L.qfrc_smooth;
  int* etype = c.wi + L.i_efc_type;
  // warmstart(): best of (qacc_warmstart, qacc_smooth) as the starting point.  The unconstrained point is evaluated FIRST so
  // that in the common case (the warm start wins) Ma / jar / force are already those of the chosen point: two cost
  // evaluations instead of three; same costs, same decision (warm start only if strictly cheaper).
  // Both candidates in ONE pass over the stacked [M; J] (every matrix element read once, two FMAs): M x / J x - aref of the
  // unconstrained point go to Ma / jar, those of the previous solution to Mv / jv (free until the first line search); the winner's
  // are then where the iterations expect them.  Per-row arithmetic and reduction order are those of solver_cost(): same costs, same decision.
  T cost;
  for (int rp_ = MJB_REP_N(c, REP_WARM); rp_ > 0; rp_--) {
    T *aref = w + L.efc_aref;
    T pa = 0, pb = 0;
    const VecLds<T> xa{qas}, xb{ws};
    for (int rho = lane; rho < nv + nefc; rho += G) {
      const bool ism = rho < nv;
      const int r = ism ? rho : rho - nv;
      const T* row = ism ? M + r * nv : J + r * nv;
      T sa = dot_lds(row, 1, xa, nv), sb = dot_lds(row, 1, xb, nv);
      if (ism) {
        Ma[r] = sa; Mv[r] = sb;
        pa += (T)0.5 * (sa - qs[r]) * (qas[r] - qas[r]);
        pb += (T)0.5 * (sb - qs[r]) * (ws[r] - qas[r]);
      } else {
        sa -= aref[r]; sb -= aref[r];
        jar[r] = sa; jv[r] = sb;
        if (sa < 0) pa += (T)0.5 * D[r] * sa * sa;
        if (sb < 0) pb += (T)0.5 * D[r] * sb * sb;
      }
    }
    const T cost_a = gsum<T, G>(pa), cost_b = gsum<T, G>(pb);
    gsync<G>();
    const bool wsw = cost_b < cost_a;                            // warm start only if strictly cheaper
    cost = wsw ? cost_b : cost_a;
    for (int i = lane; i < nv; i += G) { qacc[i] = wsw ? ws[i] : qas[i]; if (wsw) Ma[i] = Mv[i]; }
    for (int r = lane; r < nefc; r += G) {
      const T sj = wsw ? jv[r] : jar[r], d = D[r];
      if (wsw) jar[r] = sj;
      force[r] = sj < 0 ? -d * sj : (T)0;
      mark_active_row<T>(etype, jv, r, sj, d);                   // (jv's candidate residual has been read: the array is free for dw)
    }
    gsync<G>();
  }
  bool rebuild_wanted = true;                                    // first direction of the solve
  const T scale = 1 / (m.meaninertia * (T)(nv > 1 ? nv : 1));
  // the run-time options are read ONCE per solve (they sit in device memory behind c.mp: inside the loop every iteration paid the
  // scalar loads again - the group syncs are fences - and a division for the gradient tolerance)
  const int max_iter = MJB_OPT(c, iterations);
  const T tolerance = MJB_OPT(c, tolerance), gtol = tolerance / scale, gtol2 = gtol * gtol;
  for (int iter = 0; iter < max_iter; iter++) {
    MJB_STAMP(c, PH_SOLVE);
    T gn = newton_direction<T, G>(c, rebuild_wanted, gtol2);
    MJB_STAMP(c, PH_SOL_DIR);
    if (gn < gtol2) break;
    // Mv, jv and the Gauss part of the 1-D quadratic
    T p1 = 0, p2 = 0;
    const VecLds<T> xs{search};
    for (int rp_ = MJB_REP_N(c, REP_MV); rp_ > 0; rp_--) {
    p1 = 0; p2 = 0;
    for (int rho = lane; rho < nv + nefc; rho += G) {          // stacked [M; J] x search, one pass
      const bool ism = rho < nv;
      const int r = ism ? rho : rho - nv;
      T sacc = dot_lds(ism ? M + r * nv : J + r * nv, 1, xs, nv);
      if (ism) { Mv[r] = sacc; p1 += search[r] * (Ma[r] - qs[r]); p2 += search[r] * sacc; }
      else jv[r] = sacc;
    }
    }
    T g1 = gsum<T, G>(p1), g2 = gsum<T, G>(p2);
    gsync<G>();
    MJB_STAMP(c, PH_SOL_MV);
    // Exact line search on the convex piecewise-quadratic f(alpha) = Gauss + sum_r 1/2 D_r min(0, jar_r + alpha jv_r)^2
    // (f' continuous, piecewise linear, increasing): safeguarded Newton on f'.  A pure Newton step across which NO row
    // changes state lands exactly on the root of the current linear piece: stop without a confirming evaluation.
    // With nefc <= G every lane keeps its row (jar, jv, D jv) in registers: an iteration is a few VALU ops, two DPP
    // reductions and a ballot.
    T alpha = 0;
    for (int rp_ = MJB_REP_N(c, REP_LS); rp_ > 0; rp_--) {
      alpha = 0;
      const bool inreg = nefc <= G;
      T x0r = 0, jwr = 0, djr = 0;
      if (inreg && lane < nefc) { x0r = jar[lane]; jwr = jv[lane]; djr = D[lane] * jwr; }
      T lo = 0, hi = -1;
      for (int it = 0; it < 50; it++) {
        T d1p = 0, d2p = 0;
        if (inreg) {
          T x = x0r + alpha * jwr;
          if (x < 0) { d1p = djr * x; d2p = djr * jwr; }
        } else {
          for (int r = lane; r < nefc; r += G) {
            T x = jar[r] + alpha * jv[r];
            if (x < 0) { T dj = D[r] * jv[r]; d1p += dj * x; d2p += dj * jv[r]; }
          }
        }
        T d1 = g1 + alpha * g2 + gsum<T, G>(d1p), d2 = g2 + gsum<T, G>(d2p);
        if (it == 0 && d1 >= 0) { alpha = 0; break; }
        if (d2 < Num<T>::minval()) break;
        if (t_abs(d1) < (sizeof(T) == 4 ? (T)1e-6 : (T)1e-14) * (t_abs(g1) + Num<T>::minval())) break;
        if (d1 < 0) lo = alpha; else hi = alpha;
        T an = alpha - d1 / d2;
        bool newton = true;
        if (an <= lo || (hi >= 0 && an >= hi)) { an = hi >= 0 ? (T)0.5 * (lo + hi) : 2 * alpha + (T)1e-3; newton = false; }
        if (an == alpha) break;
        bool flip = false;
        if (inreg) flip = ((x0r + alpha * jwr) < 0) != ((x0r + an * jwr) < 0);
        else for (int r = lane; r < nefc; r += G) flip |= ((jar[r] + alpha * jv[r]) < 0) != ((jar[r] + an * jv[r]) < 0);
        const bool anyflip = gany<G>(flip);
        alpha = an;
#if defined(MJB_PROFILE) && !defined(MJB_HOST_EMU)
        c.pacc[PH_CNT_LS] += 1;
#endif
        if (newton && !anyflip) break;
      }
    }
    MJB_STAMP(c, PH_SOL_LS);
    if (alpha == 0) break;
    T part = 0;
    for (int i = lane; i < nv; i += G) {
      T a = qacc[i] + alpha * search[i], ma = Ma[i] + alpha * Mv[i];
      qacc[i] = a; Ma[i] = ma;
      part += (T)0.5 * (ma - qs[i]) * (a - qas[i]);
    }
    bool chg = false;
    for (int r = lane; r < nefc; r += G) {
      const T sj = jar[r] + alpha * jv[r], d = D[r];
      jar[r] = sj;
      if (sj < 0) { part += (T)0.5 * d * sj * sj; force[r] = -d * sj; } else force[r] = 0;
      chg = mark_active_row<T>(etype, jv, r, sj, d) || chg;      // the next direction's active set, while sj and D are in registers
    }
    T old = cost;
    cost = gsum<T, G>(part);
    rebuild_wanted = gany<G>(chg);
    gsync<G>();
    c.niter = iter + 1;
    if (scale * (old - cost) < tolerance) break;
  }
  if (jt_split<T, G>(nv)) {
    T jf = jt_dot<T, G>(J, force, nefc, nv, lane);
    if (lane < nv) { qc[lane] = jf; ws[lane] = qacc[lane]; }
  } else {
    for (int i = lane; i < nv; i += G) { qc[i] = dot_lds(J + i, nv, VecLds<T>{force}, nefc); ws[i] = qacc[i]; }
  }
  gsync<G>();
}

// A12 sensors of the last forward pass: jointpos, gyro, framequat, accelerometer (lanes over sensors)
template <typename T, typename TS, int G> MJB_DEV void sensors(const Ctx<T>& c, TS* dst) {
  ModelRef<T> m = MJB_MODEL_OF(c.mp); LayRef L = *c.lp; const T* w = c.w; const int lane = c.lane;
  for (int s = lane; s < m.nsensor; s += G) {
    TS* out = dst + m.sensor_adr[s];
    int id = m.sensor_objid[s], st = m.sensor_type[s];
    if (st == SENS_JOINTPOS) { out[0] = (TS)w[L.qpos + m.jnt_qposadr[id]]; continue; }
    int b = m.site_bodyid[id];
    T R[9];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = w[L.site_xmat + 9 * id + k];
    if (st == SENS_GYRO) {
      T wv[3] = {w[L.cvel + 6 * b], w[L.cvel + 6 * b + 1], w[L.cvel + 6 * b + 2]}, o3[3];
      mulmatTvec3(o3, R, wv);
      out[0] = (TS)o3[0]; out[1] = (TS)o3[1]; out[2] = (TS)o3[2];
    } else if (st == SENS_FRAMEQUAT) {
      T bq[4] = {w[L.xquat + 4 * b], w[L.xquat + 4 * b + 1], w[L.xquat + 4 * b + 2], w[L.xquat + 4 * b + 3]};
      T sq[4] = {m.site_quat[4 * id], m.site_quat[4 * id + 1], m.site_quat[4 * id + 2], m.site_quat[4 * id + 3]}, q[4];
      quat_mul(q, bq, sq);
      quat_normalize(q);
      out[0] = (TS)q[0]; out[1] = (TS)q[1]; out[2] = (TS)q[2]; out[3] = (TS)q[3];
    } else if (st == SENS_ACCEL) {
      // full com-based acceleration of the body = bias part kept from the velocity stage + sum of cdof * qacc over its dofs
      T cv[6], ca[6];
#pragma unroll
      for (int k = 0; k < 6; k++) { cv[k] = w[L.cvel + 6 * b + k]; ca[k] = w[L.cacc + 6 * b + k]; }
      for (int i = 0; i < m.nv; i++) {
        if (!((m.body_dofmask[b] >> i) & 1ull)) continue;
        T qa = w[L.qacc + i];
#pragma unroll
        for (int k = 0; k < 6; k++) ca[k] += w[L.cdof + 6 * i + k] * qa;
      }
      int r = m.body_rootid[b];
      T dif[3] = {w[L.site_xpos + 3 * id] - w[L.subtree_com + 3 * r], w[L.site_xpos + 3 * id + 1] - w[L.subtree_com + 3 * r + 1], w[L.site_xpos + 3 * id + 2] - w[L.subtree_com + 3 * r + 2]};
      T t[3], vlin[3], alin[3], wl[3], vl[3], al[3], cr[3];
      cross3(t, cv, dif); vlin[0] = cv[3] + t[0]; vlin[1] = cv[4] + t[1]; vlin[2] = cv[5] + t[2];
      cross3(t, ca, dif); alin[0] = ca[3] + t[0]; alin[1] = ca[4] + t[1]; alin[2] = ca[5] + t[2];
      mulmatTvec3(wl, R, cv); mulmatTvec3(vl, R, vlin); mulmatTvec3(al, R, alin);
      cross3(cr, wl, vl);
      out[0] = (TS)(al[0] + cr[0]); out[1] = (TS)(al[1] + cr[1]); out[2] = (TS)(al[2] + cr[2]);
    } else { out[0] = 0; }
  }
}

// ---------------------------------------------------------------------------
// mj_forward for one environment (state in LDS)
// ---------------------------------------------------------------------------
// The three stages of mj_forward, separately callable: mjd_transitionFD skips the stages a perturbed column cannot change
// (MuJoCo's mj_stepSkip: ctrl columns keep the position and velocity stages, velocity columns the position stage).
// Diagnostic build -DMJB_PHASE_REPEAT (scripts/gpu_phase_pmc.py): the phase whose index equals StepArgs::repeat_phase runs TWICE.
// Every phase wrapped below recomputes its outputs from LDS inputs it does not modify (idempotent), so the state evolves exactly
// as in the product kernel and the DIFFERENCE of the hardware counters of two passes (repeat k vs no repeat) is phase k's own
// instruction count, active-lane cycles and executed flops.  The product build compiles the plain call.
#if defined(MJB_PHASE_REPEAT) && !defined(MJB_HOST_EMU)
#define MJB_PHASE_CALL(c, k, call) do { call; if ((c).rep == (k)) { gsync<G>(); call; } } while (0)
#else
#define MJB_PHASE_CALL(c, k, call) do { call; } while (0)
#endif
template <typename T, int G> MJB_DEV void forward_position(Ctx<T>& c) {
  MJB_STAMP(c, PH_OTHER);
  MJB_PHASE_CALL(c, PH_KIN, (kinematics<T, G>(c))); MJB_STAMP(c, PH_KIN);
  MJB_PHASE_CALL(c, PH_COM, (com_pos<T, G>(c))); MJB_STAMP(c, PH_COM);
  MJB_PHASE_CALL(c, PH_COLL, (collision<T, G>(c))); MJB_STAMP(c, PH_COLL);
  MJB_PHASE_CALL(c, PH_CRB, (crb_factor<T, G>(c))); MJB_STAMP(c, PH_CRB);
}
// constraint rows (their reference acceleration depends on qvel) + bias / passive forces
template <typename T, int G> MJB_DEV void forward_velocity(Ctx<T>& c) {
  MJB_PHASE_CALL(c, PH_CONS, (make_constraint<T, G>(c))); MJB_STAMP(c, PH_CONS);
  MJB_PHASE_CALL(c, PH_VEL, (vel_bias_passive<T, G>(c))); MJB_STAMP(c, PH_VEL);
}
template <typename T, int G> MJB_DEV void forward_acceleration(Ctx<T>& c) {
  MJB_PHASE_CALL(c, PH_ACT, (actuation_acceleration<T, G>(c))); MJB_STAMP(c, PH_ACT);
  solve_constraints<T, G>(c); MJB_STAMP(c, PH_SOLVE);
  if (c.mp->nsensor > 0) { sensors<T, T, G>(c, c.w + c.lp->sens); gsync<G>(); }   // like mj_forward: sensors see the pre-integration state
}
template <typename T, int G> MJB_DEV void forward(Ctx<T>& c) {
  forward_position<T, G>(c);
  forward_velocity<T, G>(c);
  if (c.skip_dynamics) return;                              // mj_inverse: position + velocity stages only (inverse_dynamics() follows)
  forward_acceleration<T, G>(c);
}

// A16 position integration for the joints of one environment (lanes over joints)
template <typename T, int G> MJB_DEV void integrate_pos(ModelRef<T> m, T* qpos, const T* qvel, T h, int lane) {
  for (int j = lane; j < m.njnt; j += G) {
    int qa = m.jnt_irec[6 * j + 1], da = m.jnt_irec[6 * j + 2];
    if (m.jnt_irec[6 * j] == JNT_FREE) {
      qpos[qa] += h * qvel[da]; qpos[qa + 1] += h * qvel[da + 1]; qpos[qa + 2] += h * qvel[da + 2];
      T q[4] = {qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]}, wv[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]};
      quat_integrate(q, wv, h);
      qpos[qa + 3] = q[0]; qpos[qa + 4] = q[1]; qpos[qa + 5] = q[2]; qpos[qa + 6] = q[3];
    } else qpos[qa] += h * qvel[da];
  }
}

// A11 Euler with implicit joint damping (mj_Euler)
// inv != nullptr (two-wave step kernel, fp32, nv <= 32): -(M + h D)^-1 already sits in that accumulator
template <typename T, int G> MJB_DEV void euler(Ctx<T>& c, const mjb_f16v* inv = nullptr) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv;
  T *qacc = w + L.qacc, *qvel = w + L.qvel, *qpos = w + L.qpos, *tmpv = w + L.Mv, *M = w + L.M, *W = w + L.W;
  T h = m.timestep;
  if (m.has_damping) {
    T *qs = w + L.qfrc_smooth, *qc = w + L.qfrc_constraint;
#if defined(MJB_PHASE_REPEAT) && !defined(MJB_HOST_EMU)
    if (c.rep == PH_INTEG) {                                   // the implicit-damping solve twice (right-hand side rebuilt: idempotent)
      for (int i = lane; i < nv; i += G) tmpv[i] = qs[i] + qc[i];
      gsync<G>();
      factor_W<T, G>(c, 2, tmpv);
      gsync<G>();
    }
#endif
    for (int i = lane; i < nv; i += G) tmpv[i] = qs[i] + qc[i];
    gsync<G>();
#ifndef MJB_HOST_EMU
    if constexpr (sizeof(T) == 4 && G == 64) {
      if (inv) mfma_sweep_apply32(*inv, w + L.tmp, nv, lane, tmpv);
      else factor_W<T, G>(c, 2, tmpv);
    } else
#endif
    factor_W<T, G>(c, 2, tmpv);
  } else {
    for (int i = lane; i < nv; i += G) tmpv[i] = qacc[i];
    gsync<G>();
  }
  for (int i = lane; i < nv; i += G) qvel[i] += h * tmpv[i];
  gsync<G>();
  integrate_pos<T, G>(m, qpos, qvel, h, lane);
  gsync<G>();
}

// A11 RK4 (mj_RungeKutta, N = 4) as a stage machine so that forward() has a single call site.
// rk scratch: X0q[nq] X0v[nv] Fv[4nv] Fa[4nv] dv[nv].  Call after the forward pass of stage st (0..3).
template <typename T, int G> MJB_DEV void rk4_stage(Ctx<T>& c, int st) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, nq = m.nq;
  T *qacc = w + L.qacc, *qvel = w + L.qvel, *qpos = w + L.qpos;
  T *X0q = w + L.rk, *X0v = X0q + nq, *Fv = X0v + nv, *Fa = Fv + 4 * nv, *dv = Fa + 4 * nv;
  const T h = m.timestep;
  if (st == 0) {
    for (int i = lane; i < nq; i += G) X0q[i] = qpos[i];
    for (int i = lane; i < nv; i += G) X0v[i] = qvel[i];
  }
  for (int i = lane; i < nv; i += G) { Fv[st * nv + i] = qvel[i]; Fa[st * nv + i] = qacc[i]; }
  gsync<G>();
  // tableau: stage st+1 state uses row st of A = [[1/2],[0,1/2],[0,0,1]]; the last stage combines with B = [1/6,1/3,1/3,1/6]
  for (int i = lane; i < nv; i += G) {
    T sv, sa;
    if (st < 3) { T a = st == 2 ? (T)1 : (T)0.5; sv = a * Fv[st * nv + i]; sa = a * Fa[st * nv + i]; }
    else {
      sv = (Fv[i] + 2 * Fv[nv + i] + 2 * Fv[2 * nv + i] + Fv[3 * nv + i]) * (T)(1.0 / 6);
      sa = (Fa[i] + 2 * Fa[nv + i] + 2 * Fa[2 * nv + i] + Fa[3 * nv + i]) * (T)(1.0 / 6);
    }
    dv[i] = sv; qvel[i] = X0v[i] + h * sa;
  }
  for (int i = lane; i < nq; i += G) qpos[i] = X0q[i];
  gsync<G>();
  integrate_pos<T, G>(m, qpos, dv, h, lane);
  gsync<G>();
}

// ---------------------------------------------------------------------------
// counter-based uniform random ctrl (Philox4x32-10): identical to oracle/mjo.c
// ---------------------------------------------------------------------------
MJB_DEV unsigned philox_first(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  for (int r = 0; r < 10; r++) {
    unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}
template <typename T, int G> MJB_DEV void random_ctrl(ModelRef<T> m, T* ctrl, unsigned seed, unsigned env, unsigned step, T scale, int lane) {
  for (int a = lane; a < m.nu; a += G) {
    unsigned r = philox_first(env, step, (unsigned)a, 0u, seed, 0x5EEDu);
    T u = (T)(r >> 8) * (T)(1.0 / 16777216.0);
    T lo = -1, hi = 1;
    if (m.actuator_ctrllimited[a]) { lo = m.actuator_ctrlrange[2 * a]; hi = m.actuator_ctrlrange[2 * a + 1]; }
    ctrl[a] = (T)0.5 * (lo + hi) + (T)0.5 * (hi - lo) * scale * (2 * u - 1);
  }
}

// Linear state-feedback controller evaluated on the device (the LQR law of the reference's examples,
// examples/humanoid/controllers/lqr.py:147-170): ctrl = clip(u0 - K dx), dx = [q (-) q0 ; qvel - v0] in tangent space.
template <typename T, int G> MJB_DEV void feedback_ctrl(Ctx<T>& c, ArgsRef a, unsigned env, unsigned step) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, nu = m.nu;
  auto K = (const T MJB_CONST*)a.fb_K; auto u0 = (const T MJB_CONST*)a.fb_u0;
  auto q0 = (const T MJB_CONST*)a.fb_q0; auto v0 = (const T MJB_CONST*)a.fb_v0;
  T *dx = w + L.grad, *qpos = w + L.qpos, *qvel = w + L.qvel, *ctrl = w + L.ctrl;   // grad|search are contiguous: 2 nv of scratch
  for (int j = lane; j < m.njnt; j += G) {
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    if (m.jnt_type[j] == JNT_FREE) {
      dx[da] = qpos[qa] - q0[qa]; dx[da + 1] = qpos[qa + 1] - q0[qa + 1]; dx[da + 2] = qpos[qa + 2] - q0[qa + 2];
      T qa4[4] = {q0[qa + 3], q0[qa + 4], q0[qa + 5], q0[qa + 6]}, qb4[4] = {qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]}, r[3];
      quat_sub(r, qa4, qb4);
      dx[da + 3] = r[0]; dx[da + 4] = r[1]; dx[da + 5] = r[2];
    } else dx[da] = qpos[qa] - q0[qa];
  }
  for (int i = lane; i < nv; i += G) dx[nv + i] = qvel[i] - v0[i];
  gsync<G>();
  for (int act = lane; act < nu; act += G) {
    T u = u0[act];
    for (int k = 0; k < 2 * nv; k++) u -= K[act * 2 * nv + k] * dx[k];
    if (a.fb_nsteps > 0) {
      const long idx = ((long)step + (long)env * a.fb_env_stride) % a.fb_nsteps;
      u += ((const T MJB_CONST*)a.fb_noise_std)[act] * ((const T MJB_CONST*)a.fb_noise_tab)[idx * nu + act];
    }
    if (m.actuator_ctrllimited[act]) u = t_min(t_max(u, m.actuator_ctrlrange[2 * act]), m.actuator_ctrlrange[2 * act + 1]);
    ctrl[act] = u;
  }
  gsync<G>();
}

// mj_inverse after forward() with skip_dynamics (reference setpoints.py:29-31): the constraint force follows in closed
// form from jar = J qacc - aref (one-sided quadratic rows: force = -D jar where jar < 0), then
// qfrc_inverse = M qacc + qfrc_bias - qfrc_passive - J^T force.  Also writes the dense actuator moment [nu, nv]
// (joint transmission: gear at the joint's dof; site transmission: gear wrench in the site frame through the site Jacobian).
template <typename T, typename TS, int G> MJB_DEV void inverse_dynamics(Ctx<T>& c, TS* inv_out, TS* moment_out) {
  MJB_ENV(c); T* w = c.w; const int lane = c.lane, nv = m.nv, nefc = c.nefc;
  T *J = w + L.efc_J, *aref = w + L.efc_aref, *D = w + L.efc_D, *force = w + L.efc_force, *qacc = w + L.qacc, *M = w + L.M;
  T *qb = w + L.qfrc_bias, *qp = w + L.qfrc_passive, *qc = w + L.qfrc_constraint;
  for (int r = lane; r < nefc; r += G) {
    T jar = dot_lds(J + r * nv, 1, VecLds<T>{qacc}, nv) - aref[r];
    force[r] = jar < 0 ? -D[r] * jar : (T)0;
  }
  gsync<G>();
  for (int i = lane; i < nv; i += G) {
    T fc = dot_lds(J + i, nv, VecLds<T>{force}, nefc);
    T ma = dot_lds(M + i * nv, 1, VecLds<T>{qacc}, nv);
    qc[i] = fc;
    inv_out[i] = (TS)(ma + qb[i] - qp[i] - fc);
  }
  T *sx = w + L.site_xpos, *sm = w + L.site_xmat;
  for (int idx = lane; idx < m.nu * nv; idx += G) {
    int a = idx / nv, i = idx - a * nv, id = m.actuator_trnid[2 * a];
    T v = 0;
    if (m.actuator_trntype[a] == TRN_JOINT) v = (i == m.jnt_dofadr[id]) ? m.actuator_gear[6 * a] : (T)0;
    else {
      int b = m.site_bodyid[id];
      if ((m.body_dofmask[b] >> i) & 1ull) {
        T g[6], R[9], f[3], tq[3], pt[3] = {sx[3 * id], sx[3 * id + 1], sx[3 * id + 2]}, jp[3], jr[3];
#pragma unroll
        for (int q = 0; q < 6; q++) g[q] = m.actuator_gear[6 * a + q];
#pragma unroll
        for (int q = 0; q < 9; q++) R[q] = sm[9 * id + q];
        mulmatvec3(f, R, g);
        mulmatvec3(tq, R, g + 3);
        jac_col<T>(c, b, i, pt, jp, jr);
        v = dot3(jp, f) + dot3(jr, tq);
      }
    }
    moment_out[idx] = (TS)v;
  }
  gsync<G>();
}

// A13 bad-state guard
template <typename T, int G> MJB_DEV bool group_bad(const T* x, int n, int lane) {
  int bad = 0;
  for (int i = lane; i < n; i += G) { T v = x[i]; if (!(t_abs(v) <= (T)1e10)) bad = 1; }
  return gany<G>(bad != 0);
}
template <typename T, int G> MJB_DEV void reset_state(Ctx<T>& c) {
  MJB_ENV(c); T* w = c.w;
  for (int i = c.lane; i < m.nq; i += G) w[L.qpos + i] = m.qpos0[i];
  for (int i = c.lane; i < m.nv; i += G) { w[L.qvel + i] = 0; w[L.qacc + i] = 0; w[L.qacc_ws + i] = 0; }
  for (int i = c.lane; i < m.nu; i += G) w[L.ctrl + i] = 0;
  gsync<G>();
}

// flat observation (keys in sorted order, reference observations.py:171-174):
// bodies_pos, ctrl, geoms_pos, qpos, qvel, sensordata, sites_pos, subtree_com, time
template <typename T, typename TS, int G> MJB_DEV void write_obs(const Ctx<T>& c, ObsRef s, double time, TS* out) {
  ModelRef<T> m = MJB_MODEL_OF(c.mp); LayRef L = *c.lp; const T* w = c.w; const int lane = c.lane;
  int o = 0;
  const T* bsrc = (s.flags & 64) ? w + L.xipos : w + L.xpos;
  for (int i = lane; i < 3 * s.nbody; i += G) out[o + i] = (TS)bsrc[3 * s.body_ids[i / 3] + i % 3];
  o += 3 * s.nbody;
  if (s.flags & 4) { for (int i = lane; i < m.nu; i += G) out[o + i] = (TS)w[L.ctrl + i]; o += m.nu; }
  for (int i = lane; i < 3 * s.ngeom; i += G) out[o + i] = (TS)w[L.geom_xpos + 3 * s.geom_ids[i / 3] + i % 3];
  o += 3 * s.ngeom;
  if (s.flags & 1) { for (int i = lane; i < m.nq; i += G) out[o + i] = (TS)w[L.qpos + i]; o += m.nq; }
  if (s.flags & 2) { for (int i = lane; i < m.nv; i += G) out[o + i] = (TS)w[L.qvel + i]; o += m.nv; }
  if (s.flags & 8) { for (int i = lane; i < m.nsensordata; i += G) out[o + i] = (TS)w[L.sens + i]; o += m.nsensordata; }
  for (int i = lane; i < 3 * s.nsite; i += G) out[o + i] = (TS)w[L.site_xpos + 3 * s.site_ids[i / 3] + i % 3];
  o += 3 * s.nsite;
  for (int i = lane; i < 3 * s.nsubtree; i += G) out[o + i] = (TS)w[L.subtree_com + 3 * s.subtree_ids[i / 3] + i % 3];
  o += 3 * s.nsubtree;
  if (s.flags & 16) { if (lane == 0) out[o] = (TS)time; o += 1; }
}

// ---------------------------------------------------------------------------
// the per-environment driver: load -> nstep x (ctrl, forward, integrate) -> store
// ---------------------------------------------------------------------------
// Host mirror inside the step kernel (StepArgs::mirror, small batches): layout of mjb_host_view - qpos | qvel | ctrl | qacc |
// qacc_warmstart | time, each [batch, n] float64, then one word of engine flags.
template <typename T, int G, typename MRef>
MJB_DEV void mirror_in(MRef m, LayRef L, T* w, const double* mir, int mask, int env, int batch, int lane, double& time) {
  const size_t B = (size_t)batch, nq = m.nq, nv = m.nv, nu = m.nu;
  const double *mq = mir + (size_t)env * nq, *mv = mir + B * nq + (size_t)env * nv, *mc = mir + B * (nq + nv) + (size_t)env * nu;
  const double *ma = mir + B * (nq + nv + nu) + (size_t)env * nv, *mw = ma + B * nv, *mt = mir + B * (nq + 3 * nv + nu) + env;
  if (mask & 1) for (int i = lane; i < (int)nq; i += G) w[L.qpos + i] = (T)mq[i];
  if (mask & 2) for (int i = lane; i < (int)nv; i += G) w[L.qvel + i] = (T)mv[i];
  if (mask & 4) for (int i = lane; i < (int)nu; i += G) w[L.ctrl + i] = (T)mc[i];
  if (mask & 8) for (int i = lane; i < (int)nv; i += G) w[L.qacc + i] = (T)ma[i];
  if (mask & 16) for (int i = lane; i < (int)nv; i += G) w[L.qacc_ws + i] = (T)mw[i];
  if (mask & 32) time = *mt;
}
template <typename T, int G, typename MRef>
MJB_DEV void mirror_out(MRef m, LayRef L, const T* w, double* mir, int env, int batch, int lane, double time, int fl, unsigned long long seq = 0) {
  const size_t B = (size_t)batch, nq = m.nq, nv = m.nv, nu = m.nu;
  double *mq = mir + (size_t)env * nq, *mv = mir + B * nq + (size_t)env * nv, *mc = mir + B * (nq + nv) + (size_t)env * nu;
  double *ma = mir + B * (nq + nv + nu) + (size_t)env * nv, *mw = ma + B * nv, *mt = mir + B * (nq + 3 * nv + nu) + env;
  for (int i = lane; i < (int)nq; i += G) mq[i] = (double)w[L.qpos + i];
  for (int i = lane; i < (int)nv; i += G) { mv[i] = (double)w[L.qvel + i]; ma[i] = (double)w[L.qacc + i]; mw[i] = (double)w[L.qacc_ws + i]; }
  for (int i = lane; i < (int)nu; i += G) mc[i] = (double)w[L.ctrl + i];
  if (lane == 0) {
    *mt = time;
    if (fl) mir[B * (nq + 3 * nv + nu + 1)] = -1.0;           // "engine flags changed": the host fetches the sticky word (no read-modify-write over the bus)
  }
#ifndef MJB_HOST_EMU
  if (seq != 0) {
    // completion word of this environment (behind the flags word): every lane's state stores above are complete at system scope once
    // the wave has passed the fence; then ONE release store publishes them to the polling host
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store((unsigned long long*)(mir + B * (nq + 3 * nv + nu + 1) + 1) + env, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#endif
}

// Hand-over of an environment between the chunks of ONE launch (ticket mode of k_step, mjb_kernels.hpp): the wave that ends chunk
// k - 1 and the wave that starts chunk k may sit on different XCDs, so the words travel as 64-bit (value, tag) pairs written and
// read with agent-scope relaxed atomics (single-copy atomic, coherent across the XCDs' L2s): tag = tagbase + step index is unique
// per launch and hand-over, every word validates itself, no flag, no fence, and the producer never waits for its stores.
//   xfer[env, j],  j over  qpos | qvel | qacc_warmstart | qacc | time (lo, hi)          (ctrl is either regenerated every step or
//   constant over the launch and read from its array)
#ifndef MJB_HOST_EMU
MJB_DEV unsigned long long xfer_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
MJB_DEV void xfer_st(unsigned long long* p, unsigned bits, unsigned tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Sticky engine flags of the batch (bit 0 contacts dropped, 1 rows dropped, 2 bad-state reset, 3 hand-over timed out): the device word
// (atomic OR, read by the mirror pack kernel) and the same bits as plain stores of 1 into four words of pinned host memory, which the
// host reads after any stream synchronisation without a copy.  Rare events only: call from ONE lane.
template <typename TS> MJB_DEV void raise_engine_flags(DataRef<TS> d, int fl) {
  if (d.flags) atomicOr(d.flags, fl);
  if (d.flags_pin) {
#pragma unroll
    for (int k = 0; k < 4; k++) if ((fl >> k) & 1) *(volatile int*)(d.flags_pin + k) = 1;
  }
}
// the tag no hand-over of this launch can carry (steps per launch < 2^20 - 2): written into an environment's clock words by the wave
// that gave up waiting for it, so that the waves holding the environment's LATER chunks stop at once instead of timing out one by one
MJB_DEV unsigned xfer_dead_tag(unsigned tagbase) { return tagbase | 0xFFFFFu; }
#endif

// steps [s_begin, s_end) of the launch's a.nstep.  tag_in != 0: the state comes from the hand-over buffer (written by the wave that
// ran the steps before s_begin).  s_end < a.nstep (ticket mode): it goes there; else to the state arrays, and this call ends the
// launch for the environment (counters, kinematic outputs, dumps).
template <typename T, typename TS, int G>
MJB_DEV void env_run(const DevModel<T> MJB_CONST* mp, const Lay MJB_CONST* lp, DataRef<TS> d, DebugRef<TS> dbg, ArgsRef a,
                     ObsRef obs, TS* obs_out, T* w, int* wi, int env, int lane, int s_begin, int s_end, unsigned tag_in,
                     unsigned long long* tlacc = nullptr) {
  Ctx<T> c(mp, lp, w, wi, lane);
  ModelRef<T> m = MJB_MODEL_OF(mp); LayRef L = *lp;
  MJB_SPEC_ASSUME(m) MJB_SPEC_ASSUME_LAY(L)
  const int nq = m.nq, nv = m.nv, nu = m.nu;
  double time;
  bool staged_in = false;
#ifndef MJB_HOST_EMU
#ifndef MJB_NO_XFER
  if constexpr (sizeof(T) == 4 && sizeof(TS) == 4) {
    if (tag_in != 0) {
      staged_in = true;
      const unsigned long long* x = d.xfer + (size_t)env * (size_t)(nq + 3 * nv + 2);
      unsigned tlo = 0, thi = 0;
      const unsigned dead = xfer_dead_tag(a.tagbase);
      const unsigned long long t_wait0 = __builtin_amdgcn_s_memrealtime();
      bool lost = false;                                         // group-uniform: this environment's earlier steps will never arrive
      for (;;) {
        bool ok = true;
        for (int i = lane; i < nq; i += G) { const unsigned long long v = xfer_ld(x + i); ok = ok && (unsigned)(v >> 32) == tag_in; w[L.qpos + i] = __uint_as_float((unsigned)v); }
        for (int i = lane; i < nv; i += G) {
          const unsigned long long v0 = xfer_ld(x + nq + i), v1 = xfer_ld(x + nq + nv + i), v2 = xfer_ld(x + nq + 2 * nv + i);
          ok = ok && (unsigned)(v0 >> 32) == tag_in && (unsigned)(v1 >> 32) == tag_in && (unsigned)(v2 >> 32) == tag_in;
          w[L.qvel + i] = __uint_as_float((unsigned)v0); w[L.qacc_ws + i] = __uint_as_float((unsigned)v1); w[L.qacc + i] = __uint_as_float((unsigned)v2);
        }
        {                                                          // the clock: every lane reads the same two words
          const unsigned long long v0 = xfer_ld(x + nq + 3 * nv), v1 = xfer_ld(x + nq + 3 * nv + 1);
          ok = ok && (unsigned)(v0 >> 32) == tag_in && (unsigned)(v1 >> 32) == tag_in;
          tlo = (unsigned)v0; thi = (unsigned)v1;
          lost = (unsigned)(v0 >> 32) == dead;                     // a wave before this one gave the environment up
        }
        if (__builtin_amdgcn_ballot_w64(!ok && !lost) == 0) { lost = lost || gany<G>(!ok); break; }   // every word of every live environment of this wave carries the tag
        // the steps before s_begin are held by a wave with a smaller ticket, which is running: wait - bounded by the wall clock.  A
        // wave that gives up flags the launch (bit 3 of the engine flags: the host's next synchronising call returns MJB_ERR_DEVICE),
        // marks the environment dead for the waves that hold its later chunks and does NOT step it: nothing computed from a torn
        // state is ever published, the environment's state arrays keep what they held when the launch started.
        if (__builtin_amdgcn_s_memrealtime() - t_wait0 > (unsigned long long)a.xfer_timeout) {
          lost = lost || gany<G>(!ok);
          if (lost && lane == 0) {
            raise_engine_flags<TS>(d, 8);
            xfer_st(d.xfer + (size_t)env * (size_t)(nq + 3 * nv + 2) + nq + 3 * nv, 0u, dead);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (lost) return;
      time = __longlong_as_double((long long)(((unsigned long long)thi << 32) | tlo));
    }
  }
#endif
#endif
  if (!staged_in) {
    for (int i = lane; i < nq; i += G) w[L.qpos + i] = (T)d.qpos[(size_t)env * nq + i];
    for (int i = lane; i < nv; i += G) {
      w[L.qvel + i] = (T)d.qvel[(size_t)env * nv + i];
      w[L.qacc_ws + i] = (T)d.qacc_warmstart[(size_t)env * nv + i];
      w[L.qacc + i] = (T)d.qacc[(size_t)env * nv + i];
    }
    time = d.time[env];
  }
  for (int i = lane; i < nu; i += G) w[L.ctrl + i] = a.ctrl_mode == CTRL_ZERO ? (T)0 : (T)d.ctrl[(size_t)env * nu + i];
  for (int i = lane; i < nv * nv; i += G) w[L.M + i] = 0;       // structural zeros of the mass matrix (crb_factor fills the rest)
#ifndef MJB_HOST_EMU
  if (a.mirror && a.mirror_mask) mirror_in<T, G>(m, L, w, a.mirror, a.mirror_mask, env, d.batch, lane, time);
#endif
  int badqpos = 0, badqvel = 0, badqacc = 0;
  gsync<G>();
#if defined(MJB_PROFILE) && !defined(MJB_HOST_EMU)
  c.pt = __builtin_amdgcn_s_memtime();
#endif
  if (a.mode != 0) { s_begin = 0; s_end = 1; }
  c.skip_dynamics = a.mode == 2;
#if defined(MJB_PHASE_REPEAT) && !defined(MJB_HOST_EMU)
  c.rep = a.mode == 0 ? a.repeat_phase : -1;
#endif
  const int nstage = (a.mode == 0 && m.integrator == INT_RK4) ? 4 : 1;
#ifndef MJB_HOST_EMU
  const unsigned hwslot = __builtin_amdgcn_s_getreg((3 << 11) | 4);      // HW_ID[3:0]: this wave's slot on its SIMD
#endif
  for (int s = s_begin; s < s_end; s++) {
#if !defined(MJB_HOST_EMU) && defined(MJB_LANE_LAUNDER)
    // Experiment (off; MJB_SPEC_FLAGS=-DMJB_LANE_LAUNDER): the lane index made opaque once per step, so that the lane-derived loop invariants
    // of the pipeline (`lane < n` predicates, 64-bit table addresses of the baked model) are recomputed where they are used instead of being
    // hoisted in front of the step loop and spilled there.  VGPR spills 77 -> 43, scratch instructions 132 -> 73, but 2.4 % SLOWER (41.8 vs
    // 42.8 M env-steps/s): reloading a hoisted value costs less issue than recomputing it.  Bitwise identical either way.
    MJB_OPAQUE1(lane);
    __builtin_assume(lane >= 0 && lane < G);
    c.lane = lane;
#endif
#ifndef MJB_HOST_EMU
    // Issue priority of the SIMD's co-resident waves: the hardware arbitrates VALU issue by priority, then AGE, so of two waves that
    // start together the older one runs ~10 % faster for its whole life and the launch waits for the younger (profiles/r02_wave_timeline.log).
    // Bit `fair_bit` of the 100 MHz wall clock, which all waves read alike, hands the priority back and forth between odd and even slots.
#ifndef MJB_NO_FAIR
    if (a.fair_bit > 0) {
      const unsigned ph = (unsigned)(__builtin_amdgcn_s_memrealtime() >> (unsigned)a.fair_bit);
      if ((ph + hwslot) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
#endif
#endif
    if (a.mode == 0) {
      if (group_bad<T, G>(w + L.qpos, nq, lane)) { badqpos++; reset_state<T, G>(c); time = 0; }
      if (group_bad<T, G>(w + L.qvel, nv, lane)) { badqvel++; reset_state<T, G>(c); time = 0; }
      if (a.ctrl_mode == CTRL_RANDOM) {
        for (int rp_ = MJB_REP_N(c, REP_CTRL); rp_ > 0; rp_--) random_ctrl<T, G>(m, w + L.ctrl, a.seed, a.env0 + (unsigned)env, a.step0 + (unsigned)s, (T)a.ctrl_scale, lane);
        gsync<G>();
      } else if (a.ctrl_mode == CTRL_FEEDBACK) feedback_ctrl<T, G>(c, a, a.env0 + (unsigned)env, a.step0 + (unsigned)s);
    }
    bool retried = false;
    for (int st = 0; st < nstage; st++) {
      forward<T, G>(c);                                   // the only call site of the forward pipeline
      if (a.mode != 0) break;
      if (st == 0 && !retried && group_bad<T, G>(w + L.qacc, nv, lane)) {
        badqacc++; reset_state<T, G>(c); time = 0; retried = true; st = -1;
        continue;
      }
      if (nstage == 4) rk4_stage<T, G>(c, st);
    }
    if (a.mode == 2) inverse_dynamics<T, TS, G>(c, d.qfrc_inverse + (size_t)env * nv, d.actuator_moment + (size_t)env * nu * nv);
    if (a.mode != 0) break;
    if (nstage == 1) euler<T, G>(c);
    MJB_STAMP(c, PH_INTEG);
#if defined(MJB_TIMELINE) && !defined(MJB_HOST_EMU)
    if (tlacc) *tlacc += ((unsigned long long)(unsigned)c.niter << 8) | ((unsigned long long)(unsigned)c.nefc << 24) | ((unsigned long long)(unsigned)c.ncon << 44);
#endif
    time += a.dt;
    if (a.obs_every > 0 && ((s + 1) % a.obs_every) == 0) {
      size_t slot = (size_t)((s + 1) / a.obs_every - 1);
      write_obs<T, TS, G>(c, obs, time, obs_out + (slot * (size_t)d.batch + (size_t)env) * (size_t)obs.dim);
    }
  }
#if defined(MJB_PROFILE) && !defined(MJB_HOST_EMU)
  if (lane == 0 && d.prof) for (int k = 0; k < PH_N; k++) atomicAdd(d.prof + k, c.pacc[k]);
#endif
  // the accumulating counters are almost always zero: device-scope atomics only when there is something to add
  if (lane == 0) {
    int* cn = d.counters + (size_t)env * CNT_N;
#ifndef MJB_HOST_EMU
    if (c.con_dropped) atomicAdd(cn + CNT_CON_DROPPED, c.con_dropped);
    if (c.efc_dropped) atomicAdd(cn + CNT_EFC_DROPPED, c.efc_dropped);
    if (badqpos) atomicAdd(cn + CNT_BADQPOS, badqpos);
    if (badqvel) atomicAdd(cn + CNT_BADQVEL, badqvel);
    if (badqacc) atomicAdd(cn + CNT_BADQACC, badqacc);
    const int fl = (c.con_dropped ? 1 : 0) | (c.efc_dropped ? 2 : 0) | ((badqpos | badqvel | badqacc) ? 4 : 0);
    if (fl) raise_engine_flags<TS>(d, fl);                     // rare: the host reads ONE word instead of the [batch, 8] counters
#else
    cn[CNT_CON_DROPPED] += c.con_dropped; cn[CNT_EFC_DROPPED] += c.efc_dropped;
    cn[CNT_BADQPOS] += badqpos; cn[CNT_BADQVEL] += badqvel; cn[CNT_BADQACC] += badqacc;
#endif
  }
#if !defined(MJB_HOST_EMU) && !defined(MJB_NO_XFER)
  if constexpr (sizeof(T) == 4 && sizeof(TS) == 4) {
    // the hand-over made at step s_end carries tag tagbase + s_end
    unsigned tag_out = (a.mode == 0 && a.chunk_steps > 0 && s_end < a.nstep) ? a.tagbase + (unsigned)s_end : 0u;
    if (tag_out != 0) {                                        // hand the environment to whoever draws its next chunk
      if (a.xfer_poison_env != 0 && env + 1 == a.xfer_poison_env) tag_out ^= 0x80000u;   // test hook: this hand-over never validates
      unsigned long long* x = d.xfer + (size_t)env * (size_t)(nq + 3 * nv + 2);
      for (int i = lane; i < nq; i += G) xfer_st(x + i, __float_as_uint(w[L.qpos + i]), tag_out);
      for (int i = lane; i < nv; i += G) {
        xfer_st(x + nq + i, __float_as_uint(w[L.qvel + i]), tag_out);
        xfer_st(x + nq + nv + i, __float_as_uint(w[L.qacc_ws + i]), tag_out);
        xfer_st(x + nq + 2 * nv + i, __float_as_uint(w[L.qacc + i]), tag_out);
      }
      if (lane == 0) {
        const unsigned long long tb = (unsigned long long)__double_as_longlong(time);
        xfer_st(x + nq + 3 * nv, (unsigned)tb, tag_out); xfer_st(x + nq + 3 * nv + 1, (unsigned)(tb >> 32), tag_out);
      }
      return;
    }
  }
#endif
  // store state
  for (int i = lane; i < nq; i += G) d.qpos[(size_t)env * nq + i] = (TS)w[L.qpos + i];
  for (int i = lane; i < nv; i += G) {
    d.qvel[(size_t)env * nv + i] = (TS)w[L.qvel + i];
    d.qacc[(size_t)env * nv + i] = (TS)w[L.qacc + i];
    d.qacc_warmstart[(size_t)env * nv + i] = (TS)w[L.qacc_ws + i];
  }
  if (a.ctrl_mode != CTRL_KEEP || (a.mirror && (a.mirror_mask & 4))) for (int i = lane; i < nu; i += G) d.ctrl[(size_t)env * nu + i] = (TS)w[L.ctrl + i];
  if (lane == 0) {
    d.time[env] = time;
    int* cn = d.counters + (size_t)env * CNT_N;
    cn[CNT_NCON] = c.ncon; cn[CNT_NEFC] = c.nefc; cn[CNT_NITER] = c.niter;
  }
  if (a.write_kin) {
    for (int i = lane; i < 3 * m.nbody; i += G) {
      d.xpos[(size_t)env * 3 * m.nbody + i] = (TS)w[L.xpos + i];
      d.xipos[(size_t)env * 3 * m.nbody + i] = (TS)w[L.xipos + i];
      d.subtree_com[(size_t)env * 3 * m.nbody + i] = (TS)w[L.subtree_com + i];
    }
    for (int i = lane; i < 4 * m.nbody; i += G) d.xquat[(size_t)env * 4 * m.nbody + i] = (TS)w[L.xquat + i];
    for (int i = lane; i < 3 * m.nsite; i += G) d.site_xpos[(size_t)env * 3 * m.nsite + i] = (TS)w[L.site_xpos + i];
    for (int i = lane; i < 3 * m.ngeom; i += G) d.geom_xpos[(size_t)env * 3 * m.ngeom + i] = (TS)w[L.geom_xpos + i];
    for (int i = lane; i < m.nsensordata; i += G) d.sensordata[(size_t)env * m.nsensordata + i] = (TS)w[L.sens + i];
  }
  // optional per-phase dumps (forward mode, parity tests)
  if (dbg.qM) for (int i = lane; i < nv * nv; i += G) dbg.qM[(size_t)env * nv * nv + i] = (TS)w[L.M + i];
  if (dbg.qfrc_bias) for (int i = lane; i < nv; i += G) {
    dbg.qfrc_bias[(size_t)env * nv + i] = (TS)w[L.qfrc_bias + i];
    dbg.qfrc_passive[(size_t)env * nv + i] = (TS)w[L.qfrc_passive + i];
    dbg.qfrc_actuator[(size_t)env * nv + i] = (TS)w[L.qfrc_actuator + i];
    dbg.qacc_smooth[(size_t)env * nv + i] = (TS)w[L.qacc_smooth + i];
    dbg.qfrc_constraint[(size_t)env * nv + i] = (TS)w[L.qfrc_constraint + i];
  }
  if (dbg.efc_J) {
    size_t ne = (size_t)m.nefc_max;
    for (int i = lane; i < c.nefc * nv; i += G) dbg.efc_J[(size_t)env * ne * nv + i] = (TS)w[L.efc_J + i];
    for (int i = lane; i < c.nefc; i += G) {
      dbg.efc_aref[(size_t)env * ne + i] = (TS)w[L.efc_aref + i];
      dbg.efc_D[(size_t)env * ne + i] = (TS)w[L.efc_D + i];
      dbg.efc_pos[(size_t)env * ne + i] = (TS)w[L.efc_pos + i];
      dbg.efc_force[(size_t)env * ne + i] = (TS)w[L.efc_force + i];
      dbg.efc_type[(size_t)env * ne + i] = wi[L.i_efc_type + i] & 0xff;
    }
  }
  if (dbg.con) for (int i = lane; i < c.ncon * CON_STRIDE; i += G) dbg.con[(size_t)env * m.ncon_max * CON_STRIDE + i] = (TS)w[L.con + i];
  if (dbg.cdof) {
    for (int i = lane; i < 6 * nv; i += G) dbg.cdof[(size_t)env * 6 * nv + i] = (TS)w[L.cdof + i];
    for (int i = lane; i < 10 * m.nbody; i += G) dbg.cinert[(size_t)env * 10 * m.nbody + i] = (TS)w[L.cinert + i];
    for (int i = lane; i < 6 * m.nbody; i += G) dbg.cvel[(size_t)env * 6 * m.nbody + i] = (TS)w[L.cvel + i];
  }
#ifndef MJB_HOST_EMU
  // LAST: the completion word a polling host returns on (mirror_out's release store) is published behind every global store of this
  // environment above, so a consumer on another stream that reads the device arrays after the polled return sees them complete
  if (a.mirror) mirror_out<T, G>(m, L, w, a.mirror, env, d.batch, lane, time, (c.con_dropped ? 1 : 0) | (c.efc_dropped ? 2 : 0) | ((badqpos | badqvel | badqacc) ? 4 : 0), a.mirror_seq);
#endif
}


// ---------------------------------------------------------------------------
// Two waves per environment (k_step2, mjb_kernels.hpp): for batches that leave most of the chip idle (no more environments than
// half the resident slots of k_step) one environment is stepped by a 128-thread workgroup whose two wavefronts run the phases of a
// step that do not depend on each other side by side, on a FLAT LDS layout (make_layout(..., flat): temporaries of different
// overlay groups are live at the same time):
//     wave 0: [bad-state checks, ctrl, kinematics] | com_pos, crb       | bias / passive forces | actuation, M^-1, Newton solver, sensors |        | obs, I/O
//     wave 1:                 (waits)              | collision          | constraint rows       | -(M + h D)^-1 in registers             | Euler  |
// separated by s_barrier.  The phases are the very functions of the one-wave kernel in an order that respects their data
// dependences, so the results are bitwise those of k_step (tests: test_two_wave_kernel_is_bitwise_identical).  fp32, nv <= 32,
// Euler integrator, stepping mode only; everything else stays on k_step.
// ---------------------------------------------------------------------------
#ifndef MJB_HOST_EMU
template <typename T, typename TS>
MJB_DEV void env_run2(const DevModel<T> MJB_CONST* mp, const Lay MJB_CONST* lp, DataRef<TS> d, ArgsRef a, ObsRef obs, TS* obs_out, T* w, int* wi, int env, int lane, int wv) {
  constexpr int G = 64;
  Ctx<T> c(mp, lp, w, wi, lane);
  ModelRef<T> m = MJB_MODEL_OF(mp); LayRef L = *lp;
  MJB_SPEC_ASSUME(m) MJB_SPEC_ASSUME_LAY(L)
  const int nq = m.nq, nv = m.nv, nu = m.nu;
  int* mail = wi + L.i_mail;                 // [0] ncon [1] nefc [2] contacts dropped (total) [3] rows dropped (total) [4] bad acceleration in this pass
  double time = 0;
  int badqpos = 0, badqvel = 0, badqacc = 0;
  if (wv == 0) {
    for (int i = lane; i < nq; i += G) w[L.qpos + i] = (T)d.qpos[(size_t)env * nq + i];
    for (int i = lane; i < nv; i += G) {
      w[L.qvel + i] = (T)d.qvel[(size_t)env * nv + i];
      w[L.qacc_ws + i] = (T)d.qacc_warmstart[(size_t)env * nv + i];
      w[L.qacc + i] = (T)d.qacc[(size_t)env * nv + i];
    }
    for (int i = lane; i < nu; i += G) w[L.ctrl + i] = a.ctrl_mode == CTRL_ZERO ? (T)0 : (T)d.ctrl[(size_t)env * nu + i];
    for (int i = lane; i < nv * nv; i += G) w[L.M + i] = 0;     // structural zeros of the mass matrix
    if (lane < 8) mail[lane] = 0;
    time = d.time[env];
    if (a.mirror && a.mirror_mask) mirror_in<T, G>(m, L, w, a.mirror, a.mirror_mask, env, d.batch, lane, time);
  }
  __syncthreads();
  for (int s = 0; s < a.nstep; s++) {
    if (wv == 0) {
      if (group_bad<T, G>(w + L.qpos, nq, lane)) { badqpos++; reset_state<T, G>(c); time = 0; }
      if (group_bad<T, G>(w + L.qvel, nv, lane)) { badqvel++; reset_state<T, G>(c); time = 0; }
      if (a.ctrl_mode == CTRL_RANDOM) {
        random_ctrl<T, G>(m, w + L.ctrl, a.seed, a.env0 + (unsigned)env, a.step0 + (unsigned)s, (T)a.ctrl_scale, lane);
        gsync<G>();
      } else if (a.ctrl_mode == CTRL_FEEDBACK) feedback_ctrl<T, G>(c, a, a.env0 + (unsigned)env, a.step0 + (unsigned)s);
    }
    mjb_f16v inv;
    for (int pass = 0; pass < 2; pass++) {                      // pass 1 only after a bad-acceleration reset (like k_step's retry)
      // Round 3 schedule (the M^-1 solve leaves wave 0's critical path; wave 1, idle through most of the solver before, carries the
      // velocity stage and both sweep inverses):
      //     wave 0:  kinematics | com_pos        | crb -> [M ready] -> constraint rows | actuator forces |              | Newton solver, sensors |        |
      //     wave 1:   (waits)   | collision      | bias / passive forces -> -M^-1      |   (waits)       | M^-1 f       | -(M + h D)^-1          | Euler  |
      // "M ready" is one LDS word wave 0 sets after crb (wave 1 reaches its sweep later than that, and polls it to be sure).
      const int gen = 2 * s + pass + 1;                         // value of the "M ready" word for this pass
      if (wv == 0) kinematics<T, G>(c);
      __syncthreads();
      if (wv == 0) com_pos<T, G>(c);
      else {
        collision<T, G>(c);
        if (lane == 0) { mail[0] = c.ncon; mail[2] = c.con_dropped; }
      }
      __syncthreads();
      mjb_f16v invm;
      if (wv == 0) {
        crb_factor<T, G>(c);
        gsync<G>();
        if (lane == 0) __hip_atomic_store(mail + 5, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        c.ncon = mail[0];
        make_constraint<T, G>(c);
        if (lane == 0) { mail[1] = c.nefc; mail[3] = c.efc_dropped; }
      } else {
        vel_bias_passive<T, G>(c);
        while (__hip_atomic_load(mail + 5, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != gen) __builtin_amdgcn_s_sleep(1);
        invm = mfma_sweep_invert32<ModelRef<T>>(m, w + L.M, w + L.efc_J, w + L.efc_jv, 0, 0, nv, lane);
      }
      __syncthreads();
      if (wv == 0) actuation_forces<T, G>(c);
      __syncthreads();
      if (wv == 1) mfma_sweep_apply32(invm, w + L.tmp, nv, lane, w + L.qacc_smooth);
      __syncthreads();
      if (wv == 0) {
        c.nefc = mail[1];
        solve_constraints<T, G>(c);
        if (m.nsensor > 0) { sensors<T, T, G>(c, w + L.sens); gsync<G>(); }
        const bool bad = pass == 0 && group_bad<T, G>(w + L.qacc, nv, lane);
        if (lane == 0) mail[4] = bad ? 1 : 0;
      } else if (m.has_damping) {
        inv = mfma_sweep_invert32<ModelRef<T>>(m, w + L.M, w + L.efc_J, w + L.efc_jv, 0, 2, nv, lane);
      }
      __syncthreads();
      if (mail[4] == 0) break;
      if (wv == 0) { badqacc++; reset_state<T, G>(c); time = 0; }
      __syncthreads();
    }
    if (wv == 1) euler<T, G>(c, m.has_damping ? &inv : nullptr);
    __syncthreads();
    if (wv == 0) {
      time += a.dt;
      if (a.obs_every > 0 && ((s + 1) % a.obs_every) == 0) {
        size_t slot = (size_t)((s + 1) / a.obs_every - 1);
        write_obs<T, TS, G>(c, obs, time, obs_out + (slot * (size_t)d.batch + (size_t)env) * (size_t)obs.dim);
      }
    }
  }
  if (wv != 0) return;
  if (lane == 0) {
    int* cn = d.counters + (size_t)env * CNT_N;
    const int cdrop = mail[2], edrop = mail[3];
    if (cdrop) atomicAdd(cn + CNT_CON_DROPPED, cdrop);
    if (edrop) atomicAdd(cn + CNT_EFC_DROPPED, edrop);
    if (badqpos) atomicAdd(cn + CNT_BADQPOS, badqpos);
    if (badqvel) atomicAdd(cn + CNT_BADQVEL, badqvel);
    if (badqacc) atomicAdd(cn + CNT_BADQACC, badqacc);
    const int fl = (cdrop ? 1 : 0) | (edrop ? 2 : 0) | ((badqpos | badqvel | badqacc) ? 4 : 0);
    if (fl) raise_engine_flags<TS>(d, fl);
    d.time[env] = time;
    cn[CNT_NCON] = c.ncon; cn[CNT_NEFC] = c.nefc; cn[CNT_NITER] = c.niter;
  }
  for (int i = lane; i < nq; i += G) d.qpos[(size_t)env * nq + i] = (TS)w[L.qpos + i];
  for (int i = lane; i < nv; i += G) {
    d.qvel[(size_t)env * nv + i] = (TS)w[L.qvel + i];
    d.qacc[(size_t)env * nv + i] = (TS)w[L.qacc + i];
    d.qacc_warmstart[(size_t)env * nv + i] = (TS)w[L.qacc_ws + i];
  }
  if (a.ctrl_mode != CTRL_KEEP || (a.mirror && (a.mirror_mask & 4))) for (int i = lane; i < nu; i += G) d.ctrl[(size_t)env * nu + i] = (TS)w[L.ctrl + i];
  if (a.write_kin) {
    for (int i = lane; i < 3 * m.nbody; i += G) {
      d.xpos[(size_t)env * 3 * m.nbody + i] = (TS)w[L.xpos + i];
      d.xipos[(size_t)env * 3 * m.nbody + i] = (TS)w[L.xipos + i];
      d.subtree_com[(size_t)env * 3 * m.nbody + i] = (TS)w[L.subtree_com + i];
    }
    for (int i = lane; i < 4 * m.nbody; i += G) d.xquat[(size_t)env * 4 * m.nbody + i] = (TS)w[L.xquat + i];
    for (int i = lane; i < 3 * m.nsite; i += G) d.site_xpos[(size_t)env * 3 * m.nsite + i] = (TS)w[L.site_xpos + i];
    for (int i = lane; i < 3 * m.ngeom; i += G) d.geom_xpos[(size_t)env * 3 * m.ngeom + i] = (TS)w[L.geom_xpos + i];
    for (int i = lane; i < m.nsensordata; i += G) d.sensordata[(size_t)env * m.nsensordata + i] = (TS)w[L.sens + i];
  }
  // LAST (see env_run): the polled completion word goes out behind every global store of this environment
  if (a.mirror) mirror_out<T, G>(m, L, w, a.mirror, env, d.batch, lane, time, (mail[2] ? 1 : 0) | (mail[3] ? 2 : 0) | ((badqpos | badqvel | badqacc) ? 4 : 0), a.mirror_seq);
}
#endif

}  // namespace mjb
