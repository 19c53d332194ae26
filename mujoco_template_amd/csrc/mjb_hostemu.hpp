// mjb_hostemu.hpp — TEST-ONLY host emulation of the group primitives used by
// mjb_device.hpp: one std::thread per lane, pthread barriers at gsync(), and
// shared scratch for the collectives.  Lets the CPU test-suite execute the very
// kernel source (lane-strided loops, scans, shuffles, syncs) without a GPU.
// Never part of the product library: the product fails loudly without a HIP device.
#pragma once
#include <cstring>
#include <pthread.h>

#include <cmath>
#include <vector>

namespace mjb {
namespace emu {
struct Group {
  int G;
  pthread_barrier_t bar;
  std::vector<double> fbuf;
  std::vector<long long> ibuf;
  explicit Group(int g) : G(g), fbuf(g), ibuf(g) { pthread_barrier_init(&bar, nullptr, (unsigned)g); }
  ~Group() { pthread_barrier_destroy(&bar); }
};
inline thread_local Group* tl_group = nullptr;
inline thread_local int tl_lane = 0;
inline void sync() { pthread_barrier_wait(&tl_group->bar); }
}  // namespace emu

template <int G> static inline void gsync() { emu::sync(); }
template <typename T, int G> static inline T gsum(T v) {
  emu::Group* g = emu::tl_group;
  g->fbuf[emu::tl_lane] = (double)v;
  emu::sync();
  double s = 0;
  for (int i = 0; i < g->G; i++) s += g->fbuf[i];
  emu::sync();
  return (T)s;
}
template <int G> static inline int gsumi(int v) {
  emu::Group* g = emu::tl_group;
  g->ibuf[emu::tl_lane] = v;
  emu::sync();
  long long s = 0;
  for (int i = 0; i < g->G; i++) s += g->ibuf[i];
  emu::sync();
  return (int)s;
}
template <int G> static inline int gmaxi(int v) {
  emu::Group* g = emu::tl_group;
  g->ibuf[emu::tl_lane] = v;
  emu::sync();
  long long s = v;
  for (int i = 0; i < g->G; i++) if (g->ibuf[i] > s) s = g->ibuf[i];
  emu::sync();
  return (int)s;
}
template <int G> static inline bool gany(bool v) { return gsumi<G>((int)v) != 0; }
template <int G> static inline int gscan_excl(int v, int lane, int& total) {
  emu::Group* g = emu::tl_group;
  g->ibuf[lane] = v;
  emu::sync();
  long long s = 0, t = 0;
  for (int i = 0; i < g->G; i++) { if (i < lane) s += g->ibuf[i]; t += g->ibuf[i]; }
  emu::sync();
  total = (int)t;
  return (int)s;
}
template <int G> static inline int gscan_small(int v, int lane, int& total) { return gscan_excl<G>(v, lane, total); }
// ---- wave-level intrinsics of the fp32 hot path (one wavefront per environment: readlane, the 32-lane half swap, ballot and the
// 32x32x2 fp32 MFMA) so that the CPU suite executes the MFMA Cholesky / sweep-inverse code itself, not a substitute path.  Every
// call is a collective of all G lane-threads; values go through double-buffered scratch so that ONE barrier per call suffices (a
// lane cannot be two collectives ahead of another).  Not bit-exact with the hardware where that uses approximations (rsq, rcp).
namespace emu {
struct Wave {
  double f[2][64];
  unsigned long long u[2][64];
  float a[2][64], b[2][64];
  int phase[64];
};
inline Wave g_wave_storage;
inline thread_local Wave* tl_wave = &g_wave_storage;
inline int next_phase() { Wave* w = tl_wave; int p = w->phase[tl_lane] & 1; w->phase[tl_lane]++; return p; }
}  // namespace emu
template <typename T> static inline T rdlane_f(T v, int l) {
  emu::Wave* w = emu::tl_wave; const int p = emu::next_phase();
  w->f[p][emu::tl_lane] = (double)v;
  emu::sync();
  return (T)w->f[p][l];
}
static inline int rdlane_i(int v, int l) { return (int)rdlane_f<double>((double)v, l); }
static inline float half_bcast(float v, int half) { return rdlane_f<float>(v, (emu::tl_lane & 31) + 32 * half); }
static inline unsigned __float_as_uint(float x) { unsigned u; std::memcpy(&u, &x, 4); return u; }
static inline float __uint_as_float(unsigned u) { float x; std::memcpy(&x, &u, 4); return x; }
static inline void half_swap(float a, float b, float& ao, float& bo) {           // v_permlane32_swap: a' = [a.lo | b.lo], b' = [a.hi | b.hi]
  emu::Wave* w = emu::tl_wave; const int p = emu::next_phase();
  w->a[p][emu::tl_lane] = a; w->b[p][emu::tl_lane] = b;
  emu::sync();
  const int l = emu::tl_lane, c = l & 31;
  ao = l < 32 ? a : w->b[p][c];
  bo = l < 32 ? w->a[p][c + 32] : b;
}
static inline void half_bcast2(float v, float& lo, float& hi) {
  emu::Wave* w = emu::tl_wave; const int p = emu::next_phase();
  w->f[p][emu::tl_lane] = (double)v;
  emu::sync();
  const int c = emu::tl_lane & 31;
  lo = (float)w->f[p][c]; hi = (float)w->f[p][c + 32];
}
template <typename T> static inline T half_sum(T v) {
  emu::Wave* w = emu::tl_wave; const int p = emu::next_phase();
  w->f[p][emu::tl_lane] = (double)v;
  emu::sync();
  const int c = emu::tl_lane & 31;
  return (T)w->f[p][c] + (T)w->f[p][c + 32];
}
static inline unsigned long long emu_ballot(bool pred) {
  emu::Wave* w = emu::tl_wave; const int p = emu::next_phase();
  w->u[p][emu::tl_lane] = pred ? 1ull : 0ull;
  emu::sync();
  unsigned long long m = 0;
  for (int i = 0; i < emu::tl_group->G; i++) m |= w->u[p][i] << i;
  return m;
}
struct mjb_f16v { float v[16]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
// D = A (32 x 2) B (2 x 32) + C in the accumulator layout of v_mfma_f32_32x32x2_f32: lane l gives A[l % 32][l / 32] and B[l / 32][l % 32],
// and holds D[8 q + 4 (l / 32) + t][l % 32] in register 4 q + t
static inline mjb_f16v emu_mfma(float a, float b, mjb_f16v acc) {
  emu::Wave* w = emu::tl_wave; const int p = emu::next_phase();
  w->a[p][emu::tl_lane] = a; w->b[p][emu::tl_lane] = b;
  emu::sync();
  const int h = emu::tl_lane >> 5, c = emu::tl_lane & 31;
  for (int r = 0; r < 16; r++) {
    const int i = 8 * (r >> 2) + 4 * h + (r & 3);
    acc.v[r] = fmaf(w->a[p][i + 32], w->b[p][c + 32], fmaf(w->a[p][i], w->b[p][c], acc.v[r]));
  }
  return acc;
}

template <typename T, int G> static inline T gshfl(T v, int src) {
  emu::Group* g = emu::tl_group;
  g->fbuf[emu::tl_lane] = (double)v;
  emu::sync();
  double r = g->fbuf[src];
  emu::sync();
  return (T)r;
}
}  // namespace mjb
