// mjb_hostemu.hpp — TEST-ONLY host emulation of the group primitives used by
// mjb_device.hpp: one std::thread per lane, pthread barriers at gsync(), and
// shared scratch for the collectives.  Lets the CPU test-suite execute the very
// kernel source (lane-strided loops, scans, shuffles, syncs) without a GPU.
// Never part of the product library: the product fails loudly without a HIP device.
#pragma once
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
template <typename T, int G> static inline T gshfl(T v, int src) {
  emu::Group* g = emu::tl_group;
  g->fbuf[emu::tl_lane] = (double)v;
  emu::sync();
  double r = g->fbuf[src];
  emu::sync();
  return (T)r;
}
}  // namespace mjb
