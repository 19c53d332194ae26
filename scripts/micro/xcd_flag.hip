// Micro-test: does a flag + payload written by one workgroup become visible to spinning workgroups on OTHER XCDs within a kernel,
// and through which access flavours?  Every spin is bounded, so the kernel always ends.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_flag xcd_flag.hip && ./xcd_flag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Rec { unsigned saw, spins, xcc, payload_ok; };

template <int METHOD>
__global__ __launch_bounds__(64) void k(unsigned* flag, unsigned* payload, Rec* rec, unsigned* ticket, int npay) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
  // who produces: the workgroup that draws ticket 0 (so that it certainly is resident)
  unsigned t = 0;
  if (threadIdx.x == 0) t = atomicAdd(ticket, 1u);
  t = __builtin_amdgcn_readfirstlane(t);
  if (t == 0) {
    for (int i = 0; i < 40; i++) __builtin_amdgcn_s_sleep(127);          // ~150 us: let the consumers poll (and cache) the old value first
    if (METHOD == 0) {            // relaxed agent-scope atomics only + wave-level wait
      for (int i = threadIdx.x; i < npay; i += 64) __hip_atomic_store(payload + i, 1000u + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);                                       // vmcnt(0) expcnt(0) lgkmcnt(0)
      if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (METHOD == 1) {     // plain payload, __threadfence, atomic flag
      for (int i = threadIdx.x; i < npay; i += 64) payload[i] = 1000u + i;
      __threadfence();
      if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (METHOD == 2) {     // RMW on both sides
      for (int i = threadIdx.x; i < npay; i += 64) atomicExch(payload + i, 1000u + i);
      __builtin_amdgcn_s_waitcnt(0);
      if (threadIdx.x == 0) atomicExch(flag, 1u);
    } else {                      // sc1 payload, agent release fence, release store
      for (int i = threadIdx.x; i < npay; i += 64) __hip_atomic_store(payload + i, 1000u + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) { rec[t].saw = 2; rec[t].xcc = xcc; rec[t].spins = 0; rec[t].payload_ok = 1; }
    return;
  }
  unsigned spins = 0, saw = 0;
  for (; spins < 20000; spins++) {                                          // bounded: ~20000 x (sleep 8 = 512 cycles + load) ~ 10 ms
    unsigned f;
    if (METHOD == 2) f = atomicAdd(flag, 0u);
    else if (METHOD == 3) f = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    else f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f) { saw = 1; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  if (METHOD == 1) __threadfence();
  unsigned ok = 1;
  for (int i = threadIdx.x; i < npay; i += 64) {
    unsigned v;
    if (METHOD == 1) v = payload[i];
    else if (METHOD == 2) v = atomicAdd(payload + i, 0u);
    else v = __hip_atomic_load(payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != 1000u + i) ok = 0;
  }
  ok = __ballot(ok == 0) == 0 ? 1u : 0u;
  if (threadIdx.x == 0) { rec[t].saw = saw; rec[t].spins = spins; rec[t].xcc = xcc; rec[t].payload_ok = ok; }
}

template <int METHOD> int run(const char* name) {
  const int nblk = 2048, npay = 131;
  unsigned *flag, *payload, *ticket; Rec* rec;
  CHECK(hipMalloc(&flag, 4)); CHECK(hipMalloc(&payload, 4 * npay)); CHECK(hipMalloc(&ticket, 4)); CHECK(hipMalloc(&rec, sizeof(Rec) * nblk));
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipMemset(flag, 0, 4)); CHECK(hipMemset(payload, 0, 4 * npay)); CHECK(hipMemset(ticket, 0, 4)); CHECK(hipMemset(rec, 0, sizeof(Rec) * nblk));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<METHOD>, dim3(nblk), dim3(64), 16 * 1024, 0, flag, payload, rec, ticket, npay);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Rec> h(nblk); CHECK(hipMemcpy(h.data(), rec, sizeof(Rec) * nblk, hipMemcpyDeviceToHost));
    unsigned pxcc = h[0].xcc; int saw_same = 0, n_same = 0, saw_other = 0, n_other = 0, bad_payload = 0; double sp_same = 0, sp_other = 0;
    for (int i = 1; i < nblk; i++) {
      bool same = h[i].xcc == pxcc;
      (same ? n_same : n_other)++;
      if (h[i].saw) { (same ? saw_same : saw_other)++; (same ? sp_same : sp_other) += h[i].spins; if (!h[i].payload_ok) bad_payload++; }
    }
    printf("%-52s rep %d: %.3f ms; producer on XCD %u; consumers same XCD saw flag %d/%d (mean spins %.0f), other XCDs %d/%d (mean spins %.0f); stale payload after flag: %d\n",
           name, rep, ms, pxcc, saw_same, n_same, saw_same ? sp_same / saw_same : 0.0, saw_other, n_other, saw_other ? sp_other / saw_other : 0.0, bad_payload);
  }
  return 0;
}
int main() {
  if (run<0>("0: relaxed agent atomics (sc1) + s_waitcnt")) return 1;
  if (run<1>("1: plain payload + __threadfence + atomic flag")) return 1;
  if (run<2>("2: RMW atomics both sides")) return 1;
  if (run<3>("3: sc1 payload + release store / acquire load")) return 1;
  return 0;
}
