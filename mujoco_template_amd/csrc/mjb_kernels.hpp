// mjb_kernels.hpp — __global__ entry points around mjb_device.hpp and their launchers.
// One workgroup = one 64-lane wavefront = 64/G environments; the LDS slice of each
// environment is carved from dynamic shared memory.  Launchers are instantiated in
// inst_*.hip (one translation unit per precision so they compile in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include "mjb_device.hpp"

namespace mjb {

#ifndef MJB_WPS
#define MJB_WPS 2        // waves per SIMD the register budget of k_step is sized for (LDS fixes the real occupancy)
#endif
// Two ways to map work to workgroups (StepArgs::chunk_steps, chosen on the host in launch()):
//  * static (chunk_steps == 0): workgroup b runs ALL steps of environment block b.  Right when every block is resident at once.
//  * tickets (chunk_steps > 0, fp32 only): with more blocks than the chip holds, the static map runs in "rounds" and every round
//    waits for its slowest environment (a Newton iteration costs ~5.5 us and their number per step varies 0..8: over 20 steps the wave
//    durations spread 0.5..1.35 ms around 1.0, profiles/r02_wave_timeline.log).  Here only the resident workgroups are launched; they
//    stay for the whole launch and draw tickets from one atomic counter: ticket t = chunk (t / nblk) of block (t % nblk); the chunks
//    are chunk_steps consecutive steps, then ever shorter ones towards the end of the launch (chunk_plan, mjb_types.hpp).  Between its chunks an environment travels through the tagged hand-over buffer (env_run), which
//    is also what orders the chunks: a wave that draws chunk k finds the words tagged k or re-reads until it does.  Tickets are
//    handed out in order, so the wave it waits for holds a SMALLER ticket and never waits for this one: no cycle.
//
// Register pressure: the kernel sits at 256 VGPRs / ~100 SGPRs inside env_run, so NOTHING of the loop around it may stay live across
// it (a first version kept the launch arguments and the loop's invariants in registers: +130 SGPR and +43 VGPR spills, -3.7 %).  The
// body therefore reads the launch arguments through the kernarg segment pointer, laundered at the top of every iteration: each
// field is a scalar load where it is used, and an iteration carries nothing over but that pointer.
template <typename T, typename TS, int G>
MJB_DEV void k_step_body() {
  extern __shared__ __align__(16) char smem[];
  typedef StepKernArgs<T, TS> KA;
  const KA MJB_CONST* kp = (const KA MJB_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
#if defined(MJB_TIMELINE)
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();     // diagnostic: when did this WORKGROUP start / end, where, how much did it do
  unsigned long long tlacc = 0, tlt = 0;
#define MJB_TLACC &tlacc
#else
#define MJB_TLACC nullptr
#endif
  // ONE call site of env_run (the forward pipeline is inlined once): the static map is the loop below with a single pass
  for (;;) {
    asm volatile("" : "+s"(kp));
    ArgsRef a = kp->a;
    DataRef<TS> d = kp->d;
    const int lane = threadIdx.x & (G - 1), sub = threadIdx.x / G;
    const DevModel<T> MJB_CONST* mp = (const DevModel<T> MJB_CONST*)kp->mg;
    const Lay MJB_CONST* lp = (const Lay MJB_CONST*)kp->lg;
    LayRef L = *lp;
    MJB_SPEC_ASSUME_LAY(L)
    T* w = (T*)(smem + (size_t)sub * L.bytes);
    int* wi = (int*)(w + L.nT);
#ifdef MJB_NO_TICKETS                      // experiment: what does the ticket loop cost the static map?
    const bool tickets = false;
#else
    const bool tickets = a.chunk_steps > 0 && a.mode == 0;
#endif
    unsigned blk = blockIdx.x;
    int s0 = 0, s1 = a.nstep;
    unsigned tag_in = 0;
    if (tickets) {
      const unsigned nblk = (unsigned)a.nblk, nchunk = (unsigned)a.nchunk;
      unsigned t = 0;
      if (threadIdx.x == 0) t = atomicAdd(d.sched, 1u);
      t = (unsigned)__builtin_amdgcn_readfirstlane((int)t) - a.ticket_base;
      if (t >= nblk * nchunk) break;
      const unsigned k = t / nblk;
      blk = t - k * nblk;
      chunk_plan(a.nstep, a.chunk_steps, a.nuniform, (int)k, s0, s1);
      tag_in = s0 > 0 ? a.tagbase + (unsigned)s0 : 0u;         // the hand-over made at step s0 carries tag tagbase + s0
    }
    const int env = (int)blk * (64 / G) + sub;
    if (env < d.batch) env_run<T, TS, G>(mp, lp, d, kp->dbg, a, kp->obs, kp->obs_out, w, wi, env, lane, s0, s1, tag_in, MJB_TLACC);
    asm volatile("" : "+s"(kp));                              // (not even the map's flag is carried across env_run)
#ifdef MJB_NO_TICKETS
    break;
#else
    if (!(kp->a.chunk_steps > 0 && kp->a.mode == 0)) break;
#endif
#if defined(MJB_TIMELINE)
    tlt++;
#endif
  }
#if defined(MJB_TIMELINE)
  asm volatile("" : "+s"(kp));
  if (threadIdx.x == 0 && kp->d.prof) {
    unsigned long long* tl = kp->d.prof + PH_N + 4 * (size_t)blockIdx.x;
    tl[0] = tl0; tl[1] = __builtin_amdgcn_s_memrealtime();
    tl[2] = __builtin_amdgcn_s_getreg((15 << 11) | 4) | (tlt << 16);   // HW_ID[15:0]: wave slot, SIMD, pipe, CU, SH, SE; tickets served
    tl[3] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) | tlacc;   // XCC_ID[3:0], sums of Newton iterations / rows / contacts
  }
#endif
#undef MJB_TLACC
}
// (the parameters are read through the kernarg segment pointer, see k_step_body; StepKernArgs mirrors this list)
template <typename T, typename TS, int G>
__global__ __launch_bounds__(64, MJB_WPS) void k_step(const DevModel<T>* mg, const Lay* lg, DevData<TS> d, DevDebug<TS> dbg, StepArgs a, ObsSpecDev obs, TS* obs_out) {
  k_step_body<T, TS, G>();
}
// Two waves per environment (env_run2, mjb_device.hpp): 128-thread workgroups, one environment each, flat LDS layout (lg).
#ifndef MJB_WPS2
#define MJB_WPS2 2       // waves per SIMD the register budget of k_step2 is sized for: four workgroups per CU (as fast as the 512-VGPR build at <= 512 environments)
#endif
template <typename T, typename TS>
MJB_DEV void k_step2_body() {
  extern __shared__ __align__(16) char smem[];
  typedef StepKernArgs<T, TS> KA;
  const KA MJB_CONST* kp = (const KA MJB_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const DevModel<T> MJB_CONST* mp = (const DevModel<T> MJB_CONST*)kp->mg;
  const Lay MJB_CONST* lp = (const Lay MJB_CONST*)kp->lg;
  LayRef L = *lp;
  MJB_SPEC_ASSUME_LAY(L)
  T* w = (T*)smem;
  int* wi = (int*)(w + L.nT);
  const int env = blockIdx.x;
  if (env >= kp->d.batch) return;
  env_run2<T, TS>(mp, lp, kp->d, kp->a, kp->obs, kp->obs_out, w, wi, env, lane, wv);
}
template <typename T, typename TS>
__global__ __launch_bounds__(128, MJB_WPS2) void k_step2(const DevModel<T>* mg, const Lay* lg, DevData<TS> d, DevDebug<TS> dbg, StepArgs a, ObsSpecDev obs, TS* obs_out) {
  k_step2_body<T, TS>();
}
#if defined(MJB_SPEC_KERNEL) && MJB_SPEC_KERNEL == 3
}  // namespace mjb
// specialised two-wave step kernel (mjb_step2_spec_source / mjb_step2_spec_load): sizes and the FLAT layout's offsets pinned, model baked in
extern "C" __global__ __launch_bounds__(128, MJB_WPS2) void mjb_k_step2_spec(const mjb::DevModel<float>* mg, const mjb::Lay* lg, mjb::DevData<float> d, mjb::DevDebug<float> dbg,
                                                                       mjb::StepArgs a, mjb::ObsSpecDev obs, float* obs_out) {
  mjb::k_step2_body<float, float>();
}
namespace mjb {
#endif
#if defined(MJB_SPEC_KERNEL) && MJB_SPEC_KERNEL == 1
}  // namespace mjb
// The one kernel of a specialised translation unit (generated by mjb_model_spec_source, loaded by mjb_spec_load): the fp32
// step kernel with MJB_SPEC_ASSUME / MJB_SPEC_ASSUME_LAY pinning the sizes and LDS offsets of ONE compiled model.
extern "C" __global__ __launch_bounds__(64, MJB_WPS) void mjb_k_step_spec(const mjb::DevModel<float>* mg, const mjb::Lay* lg, mjb::DevData<float> d, mjb::DevDebug<float> dbg,
                                                                        mjb::StepArgs a, mjb::ObsSpecDev obs, float* obs_out) {
  mjb::k_step_body<float, float, MJB_SPEC_G>();
}
namespace mjb {
#endif

// Finite-difference columns of mjd_transitionFD (reference linearization.py:16-35): every column advances ONE perturbed replica
// by one step and stores its next state.
//   col 0 = nominal, col 1+2k = +eps on input k, col 2+2k = -eps;  k in [0, 2nv+nu): dq | dv | dctrl
// y_out[(env*ncol + col) * (nq+nv)] = [qpos', qvel'];  valid[(env*ncol+col)] = 0 when a ctrl nudge left ctrlrange.
// A lane-group runs one JOB = a chunk of columns of one kind of one environment, and - like MuJoCo's mj_stepSkip - computes the
// stages a column cannot change only ONCE per job, at the nominal state:
//   ctrl jobs (the nominal column + 2 nu ctrl columns, `cc` per job): position + velocity stages shared, per column actuation -> solve -> Euler
//   velocity jobs (2 nv columns, `cv` per job):                        position stage shared, per column constraint rows -> ... -> Euler
//   position jobs (2 nv columns, one each):                            everything
// (RK4 evaluates the dynamics at four states per step: no sharing.)  Jobs are laid out heaviest first, job-major, so that the
// long ones start first.  Results are the ones the column-per-group kernel gave: a skipped stage would have recomputed the same numbers.
template <typename T, typename TS, int G>
MJB_DEV void k_fd_body(const DevModel<T>* mg, const Lay* lg, const DevData<TS>& d, int ncol, T eps, T* y_out, int* valid, int cv, int cc) {
  extern __shared__ __align__(16) char smem[];
  const int lane = threadIdx.x & (G - 1), sub = threadIdx.x / G;
  const long gid = (long)blockIdx.x * (64 / G) + sub;
  const DevModel<T> MJB_CONST* mp = (const DevModel<T> MJB_CONST*)mg;
  const Lay MJB_CONST* lp = (const Lay MJB_CONST*)lg;
  ModelRef<T> m = MJB_MODEL_OF(mp); LayRef L = *lp;
  MJB_SPEC_ASSUME(m) MJB_SPEC_ASSUME_LAY(L)
  const int nq = m.nq, nv = m.nv, nu = m.nu;
  const int nCj = (1 + 2 * nu + cc - 1) / cc, nVj = (2 * nv + cv - 1) / cv, nQj = 2 * nv, njob = nCj + nVj + nQj;
  if (gid >= (long)d.batch * njob) return;
  const int job = (int)(gid / d.batch), env = (int)(gid % d.batch);
  const bool rk4 = m.integrator == INT_RK4;
  int share, first, cnt, colbase;                               // share: 2 = position + velocity stages, 1 = position stage, 0 = nothing
  if (job < nCj) { share = 2; first = job * cc; cnt = 1 + 2 * nu - first; if (cnt > cc) cnt = cc; colbase = 4 * nv; }       // item i > 0 -> col 4 nv + i; item 0 -> col 0
  else if (job < nCj + nVj) { share = 1; first = (job - nCj) * cv; cnt = 2 * nv - first; if (cnt > cv) cnt = cv; colbase = 1 + 2 * nv; }
  else { share = 0; first = job - nCj - nVj; cnt = 1; colbase = 1; }
  if (rk4) share = 0;
  T* w = (T*)(smem + (size_t)sub * L.bytes);
  int* wi = (int*)(w + L.nT);
  Ctx<T> c(mp, lp, w, wi, lane);
  for (int i = lane; i < nv * nv; i += G) w[L.M + i] = 0;       // structural zeros of the mass matrix (crb_factor fills the rest)
  const int nstage = rk4 ? 4 : 1;
  for (int it = share > 0 ? -1 : 0; it < cnt; it++) {           // it = -1: the shared stages at the nominal state
    for (int i = lane; i < nq; i += G) w[L.qpos + i] = (T)d.qpos[(size_t)env * nq + i];
    for (int i = lane; i < nv; i += G) {
      w[L.qvel + i] = (T)d.qvel[(size_t)env * nv + i];
      w[L.qacc_ws + i] = (T)d.qacc_warmstart[(size_t)env * nv + i];
      w[L.qacc + i] = 0;
      w[L.Mv + i] = 0;
    }
    for (int i = lane; i < nu; i += G) w[L.ctrl + i] = (T)d.ctrl[(size_t)env * nu + i];
    gsync<G>();
    int ok = 1, col = 0;
    if (it >= 0) {
      const int item = first + it;
      col = (share == 2 || (rk4 && job < nCj)) ? (item == 0 ? 0 : colbase + item) : colbase + item;
      if (col > 0) {
        const int k = (col - 1) >> 1;
        const T sgn = ((col - 1) & 1) ? (T)-1 : (T)1;
        if (k < nv) {
          if (lane == 0) w[L.Mv + k] = 1;
          gsync<G>();
          integrate_pos<T, G>(m, w + L.qpos, w + L.Mv, sgn * eps, lane);
        } else if (k < 2 * nv) {
          if (lane == 0) w[L.qvel + (k - nv)] += sgn * eps;
        } else {
          const int a = k - 2 * nv;
          const T v = w[L.ctrl + a] + sgn * eps;
          if (m.actuator_ctrllimited[a] && (v < m.actuator_ctrlrange[2 * a] || v > m.actuator_ctrlrange[2 * a + 1])) ok = 0;
          gsync<G>();
          if (lane == 0 && ok) w[L.ctrl + a] = v;
        }
        gsync<G>();
      }
    }
    const bool do_pos = share == 0 || it < 0;
    const bool do_vel = share == 0 || (share == 1 && it >= 0) || (share == 2 && it < 0);
    const bool do_acc = it >= 0;
    for (int st = 0; st < nstage; st++) {                       // single call site per stage
      if (do_pos) forward_position<T, G>(c);
      if (do_vel) forward_velocity<T, G>(c);
      if (do_acc) forward_acceleration<T, G>(c);
      if (do_acc && nstage == 4) rk4_stage<T, G>(c, st);
    }
    if (!do_acc) continue;
    if (nstage == 1) euler<T, G>(c);
    const size_t slot = (size_t)env * ncol + col;
    T* y = y_out + slot * (nq + nv);
    for (int i = lane; i < nq; i += G) y[i] = w[L.qpos + i];
    for (int i = lane; i < nv; i += G) y[nq + i] = w[L.qvel + i];
    if (lane == 0) valid[slot] = ok;
    gsync<G>();
  }
}
template <typename T, typename TS, int G>
__global__ __launch_bounds__(64) void k_fd(const DevModel<T>* mg, const Lay* lg, DevData<TS> d, int ncol, T eps, T* y_out, int* valid, int cv, int cc) {
  k_fd_body<T, TS, G>(mg, lg, d, ncol, eps, y_out, valid, cv, cc);
}
#if defined(MJB_SPEC_KERNEL) && MJB_SPEC_KERNEL == 2
}  // namespace mjb
// The one kernel of a specialised FINITE-DIFFERENCE translation unit (mjb_fd_spec_source / mjb_fd_spec_load): k_fd<double, TS, G> with the
// float64 layout's offsets and the model's sizes pinned and the model baked in as float64 constant data.
extern "C" __global__ __launch_bounds__(64) void mjb_k_fd_spec(const mjb::DevModel<double>* mg, const mjb::Lay* lg, mjb::DevData<MJB_SPEC_TS> d, int ncol, double eps,
                                                               double* y_out, int* valid, int cv, int cc) {
  mjb::k_fd_body<double, MJB_SPEC_TS, MJB_SPEC_G>(mg, lg, d, ncol, eps, y_out, valid, cv, cc);
}
namespace mjb {
#endif

// A = d[dq';dv']/d[dq;dv]  (2nv x 2nv), B = d[dq';dv']/dctrl (2nv x nu), row-major per environment.
template <typename T>
__global__ void k_fd_combine(const DevModel<T>* mg, int batch, int ncol, int centered, T eps, const T* y, const int* valid, T* A, T* B) {
  ModelRef<T> m = *(const DevModel<T> MJB_CONST*)mg;
  const int nq = m.nq, nv = m.nv, nu = m.nu, nin = 2 * nv + nu, nx = 2 * nv;
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)batch * nin) return;
  int env = (int)(tid / nin), k = (int)(tid % nin);
  const T* y0 = y + ((size_t)env * ncol) * (nq + nv);
  const T* yp = y + ((size_t)env * ncol + 1 + 2 * k) * (nq + nv);
  const T* ym = y + ((size_t)env * ncol + 2 + 2 * k) * (nq + nv);
  bool hp = valid[(size_t)env * ncol + 1 + 2 * k] != 0, hm = centered && valid[(size_t)env * ncol + 2 + 2 * k] != 0;
  const T *a, *b; T den;
  if (hp && hm) { a = ym; b = yp; den = 2 * eps; }
  else if (hp) { a = y0; b = yp; den = eps; }
  else if (hm) { a = ym; b = y0; den = eps; }
  else { a = y0; b = y0; den = 1; }
  T* Ae = A + (size_t)env * nx * nx;
  T* Be = B + (size_t)env * nx * (nu > 0 ? nu : 1);
  for (int j = 0; j < m.njnt; j++) {
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    T dq[6]; int nd;
    if (m.jnt_type[j] == JNT_FREE) {
      nd = 6;
      for (int q = 0; q < 3; q++) dq[q] = (b[qa + q] - a[qa + q]) / den;
      T qa4[4] = {a[qa + 3], a[qa + 4], a[qa + 5], a[qa + 6]}, qb4[4] = {b[qa + 3], b[qa + 4], b[qa + 5], b[qa + 6]}, r[3];
      quat_sub(r, qa4, qb4);
      dq[3] = r[0] / den; dq[4] = r[1] / den; dq[5] = r[2] / den;
    } else { nd = 1; dq[0] = (b[qa] - a[qa]) / den; }
    for (int q = 0; q < nd; q++) {
      if (k < 2 * nv) Ae[(size_t)(da + q) * nx + k] = dq[q]; else Be[(size_t)(da + q) * nu + (k - 2 * nv)] = dq[q];
    }
  }
  for (int r = 0; r < nv; r++) {
    T dv = (b[nq + r] - a[nq + r]) / den;
    if (k < 2 * nv) Ae[(size_t)(nv + r) * nx + k] = dv; else Be[(size_t)(nv + r) * nu + (k - 2 * nv)] = dv;
  }
}

// Jacobians (reference jacobians.py:26-83): kind 0 site, 1 body origin, 2 body com, 3 subtree com.
// out_p / out_r: [batch, nreq, 3, nv]
template <typename T, typename TS, int G>
__global__ __launch_bounds__(64) void k_jac(const DevModel<T>* mg, const Lay* lg, DevData<TS> d, int nreq, const int* kinds, const int* ids, T* out_p, T* out_r) {
  extern __shared__ __align__(16) char smem[];
  const int lane = threadIdx.x & (G - 1), sub = threadIdx.x / G;
  const int env = blockIdx.x * (64 / G) + sub;
  if (env >= d.batch) return;
  const DevModel<T> MJB_CONST* mp = (const DevModel<T> MJB_CONST*)mg;
  const Lay MJB_CONST* lp = (const Lay MJB_CONST*)lg;
  ModelRef<T> m = *mp; LayRef L = *lp;
  T* w = (T*)(smem + (size_t)sub * L.bytes);
  int* wi = (int*)(w + L.nT);
  Ctx<T> c(mp, lp, w, wi, lane);
  const int nq = m.nq, nv = m.nv;
  for (int i = lane; i < nq; i += G) w[L.qpos + i] = (T)d.qpos[(size_t)env * nq + i];
  gsync<G>();
  kinematics<T, G>(c);
  com_pos<T, G>(c);
  for (int r = 0; r < nreq; r++) {
    int kind = kinds[r], id = ids[r];
    T* op = out_p + ((size_t)env * nreq + r) * 3 * nv;
    T* orr = out_r + ((size_t)env * nreq + r) * 3 * nv;
    for (int i = lane; i < nv; i += G) {
      T jp[3] = {0, 0, 0}, jr[3] = {0, 0, 0};
      if (kind == 3) {
        for (int b = id > 0 ? id : 1; b < m.nbody; b++) {
          int p = b; bool inside = id == 0;
          while (p > 0 && !inside) { if (p == id) inside = true; p = m.body_parentid[p]; }
          T mass = m.body_mass[b];
          if (!inside || mass <= 0) continue;
          T pt[3] = {w[L.xipos + 3 * b], w[L.xipos + 3 * b + 1], w[L.xipos + 3 * b + 2]}, tp[3];
          jac_col<T>(c, b, i, pt, tp, (T*)0);
          jp[0] += mass * tp[0]; jp[1] += mass * tp[1]; jp[2] += mass * tp[2];
        }
        T sm = m.body_subtreemass[id];
        if (sm > Num<T>::minval()) { jp[0] /= sm; jp[1] /= sm; jp[2] /= sm; }
      } else {
        int b = kind == 0 ? m.site_bodyid[id] : id;
        const T* src = kind == 0 ? w + L.site_xpos + 3 * id : (kind == 1 ? w + L.xpos + 3 * id : w + L.xipos + 3 * id);
        T pt[3] = {src[0], src[1], src[2]};
        jac_col<T>(c, b, i, pt, jp, jr);
      }
      for (int a = 0; a < 3; a++) { op[a * nv + i] = jp[a]; orr[a * nv + i] = jr[a]; }
    }
  }
}

template <typename TS>
__global__ void k_reset(DevData<TS> d, int nq, int nv, int nu, const TS* qpos, const TS* qvel, const TS* ctrl, double time) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= d.batch) return;
  if (tid == 0) { d.flags[0] = 0; if (d.flags_pin) for (int k = 0; k < 4; k++) d.flags_pin[k] = 0; }
  for (int i = 0; i < nq; i++) d.qpos[tid * nq + i] = qpos[i];
  for (int i = 0; i < nv; i++) { d.qvel[tid * nv + i] = qvel ? qvel[i] : (TS)0; d.qacc[tid * nv + i] = 0; d.qacc_warmstart[tid * nv + i] = 0; }
  for (int i = 0; i < nu; i++) d.ctrl[tid * nu + i] = ctrl ? ctrl[i] : (TS)0;
  d.time[tid] = time;
  for (int i = 0; i < CNT_N; i++) d.counters[tid * CNT_N + i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched linear feedback law for the HOST-DRIVEN path (reference examples/humanoid/controllers/
// lqr.py:147-170): ctrl[e] = clip(u0 - K dx[e] + std o P[(step + e stride) mod nsteps]),  dx = [q (-) q0 ; qvel - v0].
// One gain K [nu, 2nv] for many environments: K dx IS a GEMM [B, 2nv] x [2nv, nu].  fp32: one wave per 32 environments builds
// their dx rows in LDS (quaternion-aware differences, lanes over (environment, joint)), then accumulates the 32 x 32 output tile
// with v_mfma_f32_32x32x2_f32 over k = 2nv in steps of 2 (A operand = dx^T slice from LDS, B operand = K^T slice straight from
// global memory through L2: the same 4 nv nu bytes for every wave), and writes ctrl through the accumulator layout.
// float64 data: the same law with plain VALU dot products (no fp32 rounding of a float64 state).
// ---------------------------------------------------------------------------------------------------------------------
struct FeedbackArgs {
  const void *K, *u0, *q0, *v0;      // dtype of the arithmetic: K [nu, 2nv], u0 [nu], q0 [nq], v0 [nv]
  const void *noise_std, *noise_tab; // optional: std [nu], table [nsteps, nu] (the reference's pre-drawn perturbations); null = none
  int nsteps, step, env_stride;
};
template <typename T>
MJB_DEV void feedback_dx(ModelRef<T> m, const T* qpos, const T* qvel, const T* q0, const T* v0, int j, T* dx) {
  // tangent-space difference of joint j of one environment (mj_differentiatePos(q0, qpos) with dt = 1)
  const int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j], nv = m.nv;
  if (m.jnt_type[j] == JNT_FREE) {
    dx[da] = qpos[qa] - q0[qa]; dx[da + 1] = qpos[qa + 1] - q0[qa + 1]; dx[da + 2] = qpos[qa + 2] - q0[qa + 2];
    T a4[4] = {q0[qa + 3], q0[qa + 4], q0[qa + 5], q0[qa + 6]}, b4[4] = {qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]}, r[3];
    quat_sub(r, a4, b4);
    dx[da + 3] = r[0]; dx[da + 4] = r[1]; dx[da + 5] = r[2];
    for (int k = 0; k < 6; k++) dx[nv + da + k] = qvel[da + k] - v0[da + k];
  } else {
    dx[da] = qpos[qa] - q0[qa];
    dx[nv + da] = qvel[da] - v0[da];
  }
}
template <typename T>
MJB_DEV T feedback_finish(ModelRef<T> m, const FeedbackArgs& a, int env, int act, T kdx) {
  T u = ((const T*)a.u0)[act] - kdx;
  if (a.noise_std && a.noise_tab && a.nsteps > 0) {
    const long idx = ((long)a.step + (long)env * a.env_stride) % a.nsteps;
    u += ((const T*)a.noise_std)[act] * ((const T*)a.noise_tab)[idx * m.nu + act];
  }
  if (m.actuator_ctrllimited[act]) u = t_min(t_max(u, m.actuator_ctrlrange[2 * act]), m.actuator_ctrlrange[2 * act + 1]);
  return u;
}
// fp32: MFMA tile per 32 environments.  Dynamic LDS: 32 * kpad floats (kpad = 2nv rounded up to even).
template <int UNUSED = 0>                                        // a template only so that several translation units may include this header
__global__ __launch_bounds__(64) void k_feedback_mfma(const DevModel<float>* mg, DevData<float> d, FeedbackArgs a) {
  extern __shared__ __align__(16) char smem[];
  float* dxs = (float*)smem;
  ModelRef<float> m = *(const DevModel<float> MJB_CONST*)mg;
  const int lane = threadIdx.x, nv = m.nv, nq = m.nq, nu = m.nu, nx = 2 * nv, kpad = (nx + 1) & ~1;
  const int env0 = blockIdx.x * 32;
  const float* K = (const float*)a.K;
  for (int i = lane; i < 32 * kpad; i += 64) dxs[i] = 0.0f;      // rows of environments beyond the batch and the k padding stay zero
  __syncthreads();
  for (int it = lane; it < 32 * m.njnt; it += 64) {
    const int r = it / m.njnt, j = it - r * m.njnt, env = env0 + r;
    if (env < d.batch) feedback_dx<float>(m, d.qpos + (size_t)env * nq, d.qvel + (size_t)env * nv, (const float*)a.q0, (const float*)a.v0, j, dxs + r * kpad);
  }
  __syncthreads();
  const int h = lane >> 5, c = lane & 31;
  for (int n0 = 0; n0 < nu; n0 += 32) {                          // output tiles of 32 actuators
    mjb_f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    const int act = n0 + c;
    const float* Krow = K + (size_t)(act < nu ? act : 0) * nx;
    for (int kk = 0; kk < kpad; kk += 2) {
      const int k = kk + h;
      const float av = dxs[c * kpad + k];                        // A[i = c (environment)][k]
      const float bv = (act < nu && k < nx) ? Krow[k] : 0.0f;    // B[k][j = c (actuator)] = K[act][k]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (act < nu) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int env = env0 + 8 * (i >> 2) + 4 * h + (i & 3);   // accumulator layout: rows 8q + 4h + t, column c
        if (env < d.batch) d.ctrl[(size_t)env * nu + act] = feedback_finish<float>(m, a, env, act, acc[i]);
      }
    }
  }
}
// float64 (and the reference for tests): one thread per (environment, actuator)
template <typename T>
__global__ void k_feedback_simple(const DevModel<T>* mg, DevData<T> d, FeedbackArgs a, T* dx_scratch) {
  ModelRef<T> m = *(const DevModel<T> MJB_CONST*)mg;
  const int nv = m.nv, nq = m.nq, nu = m.nu, nx = 2 * nv;
  const int env = blockIdx.x;
  if (env >= d.batch) return;
  T* dx = dx_scratch + (size_t)env * nx;
  for (int j = threadIdx.x; j < m.njnt; j += blockDim.x) feedback_dx<T>(m, d.qpos + (size_t)env * nq, d.qvel + (size_t)env * nv, (const T*)a.q0, (const T*)a.v0, j, dx);
  __syncthreads();
  for (int act = threadIdx.x; act < nu; act += blockDim.x) {
    T s = 0;
    for (int k = 0; k < nx; k++) s += ((const T*)a.K)[(size_t)act * nx + k] * dx[k];
    d.ctrl[(size_t)env * nu + act] = feedback_finish<T>(m, a, env, act, s);
  }
}

// Host mirror (mjb_host_view): the six state arrays as float64 in ONE staging block  qpos | qvel | ctrl | qacc | qacc_warmstart | time
// (each [batch, n]), so that a host-driven step costs one pack kernel + ONE device-to-host copy (and the reverse for edits).
template <typename TS>
MJB_DEV bool mirror_locate(const DevData<TS>& d, int nq, int nv, int nu, long i, int& field, long& k) {
  const long B = d.batch, n[6] = {B * nq, B * nv, B * nu, B * nv, B * nv, B};
  long o = i;
  for (int f = 0; f < 6; f++) { if (o < n[f]) { field = f; k = o; return true; } o -= n[f]; }
  return false;
}
template <typename TS>
__global__ void k_mirror_pack(DevData<TS> d, int nq, int nv, int nu, double* out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int f; long k;
  if (i == 0) out[total] = (double)d.flags[0];                  // the sticky engine flags ride along behind the state
  if (i >= total || !mirror_locate(d, nq, nv, nu, i, f, k)) return;
  const TS* src[5] = {d.qpos, d.qvel, d.ctrl, d.qacc, d.qacc_warmstart};
  out[i] = f == 5 ? d.time[k] : (double)src[f][k];
}
template <typename TS>
__global__ void k_mirror_unpack(DevData<TS> d, int nq, int nv, int nu, const double* in, long total, int mask) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int f; long k;
  if (i >= total || !mirror_locate(d, nq, nv, nu, i, f, k) || !((mask >> f) & 1)) return;
  TS* dst[5] = {d.qpos, d.qvel, d.ctrl, d.qacc, d.qacc_warmstart};
  if (f == 5) d.time[k] = in[i]; else dst[f][k] = (TS)in[i];
}

// flat observation gather from the HBM-resident state (reference observations.py:98-174, as_dict=False order)
template <typename TS>
__global__ void k_obs(DevData<TS> d, int nq, int nv, int nu, int nbody, int ngeom, int nsite, int nsensordata, ObsSpecDev s, TS* out) {
  int env = blockIdx.x;
  if (env >= d.batch) return;
  TS* o = out + (size_t)env * s.dim;
  int off = 0;
  const TS* bsrc = (s.flags & 64) ? d.xipos : d.xpos;
  for (int i = threadIdx.x; i < 3 * s.nbody; i += blockDim.x) o[off + i] = bsrc[(size_t)env * 3 * nbody + 3 * s.body_ids[i / 3] + i % 3];
  off += 3 * s.nbody;
  if (s.flags & 4) { for (int i = threadIdx.x; i < nu; i += blockDim.x) o[off + i] = d.ctrl[(size_t)env * nu + i]; off += nu; }
  for (int i = threadIdx.x; i < 3 * s.ngeom; i += blockDim.x) o[off + i] = d.geom_xpos[(size_t)env * 3 * ngeom + 3 * s.geom_ids[i / 3] + i % 3];
  off += 3 * s.ngeom;
  if (s.flags & 1) { for (int i = threadIdx.x; i < nq; i += blockDim.x) o[off + i] = d.qpos[(size_t)env * nq + i]; off += nq; }
  if (s.flags & 2) { for (int i = threadIdx.x; i < nv; i += blockDim.x) o[off + i] = d.qvel[(size_t)env * nv + i]; off += nv; }
  if (s.flags & 8) { for (int i = threadIdx.x; i < nsensordata; i += blockDim.x) o[off + i] = d.sensordata[(size_t)env * nsensordata + i]; off += nsensordata; }
  for (int i = threadIdx.x; i < 3 * s.nsite; i += blockDim.x) o[off + i] = d.site_xpos[(size_t)env * 3 * nsite + 3 * s.site_ids[i / 3] + i % 3];
  off += 3 * s.nsite;
  for (int i = threadIdx.x; i < 3 * s.nsubtree; i += blockDim.x) o[off + i] = d.subtree_com[(size_t)env * 3 * nbody + 3 * s.subtree_ids[i / 3] + i % 3];
  off += 3 * s.nsubtree;
  if (s.flags & 16) { if (threadIdx.x == 0) o[off] = (TS)d.time[env]; }
}

// ---------------------------------------------------------------------------
// launchers (defined in inst_*.hip)
// ---------------------------------------------------------------------------
template <typename T, typename TS>
hipError_t launch_step(int G, const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, const DevDebug<TS>& dbg, const StepArgs& a,
                       const ObsSpecDev& obs, TS* obs_out, hipStream_t stream);
template <typename T, typename TS>
hipError_t launch_step2(const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, const StepArgs& a, const ObsSpecDev& obs, TS* obs_out, hipStream_t stream);
template <typename T, typename TS>
int step_blocks_per_cu(int G, const Lay& L);       // resident workgroups of k_step per CU for this layout (<= 0: unknown)
template <typename T, typename TS>
hipError_t launch_fd(int G, const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, int ncol, int nv, int nu, int chunk, T eps, T* y, int* valid, hipStream_t stream);
template <typename T, typename TS>
hipError_t launch_jac(int G, const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, int nreq, const int* kinds, const int* ids, T* out_p, T* out_r, hipStream_t stream);

#define MJB_DISPATCH_G(G, CALL)                    \
  switch (G) {                                     \
    case 8: { constexpr int GG = 8; CALL; } break;   \
    case 16: { constexpr int GG = 16; CALL; } break; \
    case 64: { constexpr int GG = 64; CALL; } break; \
    default: return hipErrorInvalidValue;          \
  }

template <typename T, typename TS, int G>
hipError_t launch_step_g(const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, const DevDebug<TS>& dbg, const StepArgs& a,
                         const ObsSpecDev& obs, TS* obs_out, hipStream_t stream) {
  const int epb = 64 / G;
  size_t shmem = (size_t)epb * L.bytes;
  auto kern = k_step<T, TS, G>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return e;
  int grid = (d.batch + epb - 1) / epb;
  if (a.chunk_steps > 0 && a.grid_blocks > 0 && a.grid_blocks < grid) grid = a.grid_blocks;       // ticket mode: only the resident workgroups
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), shmem, stream, m, Ldev, d, dbg, a, obs, obs_out);
  return hipGetLastError();
}
template <typename T, typename TS>
hipError_t launch_step2_impl(const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, const StepArgs& a, const ObsSpecDev& obs, TS* obs_out, hipStream_t stream) {
  const size_t shmem = (size_t)L.bytes;
  auto kern = k_step2<T, TS>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return e;
  DevDebug<TS> none = DevDebug<TS>();
  hipLaunchKernelGGL(kern, dim3((unsigned)d.batch), dim3(128), shmem, stream, m, Ldev, d, none, a, obs, obs_out);
  return hipGetLastError();
}
template <typename T, typename TS, int G>
int step_blocks_per_cu_g(const Lay& L) {
  const size_t shmem = (size_t)(64 / G) * L.bytes;
  auto kern = k_step<T, TS, G>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64, shmem) != hipSuccess) return 0;
  return nb;
}
template <typename T, typename TS, int G>
hipError_t launch_fd_g(const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, int ncol, int nv, int nu, int chunk, T eps, T* y, int* valid, hipStream_t stream) {
  const int epb = 64 / G;
  size_t shmem = (size_t)epb * L.bytes;
  auto kern = k_fd<T, TS, G>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return e;
  if (chunk < 1) chunk = 1;
  const int njob = (1 + 2 * nu + chunk - 1) / chunk + (2 * nv + chunk - 1) / chunk + 2 * nv;     // must match k_fd
  long ngroups = (long)d.batch * njob;
  int grid = (int)((ngroups + epb - 1) / epb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), shmem, stream, m, Ldev, d, ncol, eps, y, valid, chunk, chunk);
  return hipGetLastError();
}
template <typename T, typename TS, int G>
hipError_t launch_jac_g(const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, int nreq, const int* kinds, const int* ids, T* out_p, T* out_r, hipStream_t stream) {
  const int epb = 64 / G;
  size_t shmem = (size_t)epb * L.bytes;
  auto kern = k_jac<T, TS, G>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return e;
  int grid = (d.batch + epb - 1) / epb;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), shmem, stream, m, Ldev, d, nreq, kinds, ids, out_p, out_r);
  return hipGetLastError();
}

}  // namespace mjb
