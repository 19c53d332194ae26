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
template <typename T, typename TS, int G>
__global__ __launch_bounds__(64, MJB_WPS) void k_step(const DevModel<T>* mg, const Lay* lg, DevData<TS> d, DevDebug<TS> dbg, StepArgs a, ObsSpecDev obs, TS* obs_out) {
  extern __shared__ __align__(16) char smem[];
  const int lane = threadIdx.x & (G - 1), sub = threadIdx.x / G;
  const int env = blockIdx.x * (64 / G) + sub;
  if (env >= d.batch) return;
  const DevModel<T> MJB_CONST* mp = (const DevModel<T> MJB_CONST*)mg;
  const Lay MJB_CONST* lp = (const Lay MJB_CONST*)lg;
  T* w = (T*)(smem + (size_t)sub * lp->bytes);
  int* wi = (int*)(w + lp->nT);
  env_run<T, TS, G>(mp, lp, d, dbg, a, obs, obs_out, w, wi, env, lane);
}

// Finite-difference columns of mjd_transitionFD (reference linearization.py:16-35): each group
// advances ONE perturbed replica by one step and stores its next state.
//   col 0 = nominal, col 1+2k = +eps on input k, col 2+2k = -eps;  k in [0, 2nv+nu): dq | dv | dctrl
// y_out[(env*ncol + col) * (nq+nv)] = [qpos', qvel'];  valid[(env*ncol+col)] = 0 when a ctrl nudge left ctrlrange.
template <typename T, typename TS, int G>
__global__ __launch_bounds__(64) void k_fd(const DevModel<T>* mg, const Lay* lg, DevData<TS> d, int ncol, T eps, T* y_out, int* valid) {
  extern __shared__ __align__(16) char smem[];
  const int lane = threadIdx.x & (G - 1), sub = threadIdx.x / G;
  const long gid = (long)blockIdx.x * (64 / G) + sub;
  if (gid >= (long)d.batch * ncol) return;
  const int env = (int)(gid / ncol), col = (int)(gid % ncol);
  const DevModel<T> MJB_CONST* mp = (const DevModel<T> MJB_CONST*)mg;
  const Lay MJB_CONST* lp = (const Lay MJB_CONST*)lg;
  ModelRef<T> m = *mp; LayRef L = *lp;
  T* w = (T*)(smem + (size_t)sub * L.bytes);
  int* wi = (int*)(w + L.nT);
  Ctx<T> c(mp, lp, w, wi, lane);
  const int nq = m.nq, nv = m.nv, nu = m.nu;
  for (int i = lane; i < nq; i += G) w[L.qpos + i] = (T)d.qpos[(size_t)env * nq + i];
  for (int i = lane; i < nv; i += G) {
    w[L.qvel + i] = (T)d.qvel[(size_t)env * nv + i];
    w[L.qacc_ws + i] = (T)d.qacc_warmstart[(size_t)env * nv + i];
    w[L.qacc + i] = 0;
    w[L.Mv + i] = 0;
  }
  for (int i = lane; i < nu; i += G) w[L.ctrl + i] = (T)d.ctrl[(size_t)env * nu + i];
  for (int i = lane; i < nv * nv; i += G) w[L.M + i] = 0;       // structural zeros of the mass matrix (crb_factor fills the rest)
  gsync<G>();
  int ok = 1;
  if (col > 0) {
    int k = (col - 1) >> 1;
    T sgn = ((col - 1) & 1) ? (T)-1 : (T)1;
    if (k < nv) {
      if (lane == 0) w[L.Mv + k] = 1;
      gsync<G>();
      integrate_pos<T, G>(m, w + L.qpos, w + L.Mv, sgn * eps, lane);
    } else if (k < 2 * nv) {
      if (lane == 0) w[L.qvel + (k - nv)] += sgn * eps;
    } else {
      int a = k - 2 * nv;
      T v = w[L.ctrl + a] + sgn * eps;
      if (m.actuator_ctrllimited[a] && (v < m.actuator_ctrlrange[2 * a] || v > m.actuator_ctrlrange[2 * a + 1])) ok = 0;
      gsync<G>();
      if (lane == 0 && ok) w[L.ctrl + a] = v;
    }
    gsync<G>();
  }
  const int nstage = m.integrator == INT_RK4 ? 4 : 1;
  for (int st = 0; st < nstage; st++) {
    forward<T, G>(c);
    if (nstage == 4) rk4_stage<T, G>(c, st);
  }
  if (nstage == 1) euler<T, G>(c);
  T* y = y_out + (size_t)gid * (nq + nv);
  for (int i = lane; i < nq; i += G) y[i] = w[L.qpos + i];
  for (int i = lane; i < nv; i += G) y[nq + i] = w[L.qvel + i];
  if (lane == 0) valid[gid] = ok;
}

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
  for (int i = 0; i < nq; i++) d.qpos[tid * nq + i] = qpos[i];
  for (int i = 0; i < nv; i++) { d.qvel[tid * nv + i] = qvel ? qvel[i] : (TS)0; d.qacc[tid * nv + i] = 0; d.qacc_warmstart[tid * nv + i] = 0; }
  for (int i = 0; i < nu; i++) d.ctrl[tid * nu + i] = ctrl ? ctrl[i] : (TS)0;
  d.time[tid] = time;
  for (int i = 0; i < CNT_N; i++) d.counters[tid * CNT_N + i] = 0;
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
hipError_t launch_fd(int G, const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, int ncol, T eps, T* y, int* valid, hipStream_t stream);
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
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), shmem, stream, m, Ldev, d, dbg, a, obs, obs_out);
  return hipGetLastError();
}
template <typename T, typename TS, int G>
hipError_t launch_fd_g(const DevModel<T>* m, const Lay* Ldev, const Lay& L, const DevData<TS>& d, int ncol, T eps, T* y, int* valid, hipStream_t stream) {
  const int epb = 64 / G;
  size_t shmem = (size_t)epb * L.bytes;
  auto kern = k_fd<T, TS, G>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return e;
  long ngroups = (long)d.batch * ncol;
  int grid = (int)((ngroups + epb - 1) / epb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), shmem, stream, m, Ldev, d, ncol, eps, y, valid);
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
