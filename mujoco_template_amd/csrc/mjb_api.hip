// mjb_api.hip — host side of the C ABI declared in include/mjbatch.h.
// Owns device memory (SoA [batch, dof] state in HBM, model constants), launches the
// kernels of mjb_kernels.hpp on the caller's HIP stream.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mjbatch.h"
#include "mjb_host.hpp"
#include "mjb_kernels.hpp"

using namespace mjb;

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                           \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) return fail(MJB_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

struct DevAlloc {
  std::vector<void*> ptrs;
  bool ok = true;
  template <typename X> const X* put(const std::vector<X>& v) {
    size_t n = v.size() ? v.size() : 1;
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(X)) != hipSuccess) { ok = false; return nullptr; }
    ptrs.push_back(p);
    if (v.size() && hipMemcpy(p, v.data(), v.size() * sizeof(X), hipMemcpyHostToDevice) != hipSuccess) ok = false;
    return (const X*)p;
  }
  const float* putf(const std::vector<float>& v) { return put(v); }
  const double* putf(const std::vector<double>& v) { return put(v); }
  const int* puti(const std::vector<int>& v) { return put(v); }
  const unsigned long long* putu(const std::vector<unsigned long long>& v) { return put(v); }
  void release() { for (void* p : ptrs) (void)hipFree(p); ptrs.clear(); }
};

struct ArrayInfo { void* ptr; long per_env; int kind; };  // kind: 0 = data dtype, 1 = float64, 2 = int32
}  // namespace

struct mjbModel {
  HostModel h;
  int disableactuator;
  int iterations;
  double tolerance;
  std::vector<char> blob;                                   // the table this model was created from, serialised (mjb_model_save / mjb_model_field)
  struct Field { std::string name; int dtype; long count; size_t off; };
  std::vector<Field> fields;
  std::map<int, std::vector<std::string>> names;            // object names by mjtObj code (table fields "names_<code>", dtype 2)
};

struct mjbObsSpec {
  ObsSpecDev dev;
  std::vector<void*> owned;
};

struct mjbData {
  mjbModel* model;
  int batch, dtype, G, G_fd, ncon_max, nefc_max, device, env0;
  hipStream_t stream;
  DevAlloc alloc;
  DevModel<float> mf;
  DevModel<double> md;
  Lay Lf, Ld;
  DevModel<float>* mf_dev = nullptr;    // device copies of the structs above (read through the constant address space)
  DevModel<double>* md_dev = nullptr;
  Lay *Lf_dev = nullptr, *Ld_dev = nullptr;
  Lay Lf2;                                  // flat fp32 layout of the two-wave step kernel (k_step2); bytes == 0: that kernel does not apply
  Lay* Lf2_dev = nullptr;
  hipModule_t spec2_mod = nullptr;          // per-model specialised two-wave kernel (mjb_step2_spec_load)
  hipFunction_t spec2_fn = nullptr;
  int ncu = 0;                              // CUs of the device (queried once)
  int two_wave = -1;                        // MJB_TWO_WAVE: 0 never, 1 whenever it applies, -1 (default) the policy in launch()
  int up_disable = -1, up_iter = -1; double up_tol = -1;
  DevData<float> df;
  DevData<double> dd;
  std::map<std::string, ArrayInfo> arrays;
  std::vector<void*> owned;
  // debug dumps
  bool dbg_ready = false;
  DevDebug<float> dbgf;
  DevDebug<double> dbgd;
  std::map<std::string, ArrayInfo> dbg_arrays;
  // device-side feedback controller gains (float and double copies): K [nu, 2nv], u0 [nu], q0 [nq], v0 [nv]
  float* fbf[4] = {nullptr, nullptr, nullptr, nullptr};
  double* fbd[4] = {nullptr, nullptr, nullptr, nullptr};
  // per-model specialised fp32 step kernel (mjb_spec_load); null = generic kernel
  hipModule_t spec_mod = nullptr;
  hipFunction_t spec_fn = nullptr;
  hipModule_t fd_spec_mod = nullptr;       // per-model specialised float64 finite-difference kernel (mjb_fd_spec_load); null = generic k_fd
  hipFunction_t fd_spec_fn = nullptr;
  // work scheduling of k_step (launch()): resident workgroups of the kernel in use on this device; < 0 = not yet queried
  long step_slots = -1;
  int sched_chunk = -1, fair_bit = -1;     // experiment overrides (MJB_CHUNK_STEPS, MJB_FAIR_BIT); -1 = policy below
  unsigned launch_seq = 0;                 // ticket launches so far (tags of the hand-over buffer)
  unsigned long long ticket_next = 0;      // value of the device ticket counter when the next ticket launch starts
  int last_sched[6] = {0, 0, 0, 0, 0, 0};  // of the last mode-0 launch: steps, environment blocks, resident slots, chunk_steps, fair_bit, two waves per environment
  // fd / jac scratch
  double *fd_y = nullptr, *fd_A = nullptr, *fd_B = nullptr;
  double *fd_A_host = nullptr, *fd_B_host = nullptr;          // pinned: the (A, B) blocks leave the device in one async copy each
  // mjb_jac: persistent buffers (grown on demand) - the request (kinds | ids) in pinned memory the kernel reads directly, the result
  // blocks pinned (small requests: written by the kernel itself) and on the device (large requests: one async copy each)
  void* io_pin = nullptr; size_t io_pin_cap = 0;               // pinned staging of mjb_get_array / mjb_set_array (grown on demand)
  int* jac_req_pin = nullptr; int jac_req_cap = 0;
  double *jac_pin[2] = {nullptr, nullptr}, *jac_dev[2] = {nullptr, nullptr};
  size_t jac_pin_cap = 0, jac_dev_cap = 0;
  int* fd_valid = nullptr;
  // standalone feedback law (mjb_feedback_ctrl): optional noise (std [nu], table [nsteps, nu]) in both precisions, dx scratch for float64
  float *fb_noise_f[2] = {nullptr, nullptr};
  double *fb_noise_d[2] = {nullptr, nullptr}, *fb_dx = nullptr;
  int fb_nsteps = 0, fb_env_stride = 0;
  // host mirror (mjb_host_view): ONE pinned float64 block  qpos | qvel | ctrl | qacc | qacc_warmstart | time , each [batch, n],
  // and its device staging twin; filled by one pack kernel + one D2H per mjb_sync_to_host
  double *mirror_host = nullptr, *mirror_dev = nullptr;
  double* mirror_shadow = nullptr;        // host copy of the block as last refreshed / uploaded: what in-place edits are detected against
  unsigned long long mirror_seq = 0;      // sequence number of the last polled host-driven step (completion words behind the flags word)
  size_t mirror_off[7] = {0, 0, 0, 0, 0, 0, 0};             // element offsets of the six fields, [6] = total
  int* flags_pin = nullptr;               // pinned: the four engine-flag bits as words the kernels set (DevData::flags_pin)
  unsigned xfer_timeout = 0;              // ticks of the 100 MHz clock a wave waits for a hand-over (0 = not yet read from MJB_XFER_TIMEOUT_MS)
  int xfer_poison_env = -1;               // test hook MJB_XFER_POISON_ENV (-1 = not yet read)
};

namespace {
// ---- serialised model table: "MJBM0001", int32 nfield, then per field: int32 name length, name, int32 dtype (0 f64, 1 i32,
// 2 bytes), int64 count, payload padded to 8 bytes.  Written by mjb_model_save, read by mjb_model_load; also what
// mjb_model_field serves pointers into. ----
size_t elem_size(int dtype) { return dtype == 0 ? 8 : (dtype == 1 ? 4 : 1); }
bool keep_table(mjbModel* m, const Table& t, std::string& err) {
  std::vector<char>& b = m->blob;
  b.clear(); m->fields.clear(); m->names.clear();
  auto put = [&](const void* p, size_t n) { const char* c = (const char*)p; b.insert(b.end(), c, c + n); };
  put("MJBM0001", 8);
  int32_t nf = t.n; put(&nf, 4);
  for (int i = 0; i < t.n; i++) {
    if (t.dtypes[i] < 0 || t.dtypes[i] > 2 || t.counts[i] < 0) { err = std::string("model field '") + t.names[i] + "': bad dtype/count"; return false; }
    int32_t nl = (int32_t)std::strlen(t.names[i]), dt = t.dtypes[i]; int64_t cnt = t.counts[i];
    put(&nl, 4); put(t.names[i], nl); put(&dt, 4); put(&cnt, 8);
    while (b.size() % 8) b.push_back(0);
    size_t off = b.size(), nb = (size_t)cnt * elem_size(dt);
    if (nb) put(t.ptrs[i], nb);
    while (b.size() % 8) b.push_back(0);
    m->fields.push_back({t.names[i], dt, (long)cnt, off});
    if (dt == 2 && !std::strncmp(t.names[i], "names_", 6)) {           // NUL-terminated names of one object type, in id order
      std::vector<std::string>& v = m->names[std::atoi(t.names[i] + 6)];
      const char* c = (const char*)t.ptrs[i];
      for (long k = 0; k < cnt;) { size_t l = strnlen(c + k, (size_t)(cnt - k)); v.emplace_back(c + k, l); k += (long)l + 1; }
    }
  }
  return true;
}
// parse a blob back into a Table (pointers into `b`)
bool parse_blob(const std::vector<char>& b, std::vector<std::string>& names, std::vector<const void*>& ptrs, std::vector<int>& dts, std::vector<long>& cnts, std::string& err) {
  size_t o = 0;
  auto need = [&](size_t n) { return o + n <= b.size(); };
  if (!need(12) || std::memcmp(b.data(), "MJBM0001", 8)) { err = "not a mjbatch model file (bad magic)"; return false; }
  int32_t nf; std::memcpy(&nf, b.data() + 8, 4); o = 12;
  if (nf < 0 || nf > 100000) { err = "corrupt model file (field count)"; return false; }
  for (int i = 0; i < nf; i++) {
    int32_t nl, dt; int64_t cnt;
    if (!need(4)) { err = "truncated model file"; return false; }
    std::memcpy(&nl, b.data() + o, 4); o += 4;
    if (nl < 0 || nl > 256 || !need((size_t)nl + 12)) { err = "corrupt model file (field name)"; return false; }
    names.emplace_back(b.data() + o, (size_t)nl); o += nl;
    std::memcpy(&dt, b.data() + o, 4); o += 4; std::memcpy(&cnt, b.data() + o, 8); o += 8;
    if (dt < 0 || dt > 2 || cnt < 0) { err = "corrupt model file (dtype/count)"; return false; }
    o = (o + 7) / 8 * 8;
    // count is checked against what is left of the file BEFORE it is multiplied: a crafted int64 count must not wrap the byte size
    if (o > b.size() || (uint64_t)cnt > (uint64_t)(b.size() - o) / elem_size(dt)) { err = "truncated model file"; return false; }
    size_t nb = (size_t)cnt * elem_size(dt);
    ptrs.push_back(b.data() + o); dts.push_back(dt); cnts.push_back((long)cnt);
    o = (o + nb + 7) / 8 * 8;
  }
  return true;
}

// the joints of one environment on the host, float64 (mj_integratePos / mj_differentiatePos act on caller-owned host vectors)
void h_quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}

template <typename X> int dev_alloc(mjbData* d, X** out, size_t n) {
  void* p = nullptr;
  if (n == 0) n = 1;
  if (hipMalloc(&p, n * sizeof(X)) != hipSuccess) return -1;
  if (hipMemset(p, 0, n * sizeof(X)) != hipSuccess) return -1;
  d->owned.push_back(p);
  *out = (X*)p;
  return 0;
}

template <typename TS> int alloc_state(mjbData* d, DevData<TS>& s) {
  const HostModel& h = d->model->h;
  size_t B = (size_t)d->batch;
  s.batch = d->batch;
  int rc = 0;
  rc |= dev_alloc(d, &s.qpos, B * h.nq); rc |= dev_alloc(d, &s.qvel, B * h.nv); rc |= dev_alloc(d, &s.ctrl, B * h.nu);
  rc |= dev_alloc(d, &s.qacc, B * h.nv); rc |= dev_alloc(d, &s.qacc_warmstart, B * h.nv); rc |= dev_alloc(d, &s.time, B);
  rc |= dev_alloc(d, &s.xpos, B * h.nbody * 3); rc |= dev_alloc(d, &s.xquat, B * h.nbody * 4); rc |= dev_alloc(d, &s.xipos, B * h.nbody * 3);
  rc |= dev_alloc(d, &s.site_xpos, B * h.nsite * 3); rc |= dev_alloc(d, &s.geom_xpos, B * h.ngeom * 3);
  rc |= dev_alloc(d, &s.subtree_com, B * h.nbody * 3); rc |= dev_alloc(d, &s.sensordata, B * h.nsensordata);
  rc |= dev_alloc(d, &s.qfrc_inverse, B * h.nv); rc |= dev_alloc(d, &s.actuator_moment, B * h.nu * h.nv);
  rc |= dev_alloc(d, &s.counters, B * CNT_N); rc |= dev_alloc(d, &s.flags, 1);
  rc |= dev_alloc(d, &s.sched, 1);
  if (!d->flags_pin) {                                          // the engine flags as four words of pinned host memory (raise_engine_flags)
    if (hipHostMalloc((void**)&d->flags_pin, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) return -1;
    std::memset(d->flags_pin, 0, 4 * sizeof(int));
  }
  s.flags_pin = d->flags_pin;
  s.xfer = nullptr;                                             // hand-over buffer of the ticket map: allocated by the first launch that needs it
  rc |= dev_alloc(d, &s.prof, (size_t)PH_N + 4 * B);        // + per-environment timeline records of the -DMJB_TIMELINE diagnostic kernel
  if (rc) return -1;
  auto& A = d->arrays;
  A["qpos"] = {s.qpos, h.nq, 0}; A["qvel"] = {s.qvel, h.nv, 0}; A["ctrl"] = {s.ctrl, h.nu, 0}; A["qacc"] = {s.qacc, h.nv, 0};
  A["qacc_warmstart"] = {s.qacc_warmstart, h.nv, 0}; A["time"] = {s.time, 1, 1};
  A["xpos"] = {s.xpos, h.nbody * 3L, 0}; A["xquat"] = {s.xquat, h.nbody * 4L, 0}; A["xipos"] = {s.xipos, h.nbody * 3L, 0};
  A["site_xpos"] = {s.site_xpos, h.nsite * 3L, 0}; A["geom_xpos"] = {s.geom_xpos, h.ngeom * 3L, 0};
  A["subtree_com"] = {s.subtree_com, h.nbody * 3L, 0}; A["sensordata"] = {s.sensordata, h.nsensordata, 0};
  A["qfrc_inverse"] = {s.qfrc_inverse, h.nv, 0}; A["actuator_moment"] = {s.actuator_moment, (long)h.nu * h.nv, 0};
  A["counters"] = {s.counters, CNT_N, 2};
  return 0;
}

template <typename TS> int alloc_debug(mjbData* d, DevDebug<TS>& g) {
  const HostModel& h = d->model->h;
  size_t B = (size_t)d->batch, nv = h.nv, ne = d->nefc_max, nc = d->ncon_max;
  int rc = 0;
  rc |= dev_alloc(d, &g.qM, B * nv * nv); rc |= dev_alloc(d, &g.qfrc_bias, B * nv); rc |= dev_alloc(d, &g.qfrc_passive, B * nv);
  rc |= dev_alloc(d, &g.qfrc_actuator, B * nv); rc |= dev_alloc(d, &g.qacc_smooth, B * nv); rc |= dev_alloc(d, &g.qfrc_constraint, B * nv);
  rc |= dev_alloc(d, &g.efc_J, B * ne * nv); rc |= dev_alloc(d, &g.efc_aref, B * ne); rc |= dev_alloc(d, &g.efc_D, B * ne);
  rc |= dev_alloc(d, &g.efc_pos, B * ne); rc |= dev_alloc(d, &g.efc_force, B * ne); rc |= dev_alloc(d, &g.con, B * nc * CON_STRIDE);
  rc |= dev_alloc(d, &g.cdof, B * 6 * nv); rc |= dev_alloc(d, &g.cinert, B * 10 * h.nbody); rc |= dev_alloc(d, &g.cvel, B * 6 * h.nbody);
  rc |= dev_alloc(d, &g.efc_type, B * ne);
  if (rc) return -1;
  auto& A = d->dbg_arrays;
  A["qM"] = {g.qM, (long)(nv * nv), 0}; A["qfrc_bias"] = {g.qfrc_bias, (long)nv, 0}; A["qfrc_passive"] = {g.qfrc_passive, (long)nv, 0};
  A["qfrc_actuator"] = {g.qfrc_actuator, (long)nv, 0}; A["qacc_smooth"] = {g.qacc_smooth, (long)nv, 0}; A["qfrc_constraint"] = {g.qfrc_constraint, (long)nv, 0};
  A["efc_J"] = {g.efc_J, (long)(ne * nv), 0}; A["efc_aref"] = {g.efc_aref, (long)ne, 0}; A["efc_D"] = {g.efc_D, (long)ne, 0};
  A["efc_pos"] = {g.efc_pos, (long)ne, 0}; A["efc_force"] = {g.efc_force, (long)ne, 0}; A["con"] = {g.con, (long)(nc * CON_STRIDE), 0};
  A["cdof"] = {g.cdof, (long)(6 * nv), 0}; A["cinert"] = {g.cinert, 10L * h.nbody, 0}; A["cvel"] = {g.cvel, 6L * h.nbody, 0};
  A["efc_type"] = {g.efc_type, (long)ne, 2};
  return 0;
}

int refresh_options(mjbData* d) {
  const mjbModel* mm = d->model;
  if (d->up_disable == mm->disableactuator && d->up_iter == mm->iterations && d->up_tol == mm->tolerance) return MJB_OK;
  d->mf.disableactuator = d->md.disableactuator = mm->disableactuator;
  d->mf.iterations = d->md.iterations = mm->iterations;
  d->md.tolerance = mm->tolerance;
  // fp32 floor of the solver tolerance: 1e-7 / 1e-8 were measured (profiles/r02_humanoid_phase_errors.log): +0.6 / +0.8 Newton
  // iterations per step, -3 % / -4.5 % throughput, no change of the humanoid drift curve (the error enters through M and the bias force)
  float tol = (float)mm->tolerance;
  d->mf.tolerance = tol < 1e-6f ? 1e-6f : tol;
  HIPCHK(hipStreamSynchronize(d->stream));
  HIPCHK(hipMemcpy(d->mf_dev, &d->mf, sizeof(d->mf), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d->md_dev, &d->md, sizeof(d->md), hipMemcpyHostToDevice));
  d->up_disable = mm->disableactuator; d->up_iter = mm->iterations; d->up_tol = mm->tolerance;
  return MJB_OK;
}


// Default caps on contacts / constraint rows held in LDS per environment (see mjb_data_create).
void choose_caps(const HostModel& h, int dtype, int lanes, int nconmax, int nefcmax, int& nc_out, int& ne_out) {
  int ne, nc;
  if (nconmax > 0 || nefcmax > 0) {
    nc = nconmax > 0 ? nconmax : (h.ncon_alloc < 32 ? h.ncon_alloc : 32);
    ne = nefcmax > 0 ? nefcmax : (h.nefc_alloc < 96 ? h.nefc_alloc : 96);
  } else {
    const size_t budget = (size_t)160 * 1024 / (dtype == MJB_F32 ? 8 : 4);
    ne = h.nefc_alloc < 96 ? h.nefc_alloc : 96; nc = 0;
    if (h.nefc_alloc <= 96 && h.ncon_alloc <= 32) {
      // the model's own worst case (every candidate pair in contact, every limit active) is small: hold all of it, nothing
      // can ever be dropped (drone2: 20 contacts / 80+ rows when it lands flat); occupancy is not traded against that
      ne = h.nefc_alloc; nc = h.ncon_alloc;
    } else for (;; ne -= 8) {
      // rows bind first (a frictional contact is four rows): for each row cap try 3/8 of it as contact cap and shave that down to a
      // third before giving up rows (humanoid fp32: 64 rows / 23 contacts = 20 448 B, eight slices in 160 KB; 64 / 24 misses by 16 B)
      int hi = ne * 3 / 8; if (hi < 8) hi = 8; if (hi > h.ncon_alloc) hi = h.ncon_alloc;
      int lo = ne / 3; if (lo < 8) lo = 8; if (lo > hi) lo = hi;
      bool fits = false;
      for (nc = hi; nc >= lo; nc--) {
        Lay t = make_layout(h, nc > 0 ? nc : 1, ne > 0 ? ne : 1, dtype == MJB_F32 ? sizeof(float) : sizeof(double));
        if ((size_t)(64 / lanes) * (size_t)t.bytes <= budget) { fits = true; break; }
      }
      if (fits) break;
      if (ne <= 32) { nc = hi; break; }
    }
  }
  if (nc < 1) nc = 1;
  if (ne < 1) ne = 1;
  nc_out = nc; ne_out = ne;
}
int auto_lanes(const HostModel& h, int lanes) { return lanes == 0 ? (h.nv <= 4 ? 8 : (h.nv <= 16 ? 16 : 64)) : lanes; }

// The model itself as `__constant__` data of the specialised translation unit (MJB_SPEC_BAKED, mjb_device.hpp): fill_dev_model runs
// against an allocator that EMITS every table as a C array and hands out recognisable tokens instead of addresses; the filled
// DevModel<float> is then written out word by word as a struct of the same layout (pointer words -> the emitted arrays, everything
// else -> the bit pattern), so the image cannot fall out of step with fill_dev_model or with the struct's member list.
struct EmitAlloc {
  // Two forms.  as_struct = false (default): one `static const __constant__` array per table - its own symbol, so a lane-indexed read
  // is global_load(table address + lane offset) with no further address arithmetic.  as_struct = true (MJB_SPEC_BAKE=struct): every
  // table a member of ONE struct - a single base address, an extra address add per lane-indexed read, far fewer address registers.
  // Measured (scripts/gpu_spec_check.py, same box, B = 4096, M env-steps/s; pointer hops / struct / arrays): humanoid 41.7 / 43.1 /
  // 43.4, drone2 178 / 193 / 218, cart-pole 215 / 230 / 248.  The humanoid kernel (256 VGPRs) spills 61 VGPRs to scratch with the
  // array form (spill stores around the ticket loop, ~7 KB per environment and chunk switch: HBM write traffic 0.15 -> 1.2 GB per
  // 1000-step launch) and far fewer with the struct form; it is still 2 % faster with the arrays (4 % on the two-wave kernel).
  bool as_struct = false;
  std::string text, decl, init;
  int n = 0;
  static constexpr unsigned long long TOKEN = 0x7E57AB1Eull << 32;
  template <typename X, typename F> const X* emit(const std::vector<X>& v, const char* ctype, F fmt) {
    std::string body;
    if (v.empty()) body = "0";
    for (size_t i = 0; i < v.size(); i++) { if (i) body += (i % 16 == 0 ? ",\n " : ","); body += fmt(v[i]); }
    if (as_struct) {
      decl += std::string("  ") + ctype + " t" + std::to_string(n) + "[" + std::to_string(v.empty() ? 1 : v.size()) + "];\n";
      init += "  {" + body + "},\n";
    } else text += std::string("static const __constant__ ") + ctype + " mjb_tab_" + std::to_string(n) + "[] = {" + body + "};\n";
    return (const X*)(uintptr_t)(TOKEN + (unsigned long long)(n++) * 16ull + 16ull);
  }
  std::string table_name(int k) const { return (as_struct ? "mjb_tabs.t" : "mjb_tab_") + std::to_string(k); }
  std::string tables_source() const {
    return as_struct ? "struct MjbBakedTables {\n" + decl + "};\nstatic const __constant__ MjbBakedTables mjb_tabs = {\n" + init + "};\n" : text;
  }
  const float* putf(const std::vector<float>& v) {
    return emit(v, "float", [](float x) {
      if (std::isnan(x)) return std::string("__builtin_nanf(\"\")");
      if (std::isinf(x)) return std::string(x > 0 ? "__builtin_inff()" : "-__builtin_inff()");
      char b[48]; std::snprintf(b, sizeof(b), "%af", (double)x); return std::string(b);             // hex float: exact
    });
  }
  const double* putf(const std::vector<double>& v) {
    return emit(v, "double", [](double x) {
      if (std::isnan(x)) return std::string("__builtin_nan(\"\")");
      if (std::isinf(x)) return std::string(x > 0 ? "__builtin_inf()" : "-__builtin_inf()");
      char b[48]; std::snprintf(b, sizeof(b), "%a", x); return std::string(b);
    });
  }
  const int* puti(const std::vector<int>& v) { return emit(v, "int", [](int x) { return std::to_string(x); }); }
  const unsigned long long* putu(const std::vector<unsigned long long>& v) {
    return emit(v, "unsigned long long", [](unsigned long long x) { char b[40]; std::snprintf(b, sizeof(b), "0x%llxull", x); return std::string(b); });
  }
};

template <typename T>
std::string baked_model_source(const HostModel& h, int ncon_max, int nefc_max, bool as_struct) {
  static_assert(sizeof(DevModel<T>) % 8 == 0, "DevModel<T> is written out in 8-byte words");
  const char* tname = sizeof(T) == 4 ? "float" : "double";
  EmitAlloc ea;
  ea.as_struct = as_struct;
  DevModel<T> m;
  fill_dev_model<T>(h, ea, ncon_max, nefc_max, m);
  const size_t nw = sizeof(m) / 8;
  std::vector<unsigned long long> wv(nw);
  std::memcpy(wv.data(), (const void*)&m, sizeof(m));
  std::string decl = "struct MjbBakedModel {", init = "static const __constant__ MjbBakedModel mjb_baked_model = {";
  for (size_t i = 0; i < nw; i++) {
    const unsigned long long v = wv[i];
    if ((v >> 32) == (EmitAlloc::TOKEN >> 32)) {
      decl += " const void MJB_CONST* p" + std::to_string(i) + ";";
      init += " (const void MJB_CONST*)" + ea.table_name((int)((v - EmitAlloc::TOKEN) / 16 - 1)) + ",";
    } else {
      char b[64];
      decl += " unsigned a" + std::to_string(i) + ", b" + std::to_string(i) + ";";
      std::snprintf(b, sizeof(b), " 0x%xu, 0x%xu,", (unsigned)(v & 0xffffffffull), (unsigned)(v >> 32));
      init += b;
    }
    if (i % 8 == 7) init += "\n ";
  }
  decl += " };\n"; init += " };\n";
  {
    // every emitted table must be referenced by exactly one pointer word of the image (a data word that happens to look like a token,
    // or a table the struct does not point at, would make the image wrong): otherwise no baked model - the kernel reads the device copy
    std::vector<int> seen((size_t)ea.n, 0);
    bool ok = true;
    for (size_t i = 0; i < nw && ok; i++)
      if ((wv[i] >> 32) == (EmitAlloc::TOKEN >> 32)) {
        const unsigned long long k = (wv[i] - EmitAlloc::TOKEN) / 16 - 1;
        if (k >= (unsigned long long)ea.n || (wv[i] - EmitAlloc::TOKEN) % 16 != 0 || seen[(size_t)k]++) ok = false;
      }
    for (int k = 0; k < ea.n && ok; k++) if (seen[(size_t)k] != 1) ok = false;
    if (!ok) return std::string("// (model not baked in: the DevModel image did not map one-to-one onto the emitted tables)\n");
  }
  std::string s = "#include \"mjb_types.hpp\"\n// the model as constant data of this translation unit (tables, then the DevModel image)\n";
  s += ea.tables_source() + decl + init;
  s += std::string("static_assert(sizeof(MjbBakedModel) == sizeof(mjb::DevModel<") + tname + ">), \"baked model image\");\n";
  s += std::string("#define MJB_SPEC_BAKED (*(const mjb::DevModel<") + tname + "> MJB_CONST*)&mjb_baked_model)\n";
  return s;
}

// Translation unit of the specialised fp32 step kernel of one compiled model: the structural sizes of DevModel (never the
// run-time options: disableactuator, iterations, tolerance) and every LDS layout offset become __builtin_assume()s.
// kind 1: the fp32 step kernel (L = the fp32 layout); kind 2: the float64 finite-difference kernel k_fd<double, ts, G> (L = the float64 layout)
std::string spec_source(const HostModel& h, const Lay& L, int G, int ncon_max, int nefc_max, int kind = 1, const char* ts = "float") {
  std::string s = kind == 1 ? "// generated by mjb_model_spec_source(): size- and layout-specialised k_step<float, float, G> of ONE compiled model\n"
                : kind == 2 ? "// generated by mjb_fd_spec_source(): size- and layout-specialised k_fd<double, TS, G> of ONE compiled model\n"
                            : "// generated by mjb_step2_spec_source(): size- and (flat) layout-specialised two-wave step kernel k_step2<float, float> of ONE compiled model\n";
  s += "#define MJB_SPEC_KERNEL " + std::to_string(kind) + "\n#define MJB_SPEC_TS " + ts + "\n#define MJB_SPEC_G " + std::to_string(G) + "\n#define MJB_SPEC_ASSUME(m)";
  auto A = [&](const char* obj, const char* f, long v) { s += std::string(" __builtin_assume((") + obj + ")." + f + " == " + std::to_string(v) + ");"; };
#define SM(f, v) A("m", #f, (long)(v))
  SM(nq, h.nq); SM(nv, h.nv); SM(nu, h.nu); SM(nbody, h.nbody); SM(njnt, h.njnt); SM(ngeom, h.ngeom); SM(nsite, h.nsite);
  SM(ntendon, h.ntendon); SM(nwrap, h.nwrap); SM(nsensor, h.nsensor); SM(nsensordata, h.nsensordata); SM(nkey, h.nkey); SM(npair, h.npair);
  SM(nlevel, h.nlevel); SM(integrator, h.integrator); SM(has_damping, h.has_damping); SM(has_fluid, h.has_fluid); SM(has_accel, h.has_accel);
  SM(nvp, h.nvp); SM(nvshift, h.nvshift); SM(ncon_max, ncon_max); SM(nefc_max, nefc_max); SM(nsiteact, h.siteact.size()); SM(nmpair, h.mpair.size());
  SM(nround, h.nround); SM(nround_inner, h.nround_inner); SM(max_nsub, h.max_nsub); SM(dfs_ok, h.dfs_ok);
#undef SM
  s += "\n#define MJB_SPEC_ASSUME_LAY(L)";
#define SL(f) A("L", #f, (long)L.f)
  SL(qpos); SL(qvel); SL(ctrl); SL(qacc); SL(qacc_ws); SL(qacc_smooth); SL(qfrc_bias); SL(qfrc_passive); SL(qfrc_actuator); SL(qfrc_smooth);
  SL(qfrc_constraint); SL(xpos); SL(xquat); SL(xmat); SL(xipos); SL(ximat); SL(xanchor); SL(xaxis); SL(geom_xpos); SL(geom_xmat); SL(site_xpos);
  SL(site_xmat); SL(subtree_com); SL(cinert); SL(crb); SL(cdof); SL(cdof_dot); SL(cvel); SL(cacc); SL(cfrc); SL(dofbuf); SL(bfrc); SL(M); SL(W);
  SL(ten_length); SL(ten_J); SL(act_force); SL(sens); SL(con); SL(efc_J); SL(efc_pos); SL(efc_D); SL(efc_aref); SL(efc_jar); SL(efc_jv);
  SL(efc_force); SL(efc_KBI); SL(Ma); SL(grad); SL(search); SL(Mv); SL(tmp); SL(cholcol); SL(rk); SL(nT); SL(i_efc_type); SL(i_efc_id);
  SL(i_con_pair); SL(i_scal); SL(i_mail); SL(nI); SL(bytes);
#undef SL
  s += "\n";
  {
    // solimp powers of the model (joint limits, tendon limits, contact pairs): when every one is 1 or 2 (2 is MuJoCo's default) the
    // impedance curve needs no powf, and the specialised kernel is compiled without that code (~2 000 instructions of a cold branch)
    bool le2 = true;
    for (const char* name : {"jnt_solimp", "tendon_solimp", "pair_solimp"}) {
      const auto& v = h.D(name);
      for (size_t k = 4; k < v.size(); k += 5) { const double pw = v[k] > 1.0 ? v[k] : 1.0; if (pw != 1.0 && pw != 2.0) le2 = false; }
    }
    if (le2) s += "#define MJB_SPEC_SOLIMP_POWER_1_OR_2 1\n";       // (model fields are read-only after compilation: only the solver options change at run time)
  }
  {
    const char* e = std::getenv("MJB_SPEC_BAKE");                 // experiments: "off", "struct", "arrays"
    const std::string mode = e ? e : "arrays";
    if (mode != "off" && !std::getenv("MJB_SPEC_NO_BAKE"))
      s += kind != 2 ? baked_model_source<float>(h, ncon_max, nefc_max, mode == "struct") : baked_model_source<double>(h, ncon_max, nefc_max, mode == "struct");
  }
  s += "#include \"mjb_kernels.hpp\"\n";
  return s;
}

// How k_step's work is mapped to workgroups (mjb_kernels.hpp).  With no more environment blocks than the chip holds at once every
// block keeps its workgroup for the whole launch; beyond that the resident workgroups draw (block, chunk-of-steps) tickets, which
// removes the "rounds" of the static map and their tails (profiles/r02_wave_timeline.log: -20 % launch time at 20 steps per launch,
// -9 % at 100, humanoid B = 4096).  Chunk length ~ sqrt(steps)/2: per-chunk cost (state through memory, one ticket) ~2 us against
// a tail of half a chunk.  The priority hand-over (StepArgs::fair_bit) matters where the launch waits for its slowest wave.
static void choose_schedule(mjbData* d, StepArgs& a) {
  a.chunk_steps = 0; a.fair_bit = 0; a.nblk = 0; a.grid_blocks = 0; a.nuniform = 0; a.nchunk = 0;
  if (a.mode != 0) return;
  if (d->sched_chunk == -1) {
    const char* e1 = std::getenv("MJB_CHUNK_STEPS"); const char* e2 = std::getenv("MJB_FAIR_BIT");
    d->sched_chunk = e1 ? std::atoi(e1) : -2; d->fair_bit = e2 ? std::atoi(e2) : -2;
  }
  const int epb = 64 / d->G;
  const long nblk = (d->batch + epb - 1) / epb;
  if (d->step_slots < 0) {
    int ncu = 0, nb = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, d->device) != hipSuccess || ncu < 1) ncu = 256;
    if (d->dtype == MJB_F32 && d->spec_fn) {
      if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, d->spec_fn, 64, (size_t)epb * d->Lf.bytes) != hipSuccess) nb = 0;
    } else nb = d->dtype == MJB_F32 ? step_blocks_per_cu<float, float>(d->G, d->Lf) : step_blocks_per_cu<double, double>(d->G, d->Ld);
    d->step_slots = nb > 0 ? (long)nb * ncu : 0;              // 0 = unknown: keep the static map
  }
  a.fair_bit = d->fair_bit >= 0 ? d->fair_bit : 15;            // 2^15 x 10 ns = 0.33 ms per turn
  a.nblk = (int)nblk; a.grid_blocks = d->step_slots > 0 && d->step_slots < nblk ? (int)d->step_slots : (int)nblk;
  if (d->sched_chunk >= 0) {
    a.chunk_steps = a.nstep > 1 ? d->sched_chunk : 0;
    if (std::getenv("MJB_SCHED_DEBUG")) std::fprintf(stderr, "[mjb] schedule (override): blocks %ld, resident slots %ld, steps %d -> chunk_steps %d, fair_bit %d\n", nblk, d->step_slots, a.nstep, a.chunk_steps, a.fair_bit);
    return;
  }
  if (d->step_slots > 0 && nblk > d->step_slots && a.nstep >= 4) {
    // uniform chunks of c steps, then a guided taper (every further chunk = half of what is left, chunk_plan): the tail of the launch
    // is half of the LAST chunk (one step), so c only trades switches (~2.7 us each: hand-over out, one ticket, hand-over in)
    // against how early imbalance starts to be evened out.  c ~ sqrt(3 N): 8 at N = 20, 17 at N = 100 (measured flat optimum
    // 6..12 and 12..16, profiles/r02_schedule.log), at most 32
    int c = (int)std::lround(std::sqrt(3.0 * (double)a.nstep));
    a.chunk_steps = c < 1 ? 1 : (c > 32 ? 32 : c);
  }
  if (std::getenv("MJB_SCHED_DEBUG")) std::fprintf(stderr, "[mjb] schedule: blocks %ld, resident slots %ld, steps %d -> chunk_steps %d, fair_bit %d\n", nblk, d->step_slots, a.nstep, a.chunk_steps, a.fair_bit);
}

// Bit 3 of the engine flags: a wave of a ticket-mode launch gave up waiting for a hand-over and left its environment where the launch
// found it - the results of that launch are incomplete.  The kernels set the word in pinned host memory, so after any stream
// synchronisation (and at the entry of the next launch) the host sees it without a copy: every entry point that synchronises
// returns MJB_ERR_DEVICE from then on, until mjb_reset clears the flags.
int engine_check(const mjbData* d) {
  if (d->flags_pin && __atomic_load_n(d->flags_pin + 3, __ATOMIC_ACQUIRE) != 0)
    return fail(MJB_ERR_DEVICE, "a ticket-mode launch timed out waiting for a state hand-over (engine flag 8): the environments concerned were not "
                                "advanced; mjb_reset() clears the condition");
  return MJB_OK;
}

int launch(mjbData* d, const StepArgs& a_in, const ObsSpecDev& obs, void* obs_out, bool debug) {
  int rc = engine_check(d);                                    // no further launches on top of a failed one
  if (rc != MJB_OK) return rc;
  rc = refresh_options(d);
  if (rc != MJB_OK) return rc;
  hipError_t e;
  StepArgs a = a_in;
  choose_schedule(d, a);
  if (a.chunk_steps > 0) {
    if (d->dtype != MJB_F32 || a.nstep >= (1 << 20) - 2) a.chunk_steps = 0;     // hand-over words are (fp32, tag) pairs; 20 bits of step index in the tag
  }
  if (a.chunk_steps > 0 && !d->df.xfer) {
    const HostModel& hm = d->model->h;
    if (dev_alloc(d, &d->df.xfer, (size_t)d->batch * (size_t)(hm.nq + 3 * hm.nv + 2))) { d->df.xfer = nullptr; a.chunk_steps = 0; }   // no memory: static map
  }
  if (a.chunk_steps > 0) chunk_plan_counts(a.nstep, a.chunk_steps, a.nuniform, a.nchunk);
  if (a.mode == 0) {
    const int epb = 64 / d->G;
    d->last_sched[0] = a.nstep; d->last_sched[1] = (d->batch + epb - 1) / epb; d->last_sched[2] = (int)(d->step_slots > 0 ? d->step_slots : 0);
    d->last_sched[3] = a.chunk_steps; d->last_sched[4] = a.fair_bit;
  }
  // host-side bookkeeping of the ticket counter and the tag sequence: COMMITTED only after the launch succeeded (a failed launch must
  // not leave the host counter ahead of the device's: every later ticket would then underflow and no step would run)
  unsigned seq_next = d->launch_seq;
  unsigned long long ticket_after = d->ticket_next;
  if (a.chunk_steps > 0) {
    do { seq_next++; } while ((seq_next & 0xFFFu) == 0);
    a.tagbase = (seq_next & 0xFFFu) << 20;
    // the ticket counter is never reset on the hot path: every launched workgroup draws tickets until one is past the end, so a
    // launch advances the counter by exactly (tickets + workgroups) and the next launch starts from there (32-bit: rewound long before it wraps)
    const unsigned long long adv = (unsigned long long)a.nblk * (unsigned)a.nchunk + (unsigned long long)a.grid_blocks;
    if (d->ticket_next + adv > 0xF0000000ull) { HIPCHK(hipMemsetAsync(d->df.sched, 0, sizeof(unsigned), d->stream)); d->ticket_next = 0; }
    a.ticket_base = (unsigned)d->ticket_next;
    ticket_after = d->ticket_next + adv;
    if (d->xfer_timeout == 0) {                                  // default 5 s of wall clock; MJB_XFER_TIMEOUT_MS for tests
      const char* e1 = std::getenv("MJB_XFER_TIMEOUT_MS");
      double ms = e1 ? std::atof(e1) : 5000.0;
      if (!(ms > 0)) ms = 5000.0;
      if (ms > 40000.0) ms = 40000.0;
      d->xfer_timeout = (unsigned)(ms * 1e5);
      if (d->xfer_timeout == 0) d->xfer_timeout = 1;
    }
    if (d->xfer_poison_env < 0) { const char* e2 = std::getenv("MJB_XFER_POISON_ENV"); d->xfer_poison_env = e2 ? std::atoi(e2) + 1 : 0; if (d->xfer_poison_env < 0) d->xfer_poison_env = 0; }
    a.xfer_timeout = d->xfer_timeout; a.xfer_poison_env = d->xfer_poison_env;
  }
  // Two waves per environment (env_run2) when the batch leaves at least half of the step kernel's resident slots empty: stepping
  // launches only (mode 0, no debug dumps), fp32, one wave per environment, nv <= 32, Euler.
  bool two = false;
  if (a.mode == 0 && !debug && d->dtype == MJB_F32 && d->Lf2.bytes > 0 && a.chunk_steps == 0) {
    if (d->two_wave == -1) { const char* e2 = std::getenv("MJB_TWO_WAVE"); d->two_wave = e2 ? (std::atoi(e2) ? 1 : 0) : 2; }
    if (d->ncu <= 0 && (hipDeviceGetAttribute(&d->ncu, hipDeviceAttributeMultiprocessorCount, d->device) != hipSuccess || d->ncu < 1)) d->ncu = 256;
    const int ncu = d->ncu;
    // every environment must be resident at once: the kernel is built for two waves per SIMD (__launch_bounds__(128, 2)), i.e. four
    // workgroups per CU.  Measured on the humanoid (profiles/r02_two_wave.log): x1.15 .. 1.18 up to 512 environments (two SIMDs per
    // environment), x1.12 at 768, x1.09 at 1024, x0.7 beyond (two rounds)
    long wg_per_cu = d->Lf2.bytes > 0 ? (160L * 1024) / d->Lf2.bytes : 0;
    if (wg_per_cu > 4) wg_per_cu = 4;
    two = d->two_wave == 1 || (d->two_wave == 2 && (long)d->batch <= wg_per_cu * ncu);
  }
  d->last_sched[5] = two ? 1 : 0;
  if (two) {
    a.fair_bit = 0;
    if (d->spec2_fn) {
      DevDebug<float> dbgarg; std::memset(&dbgarg, 0, sizeof(dbgarg));
      const DevModel<float>* mg = d->mf_dev; const Lay* lg = d->Lf2_dev;
      DevData<float> dv = d->df; StepArgs av = a; ObsSpecDev ov = obs; float* oo = (float*)obs_out;
      void* args[] = {(void*)&mg, (void*)&lg, (void*)&dv, (void*)&dbgarg, (void*)&av, (void*)&ov, (void*)&oo};
      e = hipModuleLaunchKernel(d->spec2_fn, (unsigned)d->batch, 1, 1, 128, 1, 1, (unsigned)d->Lf2.bytes, d->stream, args, nullptr);
    } else e = launch_step2<float, float>(d->mf_dev, d->Lf2_dev, d->Lf2, d->df, a, obs, (float*)obs_out, d->stream);
  } else
  if (d->dtype == MJB_F32 && d->spec_fn) {               // per-model specialised kernel: same arguments, same grid
    DevDebug<float> dbgarg; std::memset(&dbgarg, 0, sizeof(dbgarg));
    if (debug) dbgarg = d->dbgf;
    const DevModel<float>* mg = d->mf_dev; const Lay* lg = d->Lf_dev;
    DevData<float> dv = d->df; StepArgs av = a; ObsSpecDev ov = obs; float* oo = (float*)obs_out;
    void* args[] = {(void*)&mg, (void*)&lg, (void*)&dv, (void*)&dbgarg, (void*)&av, (void*)&ov, (void*)&oo};
    const int epb = 64 / d->G;
    const unsigned grid = a.chunk_steps > 0 ? (unsigned)a.grid_blocks : (unsigned)((d->batch + epb - 1) / epb);
    e = hipModuleLaunchKernel(d->spec_fn, grid, 1, 1, 64, 1, 1, (unsigned)((size_t)epb * d->Lf.bytes), d->stream, args, nullptr);
  } else if (d->dtype == MJB_F32) {
    DevDebug<float> none; std::memset(&none, 0, sizeof(none));
    e = launch_step<float, float>(d->G, d->mf_dev, d->Lf_dev, d->Lf, d->df, debug ? d->dbgf : none, a, obs, (float*)obs_out, d->stream);
  } else {
    DevDebug<double> none; std::memset(&none, 0, sizeof(none));
    e = launch_step<double, double>(d->G, d->md_dev, d->Ld_dev, d->Ld, d->dd, debug ? d->dbgd : none, a, obs, (double*)obs_out, d->stream);
  }
  if (e != hipSuccess) {
    // the device counter may or may not have been rewound above and no workgroup drew a ticket: start the next ticket launch from a
    // known state (counter zeroed in stream order, host count zero)
    if (a.chunk_steps > 0) { (void)hipMemsetAsync(d->df.sched, 0, sizeof(unsigned), d->stream); d->ticket_next = 0; }
    return fail(MJB_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
  }
  d->launch_seq = seq_next; d->ticket_next = ticket_after;
  return MJB_OK;
}

// pinned staging block of the array getters / setters: the copies run on the data's stream straight to / from pinned memory (no pageable
// staging inside the runtime, no per-call temporary), the widening / narrowing conversion happens between it and the caller's array
int ensure_io_pin(mjbData* d, size_t bytes) {
  if (bytes <= d->io_pin_cap) return MJB_OK;
  if (d->io_pin) (void)hipHostFree(d->io_pin);
  d->io_pin = nullptr; d->io_pin_cap = 0;
  const size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 4;
  HIPCHK(hipHostMalloc(&d->io_pin, cap, hipHostMallocDefault));
  d->io_pin_cap = cap;
  return MJB_OK;
}

int copy_out(mjbData* d, const ArrayInfo& ai, double* host_out) {
  size_t n = (size_t)d->batch * ai.per_env;
  if (n == 0) return MJB_OK;
  const bool wide = ai.kind == 1 || (ai.kind == 0 && d->dtype == MJB_F64);
  if (!wide && ai.kind != 0) return fail(MJB_ERR_ARG, "integer array requested as float64");
  const size_t bytes = n * (wide ? sizeof(double) : sizeof(float));
  int rc = ensure_io_pin(d, bytes);
  if (rc != MJB_OK) return rc;
  HIPCHK(hipMemcpyAsync(d->io_pin, ai.ptr, bytes, hipMemcpyDeviceToHost, d->stream));   // ordered behind whatever the stream still runs
  HIPCHK(hipStreamSynchronize(d->stream));
  { int ec = engine_check(d); if (ec != MJB_OK) return ec; }
  if (wide) std::memcpy(host_out, d->io_pin, bytes);
  else { const float* src = (const float*)d->io_pin; for (size_t i = 0; i < n; i++) host_out[i] = (double)src[i]; }
  return MJB_OK;
}
}  // namespace

extern "C" {

const char* mjb_last_error(void) { return g_err.c_str(); }
// for the other translation units of the library (mjb_mjcf.cpp): set the error string, return the code
int mjb_set_error_(int code, const char* msg) { return fail(code, msg ? msg : ""); }

int mjb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mjb_model_create(int nfield, const char* const* names, const void* const* ptrs, const int* dtypes, const long* counts, mjbModel** out) {
  if (!out) return fail(MJB_ERR_ARG, "out is NULL");
  Table t{nfield, names, ptrs, dtypes, counts};
  mjbModel* m = new mjbModel();
  std::string err;
  if (!m->h.load(t, err)) { delete m; return fail(MJB_ERR_MODEL, err); }
  if (!keep_table(m, t, err)) { delete m; return fail(MJB_ERR_MODEL, err); }
  m->disableactuator = m->h.disableactuator;
  m->iterations = m->h.iterations;
  m->tolerance = m->h.tolerance;
  *out = m;
  return MJB_OK;
}

void mjb_model_free(mjbModel* m) { delete m; }

int mjb_model_set_disableactuator(mjbModel* m, int mask) { if (!m) return fail(MJB_ERR_ARG, "model is NULL"); m->disableactuator = mask; return MJB_OK; }
int mjb_model_set_solver(mjbModel* m, int iterations, double tolerance) {
  if (!m || iterations < 1 || !(tolerance >= 0)) return fail(MJB_ERR_ARG, "bad solver options");
  m->iterations = iterations; m->tolerance = tolerance;
  return MJB_OK;
}

int mjb_data_create(mjbModel* m, int batch, int dtype, int lanes, int nconmax, int nefcmax, int device, int env0, mjbData** out) {
  if (!m || !out) return fail(MJB_ERR_ARG, "model/out is NULL");
  if (batch < 1) return fail(MJB_ERR_ARG, "batch must be >= 1");
  if (dtype != MJB_F32 && dtype != MJB_F64) return fail(MJB_ERR_ARG, "dtype must be MJB_F32 or MJB_F64");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(MJB_ERR_DEVICE, "no HIP device: the batched engine has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(MJB_ERR_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  const HostModel& h = m->h;
  lanes = auto_lanes(h, lanes);
  if (lanes != 8 && lanes != 16 && lanes != 64) return fail(MJB_ERR_ARG, "lanes must be 8, 16 or 64");
  mjbData* d = new mjbData();
  d->model = m; d->batch = batch; d->dtype = dtype; d->G = lanes; d->device = device; d->env0 = env0; d->stream = nullptr;
  // Caps on contacts / constraint rows held in LDS per environment.  Explicit values are taken as given.  Default: the
  // model's own worst case when that is small (<= 96 rows, <= 32 contacts); otherwise the largest (rows <= 96, contacts =
  // 3/8 rows) that still lets 8 fp32 (4 fp64) wavefronts share one CU's 160 KiB — the occupancy step that matters most
  // for throughput (profiles/).  Overflow drops rows and is COUNTED.
  choose_caps(h, dtype, lanes, nconmax, nefcmax, d->ncon_max, d->nefc_max);
  fill_dev_model<float>(h, d->alloc, d->ncon_max, d->nefc_max, d->mf);
  fill_dev_model<double>(h, d->alloc, d->ncon_max, d->nefc_max, d->md);
  if (!d->alloc.ok) { mjb_data_free(d); return fail(MJB_ERR_DEVICE, "device allocation of the model failed"); }
  d->Lf = make_layout(h, d->ncon_max, d->nefc_max, sizeof(float));
  d->Ld = make_layout(h, d->ncon_max, d->nefc_max, sizeof(double));
  // the float64 FD / Jacobian kernels may need more lanes per environment than the step kernel to fit LDS
  d->G_fd = d->G;
  while (d->G_fd < 64 && (size_t)(64 / d->G_fd) * (size_t)d->Ld.bytes > 160 * 1024) d->G_fd = d->G_fd == 8 ? 16 : 64;
  size_t lds = (size_t)(64 / d->G) * (size_t)(dtype == MJB_F32 ? d->Lf.bytes : d->Ld.bytes);
  if (lds > 160 * 1024 || (size_t)(64 / d->G_fd) * (size_t)d->Ld.bytes > 160 * 1024) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "per-workgroup LDS %zu B exceeds 160 KiB (lower nconmax/nefcmax or use more lanes)", lds);
    mjb_data_free(d);
    return fail(MJB_ERR_ARG, buf);
  }
  std::memset(&d->Lf2, 0, sizeof(Lay));
  if (dtype == MJB_F32 && d->G == 64 && h.nv <= 32 && h.integrator != INT_RK4) {
    d->Lf2 = make_layout(h, d->ncon_max, d->nefc_max, sizeof(float), true);
    if ((size_t)d->Lf2.bytes > 64 * 1024) std::memset(&d->Lf2, 0, sizeof(Lay));       // (never for the models this path is for)
  }
  if (dev_alloc(d, &d->mf_dev, 1) || dev_alloc(d, &d->md_dev, 1) || dev_alloc(d, &d->Lf_dev, 1) || dev_alloc(d, &d->Ld_dev, 1) || dev_alloc(d, &d->Lf2_dev, 1) ||
      hipMemcpy(d->Lf2_dev, &d->Lf2, sizeof(Lay), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d->Lf_dev, &d->Lf, sizeof(Lay), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d->Ld_dev, &d->Ld, sizeof(Lay), hipMemcpyHostToDevice) != hipSuccess) {
    mjb_data_free(d);
    return fail(MJB_ERR_DEVICE, "device allocation of the model descriptors failed");
  }
  std::memset(&d->df, 0, sizeof(d->df)); std::memset(&d->dd, 0, sizeof(d->dd));
  int rc = dtype == MJB_F32 ? alloc_state(d, d->df) : alloc_state(d, d->dd);
  if (rc) { mjb_data_free(d); return fail(MJB_ERR_DEVICE, "device allocation of the state failed"); }
  *out = d;
  int r = mjb_reset(d, -1);
  if (r != MJB_OK) { mjb_data_free(d); *out = nullptr; return r; }
  return MJB_OK;
}

void mjb_data_free(mjbData* d) {
  if (d && d->spec_mod) { hipModuleUnload(d->spec_mod); d->spec_mod = nullptr; d->spec_fn = nullptr; }
  if (d && d->fd_spec_mod) { hipModuleUnload(d->fd_spec_mod); d->fd_spec_mod = nullptr; d->fd_spec_fn = nullptr; }
  if (d && d->spec2_mod) { hipModuleUnload(d->spec2_mod); d->spec2_mod = nullptr; d->spec2_fn = nullptr; }
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (void* p : d->owned) (void)hipFree(p);
  if (d->mirror_host) (void)hipHostFree(d->mirror_host);
  std::free(d->mirror_shadow);
  if (d->fd_A_host) (void)hipHostFree(d->fd_A_host);
  if (d->fd_B_host) (void)hipHostFree(d->fd_B_host);
  if (d->jac_req_pin) (void)hipHostFree(d->jac_req_pin);
  if (d->io_pin) (void)hipHostFree(d->io_pin);
  if (d->flags_pin) (void)hipHostFree(d->flags_pin);
  for (int k = 0; k < 2; k++) { if (d->jac_pin[k]) (void)hipHostFree(d->jac_pin[k]); if (d->jac_dev[k]) (void)hipFree(d->jac_dev[k]); }
  d->alloc.release();
  delete d;
}

int mjb_set_stream(mjbData* d, void* hip_stream) { if (!d) return fail(MJB_ERR_ARG, "data is NULL"); d->stream = (hipStream_t)hip_stream; return MJB_OK; }
int mjb_sync(mjbData* d) { if (!d) return fail(MJB_ERR_ARG, "data is NULL"); HIPCHK(hipStreamSynchronize(d->stream)); return engine_check(d); }

int mjb_engine_flags(mjbData* d, int* flags_out) {
  if (!d || !flags_out) return fail(MJB_ERR_ARG, "NULL argument");
  HIPCHK(hipStreamSynchronize(d->stream));
  int fl = 0;
  for (int k = 0; k < 4; k++) if (d->flags_pin && __atomic_load_n(d->flags_pin + k, __ATOMIC_ACQUIRE) != 0) fl |= 1 << k;
  *flags_out = fl;
  return MJB_OK;
}

int mjb_data_info(mjbData* d, int* batch, int* dtype, int* lanes, int* nconmax, int* nefcmax, int* lds_bytes_per_env) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (batch) *batch = d->batch;
  if (dtype) *dtype = d->dtype;
  if (lanes) *lanes = d->G;
  if (nconmax) *nconmax = d->ncon_max;
  if (nefcmax) *nefcmax = d->nefc_max;
  if (lds_bytes_per_env) *lds_bytes_per_env = d->dtype == MJB_F32 ? d->Lf.bytes : d->Ld.bytes;
  return MJB_OK;
}

int mjb_array_ptr(mjbData* d, const char* name, void** dev_ptr, long* per_env, int* dtype) {
  if (!d || !name) return fail(MJB_ERR_ARG, "data/name is NULL");
  auto it = d->arrays.find(name);
  if (it == d->arrays.end()) return fail(MJB_ERR_ARG, std::string("unknown array: ") + name);
  if (dev_ptr) *dev_ptr = it->second.ptr;
  if (per_env) *per_env = it->second.per_env;
  if (dtype) *dtype = it->second.kind == 0 ? d->dtype : (it->second.kind == 1 ? MJB_F64 : 2);
  return MJB_OK;
}

int mjb_get_array(mjbData* d, const char* name, double* host_out) {
  if (!d || !name || !host_out) return fail(MJB_ERR_ARG, "NULL argument");
  auto it = d->arrays.find(name);
  if (it == d->arrays.end()) return fail(MJB_ERR_ARG, std::string("unknown array: ") + name);
  HIPCHK(hipSetDevice(d->device));
  return copy_out(d, it->second, host_out);
}

int mjb_set_array(mjbData* d, const char* name, const double* host_in) {
  if (!d || !name || !host_in) return fail(MJB_ERR_ARG, "NULL argument");
  auto it = d->arrays.find(name);
  if (it == d->arrays.end()) return fail(MJB_ERR_ARG, std::string("unknown array: ") + name);
  const ArrayInfo& ai = it->second;
  size_t n = (size_t)d->batch * ai.per_env;
  if (n == 0) return MJB_OK;
  HIPCHK(hipSetDevice(d->device));
  const bool wide = ai.kind == 1 || (ai.kind == 0 && d->dtype == MJB_F64);
  if (!wide && ai.kind != 0) return fail(MJB_ERR_ARG, "integer arrays are read-only");
  const size_t bytes = n * (wide ? sizeof(double) : sizeof(float));
  int rc = ensure_io_pin(d, bytes);
  if (rc != MJB_OK) return rc;
  if (wide) std::memcpy(d->io_pin, host_in, bytes);
  else { float* dst = (float*)d->io_pin; for (size_t i = 0; i < n; i++) dst[i] = (float)host_in[i]; }
  HIPCHK(hipMemcpyAsync(ai.ptr, d->io_pin, bytes, hipMemcpyHostToDevice, d->stream));    // ordered behind the launches already queued
  HIPCHK(hipStreamSynchronize(d->stream));                                               // the staging block is free again on return
  return MJB_OK;
}

int mjb_get_counters(mjbData* d, int* host_out) {
  if (!d || !host_out) return fail(MJB_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  HIPCHK(hipMemcpy(host_out, d->arrays["counters"].ptr, (size_t)d->batch * CNT_N * sizeof(int), hipMemcpyDeviceToHost));
  return engine_check(d);
}

int mjb_reset(mjbData* d, int key) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  const HostModel& h = d->model->h;
  if (key >= h.nkey) return fail(MJB_ERR_LOOKUP, "keyframe index out of range");
  HIPCHK(hipSetDevice(d->device));
  int threads = 256, grid = (d->batch + threads - 1) / threads;
  double time = key >= 0 ? h.D("key_time")[key] : 0.0;
  if (d->dtype == MJB_F32) {
    const float* qp = key >= 0 ? d->mf.key_qpos + (size_t)key * h.nq : d->mf.qpos0;
    const float* qv = key >= 0 ? d->mf.key_qvel + (size_t)key * h.nv : nullptr;
    const float* cu = key >= 0 ? d->mf.key_ctrl + (size_t)key * h.nu : nullptr;
    hipLaunchKernelGGL(k_reset<float>, dim3(grid), dim3(threads), 0, d->stream, d->df, h.nq, h.nv, h.nu, qp, qv, cu, time);
  } else {
    const double* qp = key >= 0 ? d->md.key_qpos + (size_t)key * h.nq : d->md.qpos0;
    const double* qv = key >= 0 ? d->md.key_qvel + (size_t)key * h.nv : nullptr;
    const double* cu = key >= 0 ? d->md.key_ctrl + (size_t)key * h.nu : nullptr;
    hipLaunchKernelGGL(k_reset<double>, dim3(grid), dim3(threads), 0, d->stream, d->dd, h.nq, h.nv, h.nu, qp, qv, cu, time);
  }
  HIPCHK(hipGetLastError());
  // a reset also clears the sticky engine flags (k_reset, in stream order); after a failed ticket launch the host-visible word must be
  // clear before the next launch's entry check reads it: wait for the reset in that (rare) case
  if (d->flags_pin && __atomic_load_n(d->flags_pin + 3, __ATOMIC_ACQUIRE) != 0) HIPCHK(hipStreamSynchronize(d->stream));
  return MJB_OK;
}

static StepArgs make_args(mjbData* d, int nstep, int ctrl_mode, unsigned seed, unsigned step0, double scale, int mode) {
  StepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.nstep = nstep; a.ctrl_mode = ctrl_mode; a.seed = seed; a.step0 = step0; a.env0 = (unsigned)d->env0;
  a.ctrl_scale = scale; a.dt = d->model->h.timestep; a.mode = mode; a.write_kin = 1; a.obs_every = 0;
  static const int rep = std::getenv("MJB_REPEAT_PHASE") ? std::atoi(std::getenv("MJB_REPEAT_PHASE")) : -1;   // diagnostic kernels only (-DMJB_PHASE_REPEAT)
  a.repeat_phase = rep;
  return a;
}

long mjb_model_spec_source(mjbModel* m, int dtype, int lanes, int nconmax, int nefcmax, char* buf, long cap) {
  if (!m) { fail(MJB_ERR_ARG, "model is NULL"); return -1; }
  if (dtype != MJB_F32) { fail(MJB_ERR_ARG, "only the float32 step kernel is specialised"); return -1; }
  const HostModel& h = m->h;
  lanes = auto_lanes(h, lanes);
  if (lanes != 8 && lanes != 16 && lanes != 64) { fail(MJB_ERR_ARG, "lanes must be 8, 16 or 64"); return -1; }
  int nc, ne;
  choose_caps(h, dtype, lanes, nconmax, nefcmax, nc, ne);
  const std::string src = spec_source(h, make_layout(h, nc, ne, sizeof(float)), lanes, nc, ne);
  if (buf && cap > (long)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
  return (long)src.size();
}

// the finite-difference kernel's translation unit for the creation arguments a data object of this model would get (no GPU needed:
// the same caps, float64 layout and lane-group width mjb_data_create derives)
long mjb_model_fd_spec_source(mjbModel* m, int dtype, int lanes, int nconmax, int nefcmax, char* buf, long cap) {
  if (!m) { fail(MJB_ERR_ARG, "model is NULL"); return -1; }
  if (dtype != MJB_F32 && dtype != MJB_F64) { fail(MJB_ERR_ARG, "dtype must be MJB_F32 or MJB_F64"); return -1; }
  const HostModel& h = m->h;
  lanes = auto_lanes(h, lanes);
  if (lanes != 8 && lanes != 16 && lanes != 64) { fail(MJB_ERR_ARG, "lanes must be 8, 16 or 64"); return -1; }
  int nc, ne;
  choose_caps(h, dtype, lanes, nconmax, nefcmax, nc, ne);
  const Lay Ld = make_layout(h, nc, ne, sizeof(double));
  int gfd = lanes;
  while (gfd < 64 && (size_t)(64 / gfd) * (size_t)Ld.bytes > 160 * 1024) gfd = gfd == 8 ? 16 : 64;
  const std::string src = spec_source(h, Ld, gfd, nc, ne, 2, dtype == MJB_F32 ? "float" : "double");
  if (buf && cap > (long)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
  return (long)src.size();
}

long mjb_spec_source(mjbData* d, char* buf, long cap) {
  if (!d) { fail(MJB_ERR_ARG, "data is NULL"); return -1; }
  if (d->dtype != MJB_F32) { fail(MJB_ERR_ARG, "only the float32 step kernel is specialised"); return -1; }
  const std::string src = spec_source(d->model->h, d->Lf, d->G, d->ncon_max, d->nefc_max);
  if (buf && cap > (long)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
  return (long)src.size();
}

int mjb_spec_load(mjbData* d, const void* image, long nbytes) {
  if (!d || !image || nbytes <= 0) return fail(MJB_ERR_ARG, "NULL argument");
  if (d->dtype != MJB_F32) return fail(MJB_ERR_ARG, "only the float32 step kernel is specialised");
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  if (d->spec_mod) { hipModuleUnload(d->spec_mod); d->spec_mod = nullptr; d->spec_fn = nullptr; }
  hipModule_t mod; hipFunction_t fn;
  hipError_t e = hipModuleLoadData(&mod, image);
  if (e != hipSuccess) return fail(MJB_ERR_DEVICE, std::string("hipModuleLoadData: ") + hipGetErrorString(e));
  e = hipModuleGetFunction(&fn, mod, "mjb_k_step_spec");
  if (e != hipSuccess) { hipModuleUnload(mod); return fail(MJB_ERR_DEVICE, "code object has no mjb_k_step_spec kernel"); }
  d->spec_mod = mod; d->spec_fn = fn; d->step_slots = -1;        // occupancy of the kernel in use changed
  return MJB_OK;
}

int mjb_spec_unload(mjbData* d) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (d->spec_mod) { HIPCHK(hipStreamSynchronize(d->stream)); hipModuleUnload(d->spec_mod); d->spec_mod = nullptr; d->spec_fn = nullptr; d->step_slots = -1; }
  return MJB_OK;
}

// the two-wave step kernel (small batches) of this data object / of a model's default creation arguments
long mjb_step2_spec_source(mjbData* d, char* buf, long cap) {
  if (!d) { fail(MJB_ERR_ARG, "data is NULL"); return -1; }
  if (d->Lf2.bytes <= 0) { fail(MJB_ERR_ARG, "the two-wave step kernel does not apply to this data object (fp32, one wave per environment, nv <= 32, Euler)"); return -1; }
  const std::string src = spec_source(d->model->h, d->Lf2, 64, d->ncon_max, d->nefc_max, 3, "float");
  if (buf && cap > (long)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
  return (long)src.size();
}
long mjb_model_step2_spec_source(mjbModel* m, int lanes, int nconmax, int nefcmax, char* buf, long cap) {
  if (!m) { fail(MJB_ERR_ARG, "model is NULL"); return -1; }
  const HostModel& h = m->h;
  lanes = auto_lanes(h, lanes);
  if (lanes != 64 || h.nv > 32 || h.integrator == INT_RK4) { fail(MJB_ERR_ARG, "the two-wave step kernel does not apply to this model"); return -1; }
  int nc, ne;
  choose_caps(h, MJB_F32, lanes, nconmax, nefcmax, nc, ne);
  const std::string src = spec_source(h, make_layout(h, nc, ne, sizeof(float), true), 64, nc, ne, 3, "float");
  if (buf && cap > (long)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
  return (long)src.size();
}
int mjb_step2_spec_load(mjbData* d, const void* image, long nbytes) {
  if (!d || !image || nbytes <= 0) return fail(MJB_ERR_ARG, "NULL argument");
  if (d->Lf2.bytes <= 0) return fail(MJB_ERR_ARG, "the two-wave step kernel does not apply to this data object");
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  if (d->spec2_mod) { hipModuleUnload(d->spec2_mod); d->spec2_mod = nullptr; d->spec2_fn = nullptr; }
  hipModule_t mod; hipFunction_t fn;
  hipError_t e = hipModuleLoadData(&mod, image);
  if (e != hipSuccess) return fail(MJB_ERR_DEVICE, std::string("hipModuleLoadData: ") + hipGetErrorString(e));
  e = hipModuleGetFunction(&fn, mod, "mjb_k_step2_spec");
  if (e != hipSuccess) { hipModuleUnload(mod); return fail(MJB_ERR_DEVICE, "code object has no mjb_k_step2_spec kernel"); }
  d->spec2_mod = mod; d->spec2_fn = fn;
  return MJB_OK;
}
int mjb_step2_spec_unload(mjbData* d) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (d->spec2_mod) { HIPCHK(hipStreamSynchronize(d->stream)); hipModuleUnload(d->spec2_mod); d->spec2_mod = nullptr; d->spec2_fn = nullptr; }
  return MJB_OK;
}

long mjb_fd_spec_source(mjbData* d, char* buf, long cap) {
  if (!d) { fail(MJB_ERR_ARG, "data is NULL"); return -1; }
  const std::string src = spec_source(d->model->h, d->Ld, d->G_fd, d->ncon_max, d->nefc_max, 2, d->dtype == MJB_F32 ? "float" : "double");
  if (buf && cap > (long)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
  return (long)src.size();
}

int mjb_fd_spec_load(mjbData* d, const void* image, long nbytes) {
  if (!d || !image || nbytes <= 0) return fail(MJB_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  if (d->fd_spec_mod) { hipModuleUnload(d->fd_spec_mod); d->fd_spec_mod = nullptr; d->fd_spec_fn = nullptr; }
  hipModule_t mod; hipFunction_t fn;
  hipError_t e = hipModuleLoadData(&mod, image);
  if (e != hipSuccess) return fail(MJB_ERR_DEVICE, std::string("hipModuleLoadData: ") + hipGetErrorString(e));
  e = hipModuleGetFunction(&fn, mod, "mjb_k_fd_spec");
  if (e != hipSuccess) { hipModuleUnload(mod); return fail(MJB_ERR_DEVICE, "code object has no mjb_k_fd_spec kernel"); }
  d->fd_spec_mod = mod; d->fd_spec_fn = fn;
  return MJB_OK;
}

int mjb_fd_spec_unload(mjbData* d) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (d->fd_spec_mod) { HIPCHK(hipStreamSynchronize(d->stream)); hipModuleUnload(d->fd_spec_mod); d->fd_spec_mod = nullptr; d->fd_spec_fn = nullptr; }
  return MJB_OK;
}

int mjb_forward(mjbData* d) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  HIPCHK(hipSetDevice(d->device));
  ObsSpecDev none; std::memset(&none, 0, sizeof(none));
  return launch(d, make_args(d, 1, MJB_CTRL_KEEP, 0, 0, 1.0, 1), none, nullptr, false);
}

int mjb_inverse(mjbData* d) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  HIPCHK(hipSetDevice(d->device));
  ObsSpecDev none; std::memset(&none, 0, sizeof(none));
  return launch(d, make_args(d, 1, MJB_CTRL_KEEP, 0, 0, 1.0, 2), none, nullptr, false);
}

int mjb_step(mjbData* d, int nstep) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (nstep < 1) return fail(MJB_ERR_ARG, "nstep must be >= 1");
  HIPCHK(hipSetDevice(d->device));
  ObsSpecDev none; std::memset(&none, 0, sizeof(none));
  return launch(d, make_args(d, nstep, MJB_CTRL_KEEP, 0, 0, 1.0, 0), none, nullptr, false);
}

int mjb_rollout(mjbData* d, int nstep, int ctrl_mode, unsigned seed, unsigned step0, double ctrl_scale,
                const mjbObsSpec* spec, void* obs_out_dev, int obs_every) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (nstep < 1) return fail(MJB_ERR_ARG, "nstep must be >= 1");
  if (ctrl_mode < 0 || ctrl_mode > 3) return fail(MJB_ERR_ARG, "bad ctrl_mode");
  if (ctrl_mode == 3 && !d->fbd[0]) return fail(MJB_ERR_ARG, "MJB_CTRL_FEEDBACK needs mjb_set_feedback() first");
  HIPCHK(hipSetDevice(d->device));
  StepArgs a = make_args(d, nstep, ctrl_mode, seed, step0, ctrl_scale, 0);
  if (ctrl_mode == 3) {
    if (d->dtype == MJB_F32) { a.fb_K = d->fbf[0]; a.fb_u0 = d->fbf[1]; a.fb_q0 = d->fbf[2]; a.fb_v0 = d->fbf[3]; }
    else { a.fb_K = d->fbd[0]; a.fb_u0 = d->fbd[1]; a.fb_q0 = d->fbd[2]; a.fb_v0 = d->fbd[3]; }
    if (d->fb_nsteps > 0) {
      a.fb_nsteps = d->fb_nsteps; a.fb_env_stride = d->fb_env_stride;
      a.fb_noise_std = d->dtype == MJB_F32 ? (const void*)d->fb_noise_f[0] : (const void*)d->fb_noise_d[0];
      a.fb_noise_tab = d->dtype == MJB_F32 ? (const void*)d->fb_noise_f[1] : (const void*)d->fb_noise_d[1];
    }
  }
  ObsSpecDev obs; std::memset(&obs, 0, sizeof(obs));
  if (spec && obs_out_dev && obs_every > 0) { obs = spec->dev; a.obs_every = obs_every; }
  return launch(d, a, obs, obs_out_dev, false);
}

int mjb_set_feedback(mjbData* d, const double* K, const double* u0, const double* q0, const double* v0) {
  if (!d || !K || !u0 || !q0) return fail(MJB_ERR_ARG, "NULL argument");
  const HostModel& h = d->model->h;
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  const size_t n[4] = {(size_t)h.nu * 2 * h.nv, (size_t)h.nu, (size_t)h.nq, (size_t)h.nv};
  const double* src[4] = {K, u0, q0, v0};
  for (int k = 0; k < 4; k++) {
    if (!d->fbd[k] && (dev_alloc(d, &d->fbd[k], n[k]) || dev_alloc(d, &d->fbf[k], n[k]))) return fail(MJB_ERR_DEVICE, "device allocation of feedback gains failed");
    std::vector<double> vd(n[k], 0.0);
    if (src[k]) vd.assign(src[k], src[k] + n[k]);
    std::vector<float> vf(vd.begin(), vd.end());
    if (n[k]) {
      HIPCHK(hipMemcpy(d->fbd[k], vd.data(), n[k] * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d->fbf[k], vf.data(), n[k] * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  return MJB_OK;
}


int mjb_set_feedback_noise(mjbData* d, const double* noise_std, const double* noise_table, int nsteps, int env_stride) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  d->fb_env_stride = env_stride;
  if (!noise_std || !noise_table || nsteps < 1) { d->fb_nsteps = 0; return MJB_OK; }      // switch the noise term off
  const HostModel& h = d->model->h;
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  const size_t n[2] = {(size_t)h.nu, (size_t)nsteps * h.nu};
  const double* src[2] = {noise_std, noise_table};
  for (int k = 0; k < 2; k++) {
    double* pd = nullptr; float* pf = nullptr;
    if (dev_alloc(d, &pd, n[k]) || dev_alloc(d, &pf, n[k])) return fail(MJB_ERR_DEVICE, "device allocation of the feedback noise failed");
    std::vector<float> vf(src[k], src[k] + n[k]);
    HIPCHK(hipMemcpy(pd, src[k], n[k] * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pf, vf.data(), n[k] * sizeof(float), hipMemcpyHostToDevice));
    d->fb_noise_d[k] = pd; d->fb_noise_f[k] = pf;             // earlier tables stay owned by the data object until it is freed
  }
  d->fb_nsteps = nsteps;
  return MJB_OK;
}

int mjb_feedback_ctrl(mjbData* d, int step) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (!d->fbd[0]) return fail(MJB_ERR_ARG, "mjb_feedback_ctrl needs mjb_set_feedback() first");
  const HostModel& h = d->model->h;
  if (h.nu == 0) return MJB_OK;
  HIPCHK(hipSetDevice(d->device));
  { int rc0 = refresh_options(d); if (rc0 != MJB_OK) return rc0; }
  FeedbackArgs a;
  std::memset(&a, 0, sizeof(a));
  a.nsteps = d->fb_nsteps; a.step = step; a.env_stride = d->fb_env_stride;
  if (d->dtype == MJB_F32) {
    a.K = d->fbf[0]; a.u0 = d->fbf[1]; a.q0 = d->fbf[2]; a.v0 = d->fbf[3];
    if (d->fb_nsteps > 0) { a.noise_std = d->fb_noise_f[0]; a.noise_tab = d->fb_noise_f[1]; }
    const int kpad = (2 * h.nv + 1) & ~1;
    const size_t shmem = (size_t)32 * kpad * sizeof(float);
    if (shmem > 64 * 1024) return fail(MJB_ERR_ARG, "mjb_feedback_ctrl: 2 nv too large for the LDS tile");
    hipLaunchKernelGGL(k_feedback_mfma<0>, dim3((unsigned)((d->batch + 31) / 32)), dim3(64), shmem, d->stream, (const DevModel<float>*)d->mf_dev, d->df, a);
  } else {
    a.K = d->fbd[0]; a.u0 = d->fbd[1]; a.q0 = d->fbd[2]; a.v0 = d->fbd[3];
    if (d->fb_nsteps > 0) { a.noise_std = d->fb_noise_d[0]; a.noise_tab = d->fb_noise_d[1]; }
    if (!d->fb_dx && dev_alloc(d, &d->fb_dx, (size_t)d->batch * 2 * h.nv)) return fail(MJB_ERR_DEVICE, "device allocation of the feedback scratch failed");
    hipLaunchKernelGGL(k_feedback_simple<double>, dim3((unsigned)d->batch), dim3(64), 0, d->stream, (const DevModel<double>*)d->md_dev, d->dd, a, d->fb_dx);
  }
  HIPCHK(hipGetLastError());
  return MJB_OK;
}

int mjb_obs_spec_create(mjbData* d, int flags, int nsite, const int* site_ids, int nbody, const int* body_ids,
                        int ngeom, const int* geom_ids, int nsubtree, const int* subtree_ids, mjbObsSpec** out) {
  if (!d || !out) return fail(MJB_ERR_ARG, "NULL argument");
  const HostModel& h = d->model->h;
  for (int i = 0; i < nsite; i++) if (site_ids[i] < 0 || site_ids[i] >= h.nsite) return fail(MJB_ERR_LOOKUP, "site id out of range");
  for (int i = 0; i < nbody; i++) if (body_ids[i] < 0 || body_ids[i] >= h.nbody) return fail(MJB_ERR_LOOKUP, "body id out of range");
  for (int i = 0; i < ngeom; i++) if (geom_ids[i] < 0 || geom_ids[i] >= h.ngeom) return fail(MJB_ERR_LOOKUP, "geom id out of range");
  for (int i = 0; i < nsubtree; i++) if (subtree_ids[i] < 0 || subtree_ids[i] >= h.nbody) return fail(MJB_ERR_LOOKUP, "subtree body id out of range");
  HIPCHK(hipSetDevice(d->device));
  mjbObsSpec* s = new mjbObsSpec();
  std::memset(&s->dev, 0, sizeof(s->dev));
  auto up = [&](const int* ids, int n) -> const int* {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(int) * (n > 0 ? n : 1)) != hipSuccess) return nullptr;
    s->owned.push_back(p);
    if (n > 0) (void)hipMemcpy(p, ids, sizeof(int) * n, hipMemcpyHostToDevice);
    return (const int*)p;
  };
  s->dev.flags = flags; s->dev.nsite = nsite; s->dev.nbody = nbody; s->dev.ngeom = ngeom; s->dev.nsubtree = nsubtree;
  s->dev.site_ids = up(site_ids, nsite); s->dev.body_ids = up(body_ids, nbody); s->dev.geom_ids = up(geom_ids, ngeom); s->dev.subtree_ids = up(subtree_ids, nsubtree);
  int dim = 3 * (nsite + nbody + ngeom + nsubtree);
  if (flags & 1) dim += h.nq;
  if (flags & 2) dim += h.nv;
  if (flags & 4) dim += h.nu;
  if (flags & 8) dim += h.nsensordata;
  if (flags & 16) dim += 1;
  s->dev.dim = dim;
  *out = s;
  return MJB_OK;
}

void mjb_obs_spec_free(mjbObsSpec* s) {
  if (!s) return;
  for (void* p : s->owned) (void)hipFree(p);
  delete s;
}

int mjb_obs_dim(const mjbObsSpec* s) { return s ? s->dev.dim : -1; }

int mjb_obs_gather(mjbData* d, const mjbObsSpec* s, void* out_dev) {
  if (!d || !s || !out_dev) return fail(MJB_ERR_ARG, "NULL argument");
  const HostModel& h = d->model->h;
  HIPCHK(hipSetDevice(d->device));
  if (d->dtype == MJB_F32)
    hipLaunchKernelGGL(k_obs<float>, dim3(d->batch), dim3(64), 0, d->stream, d->df, h.nq, h.nv, h.nu, h.nbody, h.ngeom, h.nsite, h.nsensordata, s->dev, (float*)out_dev);
  else
    hipLaunchKernelGGL(k_obs<double>, dim3(d->batch), dim3(64), 0, d->stream, d->dd, h.nq, h.nv, h.nu, h.nbody, h.ngeom, h.nsite, h.nsensordata, s->dev, (double*)out_dev);
  HIPCHK(hipGetLastError());
  return MJB_OK;
}

static int transition_fd_impl(mjbData* d, double eps, int centered) {
  if (!d) return fail(MJB_ERR_ARG, "NULL argument");
  if (!(eps > 0)) return fail(MJB_ERR_ARG, "eps must be > 0");
  const HostModel& h = d->model->h;
  HIPCHK(hipSetDevice(d->device));
  { int rc0 = refresh_options(d); if (rc0 != MJB_OK) return rc0; }
  const int nin = 2 * h.nv + h.nu, ncol = 1 + 2 * nin, nx = 2 * h.nv;
  size_t B = (size_t)d->batch;
  if (!d->fd_y) {
    if (dev_alloc(d, &d->fd_y, B * ncol * (h.nq + h.nv)) || dev_alloc(d, &d->fd_valid, B * ncol) ||
        dev_alloc(d, &d->fd_A, B * nx * nx) || dev_alloc(d, &d->fd_B, B * nx * (h.nu > 0 ? h.nu : 1)))
      return fail(MJB_ERR_DEVICE, "device allocation of FD scratch failed");
    HIPCHK(hipHostMalloc((void**)&d->fd_A_host, B * nx * nx * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&d->fd_B_host, B * nx * (h.nu > 0 ? h.nu : 1) * sizeof(double), hipHostMallocDefault));
  }
  // columns per job (k_fd shares the stages a chunk of columns cannot change): as many as keep >= ~4 jobs per residency slot,
  // at most 8; a single environment keeps one column per job (latency over throughput)
  int chunk = 8;
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, d->device) != hipSuccess || ncu < 1) ncu = 256;
    const size_t per_wg = (size_t)(64 / d->G_fd) * (size_t)d->Ld.bytes;
    size_t wg_per_cu = per_wg ? (size_t)160 * 1024 / per_wg : 8;
    if (wg_per_cu > 32) wg_per_cu = 32;
    if (wg_per_cu < 1) wg_per_cu = 1;
    const long slots = (long)wg_per_cu * ncu * (64 / d->G_fd);
    const long want = (long)B * ncol / (4 * slots);
    if (const char* e = std::getenv("MJB_FD_CHUNK")) chunk = std::atoi(e);        // experiments (scripts/gpu_fd_timing.py)
    else chunk = want < 1 ? 1 : (want > 8 ? 8 : (int)want);
  }
  hipError_t e;
  if (d->fd_spec_fn) {                                          // per-model specialised kernel: same arguments, same grid as launch_fd_g
    if (chunk < 1) chunk = 1;
    const int epb = 64 / d->G_fd;
    const int njob = (1 + 2 * h.nu + chunk - 1) / chunk + (2 * h.nv + chunk - 1) / chunk + 2 * h.nv;
    const long ngroups = (long)d->batch * njob;
    const DevModel<double>* mg = d->md_dev; const Lay* lg = d->Ld_dev;
    DevData<float> dvf = d->df; DevData<double> dvd = d->dd;
    int ncol_ = ncol, cv = chunk, cc = chunk; double eps_ = eps; double* yy = d->fd_y; int* vv = d->fd_valid;
    void* args[] = {(void*)&mg, (void*)&lg, d->dtype == MJB_F32 ? (void*)&dvf : (void*)&dvd, (void*)&ncol_, (void*)&eps_, (void*)&yy, (void*)&vv, (void*)&cv, (void*)&cc};
    e = hipModuleLaunchKernel(d->fd_spec_fn, (unsigned)((ngroups + epb - 1) / epb), 1, 1, 64, 1, 1, (unsigned)((size_t)epb * d->Ld.bytes), d->stream, args, nullptr);
  } else
  e = d->dtype == MJB_F32 ? launch_fd<double, float>(d->G_fd, d->md_dev, d->Ld_dev, d->Ld, d->df, ncol, h.nv, h.nu, chunk, eps, d->fd_y, d->fd_valid, d->stream)
                          : launch_fd<double, double>(d->G_fd, d->md_dev, d->Ld_dev, d->Ld, d->dd, ncol, h.nv, h.nu, chunk, eps, d->fd_y, d->fd_valid, d->stream);
  if (e != hipSuccess) return fail(MJB_ERR_DEVICE, std::string("fd launch: ") + hipGetErrorString(e));
  long nthreads = (long)B * nin;
  // a few environments (the reference's batch-1 loops with needs_linearization controllers): the combine kernel writes (A, B) straight
  // into the pinned result blocks (device-visible) - no staging copies; larger batches keep the device blocks + one async copy each
  const bool zero_copy = B * nx * (nx + (size_t)h.nu) * sizeof(double) <= (size_t)256 * 1024;
  hipLaunchKernelGGL(k_fd_combine<double>, dim3((unsigned)((nthreads + 127) / 128)), dim3(128), 0, d->stream, (const DevModel<double>*)d->md_dev, d->batch, ncol, centered, eps,
                     (const double*)d->fd_y, (const int*)d->fd_valid, zero_copy ? d->fd_A_host : d->fd_A, zero_copy ? d->fd_B_host : d->fd_B);
  HIPCHK(hipGetLastError());
  if (!zero_copy) {
    HIPCHK(hipMemcpyAsync(d->fd_A_host, d->fd_A, B * nx * nx * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    if (h.nu > 0) HIPCHK(hipMemcpyAsync(d->fd_B_host, d->fd_B, B * nx * h.nu * sizeof(double), hipMemcpyDeviceToHost, d->stream));
  }
  HIPCHK(hipStreamSynchronize(d->stream));
  return MJB_OK;
}

int mjb_transition_fd(mjbData* d, double eps, int centered, double* A_host, double* B_host) {
  if (!d || !A_host || !B_host) return fail(MJB_ERR_ARG, "NULL argument");
  int rc = transition_fd_impl(d, eps, centered);
  if (rc != MJB_OK) return rc;
  const HostModel& h = d->model->h;
  const size_t B = (size_t)d->batch, nx = 2 * (size_t)h.nv;
  std::memcpy(A_host, d->fd_A_host, B * nx * nx * sizeof(double));
  if (h.nu > 0) std::memcpy(B_host, d->fd_B_host, B * nx * h.nu * sizeof(double));
  return MJB_OK;
}

int mjb_transition_fd_pinned(mjbData* d, double eps, int centered, const double** A_pinned, const double** B_pinned) {
  if (!d || !A_pinned || !B_pinned) return fail(MJB_ERR_ARG, "NULL argument");
  int rc = transition_fd_impl(d, eps, centered);
  if (rc != MJB_OK) return rc;
  *A_pinned = d->fd_A_host; *B_pinned = d->fd_B_host;
  return MJB_OK;
}



int mjb_jac(mjbData* d, int nreq, const int* kinds, const int* ids, double* jacp_host, double* jacr_host) {
  if (!d || !kinds || !ids || !jacp_host || nreq < 1) return fail(MJB_ERR_ARG, "bad argument");
  const HostModel& h = d->model->h;
  for (int i = 0; i < nreq; i++) {
    if (kinds[i] < 0 || kinds[i] > 3) return fail(MJB_ERR_ARG, "jacobian kind must be 0..3");
    int lim = kinds[i] == 0 ? h.nsite : h.nbody;
    if (ids[i] < 0 || ids[i] >= lim) return fail(MJB_ERR_LOOKUP, "jacobian object id out of range");
  }
  HIPCHK(hipSetDevice(d->device));
  { int rc0 = refresh_options(d); if (rc0 != MJB_OK) return rc0; }
  const size_t n = (size_t)d->batch * nreq * 3 * h.nv;
  if (n == 0) return MJB_OK;
  // persistent buffers instead of four hipMalloc / hipFree and four blocking copies per call (a controller with needs_jacobians calls this
  // once per Env.step): request in pinned memory read by the kernel, results pinned, device staging only for large requests
  if (nreq > d->jac_req_cap) {
    if (d->jac_req_pin) (void)hipHostFree(d->jac_req_pin);
    d->jac_req_pin = nullptr; d->jac_req_cap = 0;
    HIPCHK(hipHostMalloc((void**)&d->jac_req_pin, sizeof(int) * 2 * (size_t)nreq, hipHostMallocDefault));
    d->jac_req_cap = nreq;
  }
  if (n > d->jac_pin_cap) {
    for (int k = 0; k < 2; k++) { if (d->jac_pin[k]) (void)hipHostFree(d->jac_pin[k]); d->jac_pin[k] = nullptr; }
    d->jac_pin_cap = 0;
    for (int k = 0; k < 2; k++) HIPCHK(hipHostMalloc((void**)&d->jac_pin[k], sizeof(double) * n, hipHostMallocDefault));
    d->jac_pin_cap = n;
  }
  const bool zero_copy = 2 * n * sizeof(double) <= (size_t)256 * 1024;
  if (!zero_copy && n > d->jac_dev_cap) {
    for (int k = 0; k < 2; k++) { if (d->jac_dev[k]) (void)hipFree(d->jac_dev[k]); d->jac_dev[k] = nullptr; }
    d->jac_dev_cap = 0;
    for (int k = 0; k < 2; k++) HIPCHK(hipMalloc((void**)&d->jac_dev[k], sizeof(double) * n));
    d->jac_dev_cap = n;
  }
  HIPCHK(hipStreamSynchronize(d->stream));                       // nothing queued may still be reading the previous request
  std::memcpy(d->jac_req_pin, kinds, sizeof(int) * nreq);
  std::memcpy(d->jac_req_pin + nreq, ids, sizeof(int) * nreq);
  double *op = zero_copy ? d->jac_pin[0] : d->jac_dev[0], *orr = zero_copy ? d->jac_pin[1] : d->jac_dev[1];
  const int *dk = d->jac_req_pin, *di = d->jac_req_pin + nreq;
  hipError_t e = d->dtype == MJB_F32 ? launch_jac<double, float>(d->G_fd, d->md_dev, d->Ld_dev, d->Ld, d->df, nreq, dk, di, op, orr, d->stream)
                                     : launch_jac<double, double>(d->G_fd, d->md_dev, d->Ld_dev, d->Ld, d->dd, nreq, dk, di, op, orr, d->stream);
  if (e != hipSuccess) return fail(MJB_ERR_DEVICE, std::string("jac launch: ") + hipGetErrorString(e));
  if (!zero_copy) {
    HIPCHK(hipMemcpyAsync(d->jac_pin[0], d->jac_dev[0], n * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    if (jacr_host) HIPCHK(hipMemcpyAsync(d->jac_pin[1], d->jac_dev[1], n * sizeof(double), hipMemcpyDeviceToHost, d->stream));
  }
  if (hipStreamSynchronize(d->stream) != hipSuccess) return fail(MJB_ERR_DEVICE, "jac sync failed");
  std::memcpy(jacp_host, d->jac_pin[0], n * sizeof(double));
  if (jacr_host) std::memcpy(jacr_host, d->jac_pin[1], n * sizeof(double));
  return MJB_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// model: names, fields, binary save / load
// ---------------------------------------------------------------------------------------------------------------------
int mjb_model_name2id(const mjbModel* m, int objtype, const char* name) {
  if (!m || !name) { fail(MJB_ERR_ARG, "model/name is NULL"); return -1; }
  if (objtype == 2) objtype = 1;                                       // mjOBJ_XBODY aliases mjOBJ_BODY
  auto it = m->names.find(objtype);
  if (it == m->names.end()) return -1;
  for (size_t i = 0; i < it->second.size(); i++) if (!it->second[i].empty() && it->second[i] == name) return (int)i;
  return -1;
}

const char* mjb_model_id2name(const mjbModel* m, int objtype, int id) {
  if (!m) { fail(MJB_ERR_ARG, "model is NULL"); return nullptr; }
  if (objtype == 2) objtype = 1;
  auto it = m->names.find(objtype);
  if (it == m->names.end() || id < 0 || id >= (int)it->second.size() || it->second[id].empty()) return nullptr;
  return it->second[id].c_str();
}

int mjb_model_field(const mjbModel* m, const char* name, const void** ptr, long* count, int* dtype) {
  if (!m || !name) return fail(MJB_ERR_ARG, "model/name is NULL");
  for (const auto& f : m->fields) if (f.name == name) {
    if (ptr) *ptr = m->blob.data() + f.off;
    if (count) *count = f.count;
    if (dtype) *dtype = f.dtype;
    return MJB_OK;
  }
  return fail(MJB_ERR_ARG, std::string("unknown model field: ") + name);
}

int mjb_model_field_at(const mjbModel* m, int index, const char** name, const void** ptr, long* count, int* dtype) {
  if (!m) { fail(MJB_ERR_ARG, "model is NULL"); return -1; }
  if (index < 0 || index >= (int)m->fields.size()) return (int)m->fields.size();     // out of range: returns the field count
  const auto& f = m->fields[index];
  if (name) *name = f.name.c_str();
  if (ptr) *ptr = m->blob.data() + f.off;
  if (count) *count = f.count;
  if (dtype) *dtype = f.dtype;
  return MJB_OK;
}

int mjb_model_save(const mjbModel* m, const char* path) {
  if (!m || !path) return fail(MJB_ERR_ARG, "model/path is NULL");
  // like mj_saveModel the file carries the options as they are NOW (disableactuator / iterations / tolerance may have been edited
  // since the table was compiled): they are patched into a copy of the creation-time blob
  std::vector<char> blob = m->blob;
  for (const auto& fl : m->fields) {
    if (fl.count != 1) continue;
    if (fl.dtype == 1 && fl.name == "disableactuator") { int32_t v = m->disableactuator; std::memcpy(blob.data() + fl.off, &v, 4); }
    else if (fl.dtype == 1 && fl.name == "iterations") { int32_t v = m->iterations; std::memcpy(blob.data() + fl.off, &v, 4); }
    else if (fl.dtype == 0 && fl.name == "tolerance") { double v = m->tolerance; std::memcpy(blob.data() + fl.off, &v, 8); }
  }
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(MJB_ERR_ARG, std::string("cannot open for writing: ") + path);
  size_t n = std::fwrite(blob.data(), 1, blob.size(), f);
  int rc = std::fclose(f);
  if (n != blob.size() || rc != 0) return fail(MJB_ERR_ARG, std::string("short write: ") + path);
  return MJB_OK;
}

int mjb_model_load(const char* path, mjbModel** out) {
  if (!path || !out) return fail(MJB_ERR_ARG, "path/out is NULL");
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(MJB_ERR_ARG, std::string("cannot open: ") + path);
  std::vector<char> b;
  char buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
  std::fclose(f);
  std::vector<std::string> names; std::vector<const void*> ptrs; std::vector<int> dts; std::vector<long> cnts;
  std::string err;
  if (!parse_blob(b, names, ptrs, dts, cnts, err)) return fail(MJB_ERR_MODEL, err + ": " + path);
  std::vector<const char*> cn;
  for (auto& s : names) cn.push_back(s.c_str());
  return mjb_model_create((int)cn.size(), cn.data(), ptrs.data(), dts.data(), cnts.data(), out);
}

// ---------------------------------------------------------------------------------------------------------------------
// mj_integratePos / mj_differentiatePos on caller-owned HOST vectors, float64, batched [batch, nq] / [batch, nv]
// ---------------------------------------------------------------------------------------------------------------------
int mjb_integrate_pos(const mjbModel* m, int batch, double* qpos, const double* qvel, double dt) {
  if (!m || !qpos || !qvel || batch < 1) return fail(MJB_ERR_ARG, "bad argument");
  const HostModel& h = m->h;
  const auto& jt = h.I("jnt_type"); const auto& jq = h.I("jnt_qposadr"); const auto& jd = h.I("jnt_dofadr");
  for (int e = 0; e < batch; e++) {
    double* q = qpos + (size_t)e * h.nq; const double* v = qvel + (size_t)e * h.nv;
    for (int j = 0; j < h.njnt; j++) {
      const int qa = jq[j], da = jd[j];
      if (jt[j] == JNT_FREE) {
        for (int k = 0; k < 3; k++) q[qa + k] += dt * v[da + k];
        double ax[3] = {v[da + 3], v[da + 4], v[da + 5]};
        const double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        if (nrm < 1e-15) continue;
        const double ang = dt * nrm, sn = std::sin(0.5 * ang) / nrm;
        double qr[4] = {std::cos(0.5 * ang), ax[0] * sn, ax[1] * sn, ax[2] * sn}, out[4];
        h_quat_mul(out, q + qa + 3, qr);
        const double on = std::sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2] + out[3] * out[3]);
        if (on < 1e-15) { q[qa + 3] = 1; q[qa + 4] = q[qa + 5] = q[qa + 6] = 0; }
        else for (int k = 0; k < 4; k++) q[qa + 3 + k] = out[k] / on;
      } else q[qa] += dt * v[da];
    }
  }
  return MJB_OK;
}

int mjb_differentiate_pos(const mjbModel* m, int batch, double* qvel_out, double dt, const double* qpos1, const double* qpos2) {
  if (!m || !qvel_out || !qpos1 || !qpos2 || batch < 1) return fail(MJB_ERR_ARG, "bad argument");
  if (dt == 0) return fail(MJB_ERR_ARG, "dt must be non-zero");
  const HostModel& h = m->h;
  const auto& jt = h.I("jnt_type"); const auto& jq = h.I("jnt_qposadr"); const auto& jd = h.I("jnt_dofadr");
  const double PI = 3.14159265358979323846;
  for (int e = 0; e < batch; e++) {
    const double *q1 = qpos1 + (size_t)e * h.nq, *q2 = qpos2 + (size_t)e * h.nq; double* v = qvel_out + (size_t)e * h.nv;
    for (int j = 0; j < h.njnt; j++) {
      const int qa = jq[j], da = jd[j];
      if (jt[j] == JNT_FREE) {
        for (int k = 0; k < 3; k++) v[da + k] = (q2[qa + k] - q1[qa + k]) / dt;
        const double qn[4] = {q1[qa + 3], -q1[qa + 4], -q1[qa + 5], -q1[qa + 6]};
        double qd[4];
        h_quat_mul(qd, qn, q2 + qa + 3);
        const double sn = std::sqrt(qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
        if (sn < 1e-15) { v[da + 3] = v[da + 4] = v[da + 5] = 0; continue; }
        double ang = 2 * std::atan2(sn, qd[0]);
        if (ang > PI) ang -= 2 * PI;
        const double k = ang / sn / dt;
        v[da + 3] = qd[1] * k; v[da + 4] = qd[2] * k; v[da + 5] = qd[3] * k;
      } else v[da] = (q2[qa] - q1[qa]) / dt;
    }
  }
  return MJB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// host mirror: ONE pinned float64 block per data object, one pack kernel + ONE copy per direction
// ---------------------------------------------------------------------------------------------------------------------
static int ensure_mirror(mjbData* d) {
  if (d->mirror_host) return MJB_OK;
  const HostModel& h = d->model->h;
  const size_t B = (size_t)d->batch, n[6] = {(size_t)h.nq, (size_t)h.nv, (size_t)h.nu, (size_t)h.nv, (size_t)h.nv, 1};
  size_t o = 0;
  for (int k = 0; k < 6; k++) { d->mirror_off[k] = o; o += B * n[k]; }
  d->mirror_off[6] = o;
  HIPCHK(hipSetDevice(d->device));
  // + the engine-flags word + one completion word per environment (polled by mjb_step_host on the zero-copy path: coherent memory, the
  // device's stores must be visible to the host while the kernel is still running)
  HIPCHK(hipHostMalloc((void**)&d->mirror_host, (o + 1 + B) * sizeof(double), hipHostMallocCoherent));
  std::memset(d->mirror_host, 0, (o + 1 + B) * sizeof(double));
  if (dev_alloc(d, &d->mirror_dev, o + 1)) return fail(MJB_ERR_DEVICE, "device allocation of the mirror staging block failed");
  return MJB_OK;
}
static const char* const kMirrorNames[6] = {"qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "time"};

int mjb_host_view(mjbData* d, const char* name, double** host_ptr, long* per_env) {
  if (!d || !name || !host_ptr) return fail(MJB_ERR_ARG, "NULL argument");
  int rc = ensure_mirror(d);
  if (rc != MJB_OK) return rc;
  const HostModel& h = d->model->h;
  const long n[6] = {h.nq, h.nv, h.nu, h.nv, h.nv, 1};
  if (!std::strcmp(name, "engine_flags")) { *host_ptr = d->mirror_host + d->mirror_off[6]; if (per_env) *per_env = 0; return MJB_OK; }
  for (int k = 0; k < 6; k++) if (!std::strcmp(name, kMirrorNames[k])) {
    *host_ptr = d->mirror_host + d->mirror_off[k];
    if (per_env) *per_env = n[k];
    return MJB_OK;
  }
  return fail(MJB_ERR_ARG, std::string("no host mirror for array: ") + name);
}

// Small blocks (a few environments: the reference's batch-1 loops) skip the staging copy: the pack / unpack kernels read and write
// the pinned host block itself (device-visible, unified addressing) - two runtime calls less per host-driven step.
static bool mirror_zero_copy(const mjbData* d) { return d->mirror_off[6] + 1 <= 8192; }

static int mirror_pull(mjbData* d) {
  const HostModel& h = d->model->h;
  const long total = (long)d->mirror_off[6];
  const unsigned grid = (unsigned)((total + 255) / 256);
  double* dst = mirror_zero_copy(d) ? d->mirror_host : d->mirror_dev;
  if (d->dtype == MJB_F32) hipLaunchKernelGGL(k_mirror_pack<float>, dim3(grid), dim3(256), 0, d->stream, d->df, h.nq, h.nv, h.nu, dst, total);
  else hipLaunchKernelGGL(k_mirror_pack<double>, dim3(grid), dim3(256), 0, d->stream, d->dd, h.nq, h.nv, h.nu, dst, total);
  HIPCHK(hipGetLastError());
  if (dst == d->mirror_dev) HIPCHK(hipMemcpyAsync(d->mirror_host, d->mirror_dev, (size_t)(total + 1) * sizeof(double), hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  return engine_check(d);
}

static int mirror_push(mjbData* d, int mask) {
  mask &= 63;
  if (!mask) return MJB_OK;
  const HostModel& h = d->model->h;
  int lo = 0, hi = 5;
  while (!((mask >> lo) & 1)) lo++;
  while (!((mask >> hi) & 1)) hi--;
  const size_t a = d->mirror_off[lo], b = d->mirror_off[hi + 1];       // one contiguous span covering the edited fields
  const double* src = mirror_zero_copy(d) ? d->mirror_host : d->mirror_dev;
  if (src == d->mirror_dev) HIPCHK(hipMemcpyAsync(d->mirror_dev + a, d->mirror_host + a, (b - a) * sizeof(double), hipMemcpyHostToDevice, d->stream));
  const long total = (long)d->mirror_off[6];
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (d->dtype == MJB_F32) hipLaunchKernelGGL(k_mirror_unpack<float>, dim3(grid), dim3(256), 0, d->stream, d->df, h.nq, h.nv, h.nu, src, total, mask);
  else hipLaunchKernelGGL(k_mirror_unpack<double>, dim3(grid), dim3(256), 0, d->stream, d->dd, h.nq, h.nv, h.nu, src, total, mask);
  HIPCHK(hipGetLastError());
  return MJB_OK;
}

int mjb_sync_to_host(mjbData* d) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  int rc = ensure_mirror(d);
  if (rc != MJB_OK) return rc;
  HIPCHK(hipSetDevice(d->device));
  return mirror_pull(d);
}

int mjb_sync_to_device(mjbData* d, int field_mask) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  int rc = ensure_mirror(d);
  if (rc != MJB_OK) return rc;
  HIPCHK(hipSetDevice(d->device));
  return mirror_push(d, field_mask);
}

int mjb_step_host(mjbData* d, int nstep, int field_mask) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  if (nstep < 0) return fail(MJB_ERR_ARG, "nstep must be >= 0");
  int rc = ensure_mirror(d);
  if (rc != MJB_OK) return rc;
  HIPCHK(hipSetDevice(d->device));
  ObsSpecDev none; std::memset(&none, 0, sizeof(none));
  if (nstep > 0 && mirror_zero_copy(d) && d->dtype == MJB_F32) {
    // small batches: ONE launch - the step kernel reads the edited fields from the pinned block and writes the new state back into it
    StepArgs a = make_args(d, nstep, MJB_CTRL_KEEP, 0, 0, 1.0, 0);
    a.mirror = d->mirror_host; a.mirror_mask = field_mask & 63;
    double* flagword = d->mirror_host + d->mirror_off[6];
    // short launches: poll the environments' completion words in the pinned block instead of waiting for the end-of-kernel signal
    // (the state words are published before them; whatever else the kernel's epilogue writes is ordered by the stream for later calls)
    static const bool poll_ok = !(std::getenv("MJB_HOST_POLL") && std::atoi(std::getenv("MJB_HOST_POLL")) == 0);
    const bool poll = poll_ok && nstep <= 64;
    if (poll) a.mirror_seq = ++d->mirror_seq;
    if ((rc = launch(d, a, none, nullptr, false)) != MJB_OK) return rc;
    bool done = false;
    if (poll) {
      const unsigned long long* words = (const unsigned long long*)(flagword + 1);
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned long spins = 0;; spins++) {
        int e = 0;
        while (e < d->batch && __atomic_load_n(words + e, __ATOMIC_ACQUIRE) == a.mirror_seq) e++;
        if (e == d->batch) { done = true; break; }
        __builtin_ia32_pause();
        // a launch that has not finished after 20 ms is not the short step this path is for (or has faulted): let the runtime wait and report
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
      }
    }
    if (!done) HIPCHK(hipStreamSynchronize(d->stream));
    if (*flagword < 0) {                                       // some environment raised an engine flag: fetch the sticky word
      int fl = 0;
      if (done) HIPCHK(hipStreamSynchronize(d->stream));        // (rare) the polled return left the kernel's epilogue in flight: the copy below must not overtake it
      HIPCHK(hipMemcpy(&fl, d->df.flags, sizeof(int), hipMemcpyDeviceToHost));
      *flagword = (double)fl;
    }
    return MJB_OK;
  }
  if ((rc = mirror_push(d, field_mask)) != MJB_OK) return rc;
  if (nstep > 0 && (rc = launch(d, make_args(d, nstep, MJB_CTRL_KEEP, 0, 0, 1.0, 0), none, nullptr, false)) != MJB_OK) return rc;
  if (nstep == 0 && (rc = launch(d, make_args(d, 1, MJB_CTRL_KEEP, 0, 0, 1.0, 1), none, nullptr, false)) != MJB_OK) return rc;   // mj_forward
  return mirror_pull(d);
}

static int ensure_shadow(mjbData* d) {
  int rc = ensure_mirror(d);
  if (rc != MJB_OK) return rc;
  if (!d->mirror_shadow) {
    d->mirror_shadow = (double*)std::malloc(d->mirror_off[6] * sizeof(double));
    if (!d->mirror_shadow) return fail(MJB_ERR_ARG, "out of host memory for the mirror shadow");
    std::memcpy(d->mirror_shadow, d->mirror_host, d->mirror_off[6] * sizeof(double));
  }
  return MJB_OK;
}

int mjb_mirror_edited_mask(mjbData* d, int* mask_out) {
  if (!d || !mask_out) return fail(MJB_ERR_ARG, "NULL argument");
  int rc = ensure_shadow(d);
  if (rc != MJB_OK) return rc;
  int mask = 0;
  for (int k = 0; k < 6; k++) {
    const size_t a = d->mirror_off[k], b = d->mirror_off[k + 1];
    if (b > a && std::memcmp(d->mirror_host + a, d->mirror_shadow + a, (b - a) * sizeof(double)) != 0) mask |= 1 << k;   // bitwise: -0.0 / NaN payload edits count
  }
  *mask_out = mask;
  return MJB_OK;
}

int mjb_mirror_commit(mjbData* d, int field_mask) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  int rc = ensure_shadow(d);
  if (rc != MJB_OK) return rc;
  for (int k = 0; k < 6; k++) if ((field_mask >> k) & 1) {
    const size_t a = d->mirror_off[k], b = d->mirror_off[k + 1];
    std::memcpy(d->mirror_shadow + a, d->mirror_host + a, (b - a) * sizeof(double));
  }
  return MJB_OK;
}

int mjb_step_host_auto(mjbData* d, int nstep, int compare, int* mask_out) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  int mask = 0, rc;
  if (compare && (rc = mjb_mirror_edited_mask(d, &mask)) != MJB_OK) return rc;
  if (mask_out) *mask_out = mask;
  if ((rc = mjb_step_host(d, nstep, mask)) != MJB_OK) return rc;
  return mjb_mirror_commit(d, 63);
}

// ---------------------------------------------------------------------------------------------------------------------
// observation all-gather over RCCL for a host that owns an ncclComm_t (SURVEY.md §8(b) "allgather_obs", §8(e)): the ONE
// collective of the path.  The library does not link RCCL: the symbol is taken from the RCCL the process already loaded
// (torch's bundled one, or the host's), else from librccl.so.1.
// ---------------------------------------------------------------------------------------------------------------------
int mjb_allgather_obs(void* nccl_comm, const void* send_dev, void* recv_dev, long count_per_rank, int dtype, void* hip_stream) {
  if (!nccl_comm || !send_dev || !recv_dev || count_per_rank < 0) return fail(MJB_ERR_ARG, "mjb_allgather_obs: NULL / negative argument");
  if (dtype != MJB_F32 && dtype != MJB_F64) return fail(MJB_ERR_ARG, "dtype must be MJB_F32 or MJB_F64");
  typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
  static allgather_fn fn = nullptr;
  if (!fn) {
    fn = (allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (!fn) {
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (h) fn = (allgather_fn)dlsym(h, "ncclAllGather");
    }
    if (!fn) return fail(MJB_ERR_DEVICE, "ncclAllGather not found: no RCCL in this process and librccl.so.1 is not loadable");
  }
  const int nccl_dtype = dtype == MJB_F32 ? 7 /* ncclFloat32 */ : 8 /* ncclFloat64 */;
  const int rc = fn(send_dev, recv_dev, (size_t)count_per_rank, nccl_dtype, nccl_comm, (hipStream_t)hip_stream);
  if (rc != 0) return fail(MJB_ERR_DEVICE, "ncclAllGather failed with ncclResult " + std::to_string(rc));
  return MJB_OK;
}

int mjb_profile_get(mjbData* d, unsigned long long* host_out /* [24] per-phase cycle sums; zero unless built with -DMJB_PROFILE */) {
  if (!d || !host_out) return fail(MJB_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  void* p = d->dtype == MJB_F32 ? (void*)d->df.prof : (void*)d->dd.prof;
  HIPCHK(hipMemcpy(host_out, p, sizeof(unsigned long long) * PH_N, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(p, 0, sizeof(unsigned long long) * PH_N));
  return MJB_OK;
}

int mjb_step_schedule(mjbData* d, int* out6) {
  if (!d || !out6) return fail(MJB_ERR_ARG, "NULL argument");
  for (int i = 0; i < 6; i++) out6[i] = d->last_sched[i];
  return MJB_OK;
}

int mjb_profile_env_get(mjbData* d, unsigned long long* host_out /* [batch, 4]: start, end (100 MHz), HW_ID, XCC_ID; zeros unless the kernel was built with -DMJB_TIMELINE */) {
  if (!d || !host_out) return fail(MJB_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipStreamSynchronize(d->stream));
  unsigned long long* p = (d->dtype == MJB_F32 ? d->df.prof : d->dd.prof) + PH_N;
  HIPCHK(hipMemcpy(host_out, p, sizeof(unsigned long long) * 4 * (size_t)d->batch, hipMemcpyDeviceToHost));
  return MJB_OK;
}

int mjb_debug_forward(mjbData* d) {
  if (!d) return fail(MJB_ERR_ARG, "data is NULL");
  HIPCHK(hipSetDevice(d->device));
  if (!d->dbg_ready) {
    std::memset(&d->dbgf, 0, sizeof(d->dbgf)); std::memset(&d->dbgd, 0, sizeof(d->dbgd));
    int rc = d->dtype == MJB_F32 ? alloc_debug(d, d->dbgf) : alloc_debug(d, d->dbgd);
    if (rc) return fail(MJB_ERR_DEVICE, "device allocation of debug buffers failed");
    d->dbg_ready = true;
  }
  ObsSpecDev none; std::memset(&none, 0, sizeof(none));
  return launch(d, make_args(d, 1, MJB_CTRL_KEEP, 0, 0, 1.0, 1), none, nullptr, true);
}

int mjb_debug_get(mjbData* d, const char* name, void* host_out, long capacity_elems) {
  if (!d || !name || !host_out) return fail(MJB_ERR_ARG, "NULL argument");
  if (!d->dbg_ready) return fail(MJB_ERR_ARG, "call mjb_debug_forward first");
  auto it = d->dbg_arrays.find(name);
  if (it == d->dbg_arrays.end()) return fail(MJB_ERR_ARG, std::string("unknown debug array: ") + name);
  const ArrayInfo& ai = it->second;
  long n = (long)d->batch * ai.per_env;
  if (capacity_elems < n) return fail(MJB_ERR_ARG, "output buffer too small");
  HIPCHK(hipSetDevice(d->device));
  if (ai.kind == 2) {
    HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipMemcpy(host_out, ai.ptr, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return MJB_OK;
  }
  return copy_out(d, ai, (double*)host_out);
}

}  // extern "C"
