// TEST-ONLY: runs mjb_device.hpp's env_run<> for ONE environment on the host with one
// std::thread per lane (mjb_hostemu.hpp).  Used by tests/test_kernel_hostemu.py to check
// the kernel logic against the oracle without a GPU.  Not part of the product library.
#define MJB_HOST_EMU 1
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../mujoco_template_amd/csrc/mjb_device.hpp"
#include "../../mujoco_template_amd/csrc/mjb_host.hpp"

using namespace mjb;

namespace {
std::string g_err;

template <typename T> struct HostAlloc {
  std::vector<std::vector<T>> f; std::vector<std::vector<int>> i; std::vector<std::vector<unsigned long long>> u;
  const T* putf(const std::vector<T>& v) { f.push_back(v); if (f.back().empty()) f.back().resize(1); return f.back().data(); }
  const int* puti(const std::vector<int>& v) { i.push_back(v); if (i.back().empty()) i.back().resize(1); return i.back().data(); }
  const unsigned long long* putu(const std::vector<unsigned long long>& v) { u.push_back(v); return u.back().data(); }
};

template <typename T, int G>
void run_group(const DevModel<T>& m, const Lay& L, const DevData<double>& d, const DevDebug<double>& dbg, const StepArgs& a,
               const ObsSpecDev& obs, double* obs_out) {
  std::vector<char> lds((size_t)L.bytes + 64, 0);
  T* w = (T*)lds.data();
  int* wi = (int*)(w + L.nT);
  emu::Group grp(G);
  std::vector<std::thread> th;
  for (int lane = 0; lane < G; lane++) {
    th.emplace_back([&, lane]() {
      emu::tl_group = &grp; emu::tl_lane = lane;
      env_run<T, double, G>(&m, &L, d, dbg, a, obs, obs_out, w, wi, 0, lane, 0, a.nstep, 0u);
    });
  }
  for (auto& t : th) t.join();
}

template <typename T>
int run_typed(const HostModel& h, int G, int ncon_max, int nefc_max, const DevData<double>& d, const DevDebug<double>& dbg,
              const StepArgs& a, const ObsSpecDev& obs, double* obs_out) {
  HostAlloc<T> alloc;
  DevModel<T> m;
  fill_dev_model<T>(h, alloc, ncon_max, nefc_max, m);
  Lay L = make_layout(h, ncon_max, nefc_max, sizeof(T));
  switch (G) {
    case 1: run_group<T, 1>(m, L, d, dbg, a, obs, obs_out); break;
    case 8: run_group<T, 8>(m, L, d, dbg, a, obs, obs_out); break;
    case 16: run_group<T, 16>(m, L, d, dbg, a, obs, obs_out); break;
    case 64: run_group<T, 64>(m, L, d, dbg, a, obs, obs_out); break;
    default: g_err = "unsupported G"; return -1;
  }
  return 0;
}
}  // namespace

extern "C" {
const char* mjbemu_last_error() { return g_err.c_str(); }

// io table: named float64/int32 arrays for state (qpos,qvel,ctrl,qacc,qacc_warmstart,time,counters),
// kinematic outputs and optional debug dumps; absent names are simply not written.
int mjbemu_run(int nfield, const char* const* names, const void* const* ptrs, const int* dtypes, const long* counts,
               int nio, const char* const* io_names, void* const* io_ptrs,
               int G, int use_double, int ncon_max, int nefc_max,
               int nstep, int ctrl_mode, unsigned seed, unsigned step0, unsigned env0, float scale, int mode) {
  Table t{nfield, names, ptrs, dtypes, counts};
  HostModel h;
  if (!h.load(t, g_err)) return -1;
  if (ncon_max <= 0) ncon_max = h.ncon_alloc;
  if (nefc_max <= 0) nefc_max = h.nefc_alloc;
  auto io = [&](const char* k) -> void* { for (int i = 0; i < nio; i++) if (!std::strcmp(io_names[i], k)) return io_ptrs[i]; return nullptr; };
  DevData<double> d;
  std::memset(&d, 0, sizeof(d));
  d.batch = 1;
  d.qpos = (double*)io("qpos"); d.qvel = (double*)io("qvel"); d.ctrl = (double*)io("ctrl"); d.qacc = (double*)io("qacc");
  d.qacc_warmstart = (double*)io("qacc_warmstart"); d.time = (double*)io("time"); d.counters = (int*)io("counters");
  d.xpos = (double*)io("xpos"); d.xquat = (double*)io("xquat"); d.xipos = (double*)io("xipos"); d.site_xpos = (double*)io("site_xpos");
  d.geom_xpos = (double*)io("geom_xpos"); d.subtree_com = (double*)io("subtree_com"); d.sensordata = (double*)io("sensordata");
  d.qfrc_inverse = (double*)io("qfrc_inverse"); d.actuator_moment = (double*)io("actuator_moment");
  if (mode == 2 && (!d.qfrc_inverse || !d.actuator_moment)) { g_err = "inverse mode needs qfrc_inverse and actuator_moment"; return -1; }
  if (!d.qpos || !d.qvel || !d.ctrl || !d.qacc || !d.qacc_warmstart || !d.time || !d.counters) { g_err = "missing state arrays"; return -1; }
  DevDebug<double> dbg;
  std::memset(&dbg, 0, sizeof(dbg));
  dbg.qM = (double*)io("qM"); dbg.qfrc_bias = (double*)io("qfrc_bias"); dbg.qfrc_passive = (double*)io("qfrc_passive");
  dbg.qfrc_actuator = (double*)io("qfrc_actuator"); dbg.qacc_smooth = (double*)io("qacc_smooth"); dbg.qfrc_constraint = (double*)io("qfrc_constraint");
  dbg.efc_J = (double*)io("efc_J"); dbg.efc_aref = (double*)io("efc_aref"); dbg.efc_D = (double*)io("efc_D"); dbg.efc_pos = (double*)io("efc_pos");
  dbg.efc_force = (double*)io("efc_force"); dbg.efc_type = (int*)io("efc_type"); dbg.con = (double*)io("con");
  dbg.cdof = (double*)io("cdof"); dbg.cinert = (double*)io("cinert"); dbg.cvel = (double*)io("cvel");
  StepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.nstep = nstep; a.ctrl_mode = ctrl_mode; a.seed = seed; a.step0 = step0; a.env0 = env0; a.ctrl_scale = scale; a.dt = h.timestep; a.mode = mode;
  a.write_kin = d.xpos != nullptr; a.obs_every = 0;
  ObsSpecDev obs;
  std::memset(&obs, 0, sizeof(obs));
  return use_double ? run_typed<double>(h, G, ncon_max, nefc_max, d, dbg, a, obs, nullptr)
                    : run_typed<float>(h, G, ncon_max, nefc_max, d, dbg, a, obs, nullptr);
}
}
