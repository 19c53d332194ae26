// mjb_host.hpp — host-side model construction shared by the HIP library and the
// (test-only) host emulation: named-array table -> DevModel<T> + derived tables
// (tree levels, child lists, dof masks) + the per-environment LDS layout.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "mjb_types.hpp"

namespace mjb {

struct Table {
  int n;
  const char* const* names;
  const void* const* ptrs;
  const int* dtypes;   // 0 f64, 1 i32
  const long* counts;
  int find(const char* name) const {
    for (int i = 0; i < n; i++) if (!std::strcmp(names[i], name)) return i;
    return -1;
  }
  bool getd(const char* name, long count, std::vector<double>& out, std::string& err) const {
    int i = find(name);
    if (i < 0 || dtypes[i] != 0 || counts[i] != count) {
      err = std::string("model field '") + name + "': missing or wrong dtype/size (want f64 x " + std::to_string(count) + ")";
      return false;
    }
    out.assign((const double*)ptrs[i], (const double*)ptrs[i] + count);
    return true;
  }
  bool geti(const char* name, long count, std::vector<int>& out, std::string& err) const {
    int i = find(name);
    if (i < 0 || dtypes[i] != 1 || counts[i] != count) {
      err = std::string("model field '") + name + "': missing or wrong dtype/size (want i32 x " + std::to_string(count) + ")";
      return false;
    }
    out.assign((const int*)ptrs[i], (const int*)ptrs[i] + count);
    return true;
  }
};

// Host copy of everything (float64 / int32), plus derived tables.
struct HostModel {
  int nq = 0, nv = 0, nu = 0, nbody = 0, njnt = 0, ngeom = 0, nsite = 0, ntendon = 0, nwrap = 0, nsensor = 0, nsensordata = 0, nkey = 0, npair = 0;
  int integrator = 0, disableactuator = 0, iterations = 100, nlevel = 0, has_damping = 0, has_fluid = 0, has_accel = 0, nvp = 1, nvshift = 0;
  int ncon_alloc = 0, nefc_alloc = 0;
  double timestep = 0, gravity[3] = {0, 0, 0}, density = 0, viscosity = 0, tolerance = 1e-8, meaninertia = 1;
  std::vector<std::pair<std::string, std::vector<double>>> fd;
  std::vector<std::pair<std::string, std::vector<int>>> id;
  std::vector<int> level_adr, level_body, child_adr, child_id, tri_tab, dofact_adr, dofact_act, siteact, mpair, body_round;
  int nround = 0, nround_inner = 0, max_nsub = 0, dfs_ok = 0;
  std::vector<int> body_nsub, body_anc;
  std::vector<unsigned long long> body_dofmask, dof_ancmask;

  const std::vector<double>& D(const char* k) const {
    for (auto& p : fd) if (p.first == k) return p.second;
    static std::vector<double> empty; return empty;
  }
  const std::vector<int>& I(const char* k) const {
    for (auto& p : id) if (p.first == k) return p.second;
    static std::vector<int> empty; return empty;
  }

  bool load(const Table& t, std::string& err) {
    std::vector<int> iv; std::vector<double> dv;
#define ISC(f) do { if (!t.geti(#f, 1, iv, err)) return false; f = iv[0]; } while (0)
#define DSC(f) do { if (!t.getd(#f, 1, dv, err)) return false; f = dv[0]; } while (0)
    ISC(nq); ISC(nv); ISC(nu); ISC(nbody); ISC(njnt); ISC(ngeom); ISC(nsite); ISC(ntendon); ISC(nwrap); ISC(nsensor);
    ISC(nsensordata); ISC(nkey); ISC(npair); ISC(integrator); ISC(disableactuator); ISC(iterations);
    DSC(timestep); DSC(density); DSC(viscosity); DSC(tolerance); DSC(meaninertia);
#undef ISC
#undef DSC
    if (!t.getd("gravity", 3, dv, err)) return false;
    gravity[0] = dv[0]; gravity[1] = dv[1]; gravity[2] = dv[2];
    if (nv > 64) { err = "nv > 64 is outside the supported subset (dof masks are 64-bit)"; return false; }
    if (npair > 65535) { err = "more than 65535 collision pairs is outside the supported subset"; return false; }
    struct FD { const char* k; long c; };
    const FD fds[] = {
      {"body_pos", nbody * 3L}, {"body_quat", nbody * 4L}, {"body_ipos", nbody * 3L}, {"body_iquat", nbody * 4L}, {"body_mass", nbody},
      {"body_inertia", nbody * 3L}, {"body_subtreemass", nbody}, {"body_invweight0", nbody * 2L},
      {"jnt_pos", njnt * 3L}, {"jnt_axis", njnt * 3L}, {"jnt_range", njnt * 2L}, {"jnt_stiffness", njnt}, {"jnt_margin", njnt},
      {"jnt_solref", njnt * 2L}, {"jnt_solimp", njnt * 5L}, {"qpos0", nq}, {"qpos_spring", nq},
      {"dof_armature", nv}, {"dof_damping", nv}, {"dof_invweight0", nv},
      {"geom_pos", ngeom * 3L}, {"geom_quat", ngeom * 4L}, {"geom_size", ngeom * 3L}, {"site_pos", nsite * 3L}, {"site_quat", nsite * 4L},
      {"tendon_range", ntendon * 2L}, {"tendon_margin", ntendon}, {"tendon_solref", ntendon * 2L}, {"tendon_solimp", ntendon * 5L},
      {"tendon_invweight0", ntendon}, {"wrap_prm", nwrap},
      {"actuator_gear", nu * 6L}, {"actuator_gainprm", nu * 3L}, {"actuator_biasprm", nu * 3L}, {"actuator_ctrlrange", nu * 2L}, {"actuator_forcerange", nu * 2L},
      {"pair_friction", npair * 5L}, {"pair_solref", npair * 2L}, {"pair_solimp", npair * 5L}, {"pair_margin", npair}, {"pair_gap", npair},
      {"key_qpos", (long)nkey * nq}, {"key_qvel", (long)nkey * nv}, {"key_ctrl", (long)nkey * nu}, {"key_time", nkey}};
    for (auto& f : fds) { std::vector<double> v; if (!t.getd(f.k, f.c, v, err)) return false; fd.emplace_back(f.k, std::move(v)); }
    const FD ids[] = {
      {"body_parentid", nbody}, {"body_rootid", nbody}, {"body_weldid", nbody}, {"body_jntadr", nbody}, {"body_jntnum", nbody}, {"body_dofadr", nbody}, {"body_dofnum", nbody},
      {"jnt_type", njnt}, {"jnt_qposadr", njnt}, {"jnt_dofadr", njnt}, {"jnt_bodyid", njnt}, {"jnt_limited", njnt},
      {"dof_bodyid", nv}, {"dof_jntid", nv}, {"dof_parentid", nv}, {"geom_type", ngeom}, {"geom_bodyid", ngeom}, {"site_bodyid", nsite},
      {"tendon_adr", ntendon}, {"tendon_num", ntendon}, {"tendon_limited", ntendon}, {"wrap_objid", nwrap},
      {"actuator_trntype", nu}, {"actuator_trnid", nu * 2L}, {"actuator_biastype", nu}, {"actuator_ctrllimited", nu}, {"actuator_forcelimited", nu}, {"actuator_group", nu},
      {"sensor_type", nsensor}, {"sensor_objid", nsensor}, {"sensor_adr", nsensor},
      {"pair_geom1", npair}, {"pair_geom2", npair}, {"pair_condim", npair}, {"body_depth", nbody}};
    for (auto& f : ids) { std::vector<int> v; if (!t.geti(f.k, f.c, v, err)) return false; id.emplace_back(f.k, std::move(v)); }
    derive();
    return true;
  }

  void derive() {
    const auto& parent = I("body_parentid");
    const auto& depth = I("body_depth");
    int maxd = 0;
    for (int b = 1; b < nbody; b++) if (depth[b] > maxd) maxd = depth[b];
    nlevel = maxd;
    level_adr.assign(nlevel + 1, 0);
    level_body.clear();
    for (int lev = 1; lev <= maxd; lev++) {
      level_adr[lev - 1] = (int)level_body.size();
      for (int b = 1; b < nbody; b++) if (depth[b] == lev) level_body.push_back(b);
    }
    level_adr[nlevel] = (int)level_body.size();
    // schedule of the subtree sums (tree_backward_sum): deepest level first, one sibling rank per round; the rounds of
    // the depth-1 bodies (which add into the world body) come last
    {
      std::vector<int> rank(nbody, 0), nrank(maxd + 2, 0), base(maxd + 2, 0);
      for (int b = 1; b < nbody; b++) {
        int r = 0;
        for (int o = 1; o < b; o++) if (parent[o] == parent[b]) r++;
        rank[b] = r;
        if (r + 1 > nrank[depth[b]]) nrank[depth[b]] = r + 1;
      }
      int acc = 0;
      for (int lev = maxd; lev >= 1; lev--) { if (lev == 1) nround_inner = acc; base[lev] = acc; acc += nrank[lev]; }
      nround = acc;
      body_round.assign(nbody, -1);
      for (int b = 1; b < nbody; b++) body_round[b] = base[depth[b]] + rank[b];
    }
    // flat tree sums: descendants count (and whether the descendants of every body are the id range right behind it, i.e.
    // the bodies are in depth-first order) and the table of proper ancestors (parent, grandparent, ...; world excluded)
    {
      body_nsub.assign(nbody, 0);
      for (int b = nbody - 1; b >= 1; b--) body_nsub[parent[b]] += body_nsub[b] + 1;
      dfs_ok = 1; max_nsub = 0;
      for (int b = 0; b < nbody; b++) {
        if (b >= 1 && body_nsub[b] > max_nsub) max_nsub = body_nsub[b];
        for (int d = b + 1; d <= b + body_nsub[b] && d < nbody; d++) {          // every id in the range must descend from b
          int a = d; while (a != b && a != 0) a = parent[a];
          if (a != b) dfs_ok = 0;
        }
        if (b + body_nsub[b] >= nbody) dfs_ok = 0;
      }
      if (body_nsub[0] > max_nsub) max_nsub = body_nsub[0];
      const int nl = nlevel > 0 ? nlevel : 1;
      body_anc.assign((size_t)nbody * nl, 0);
      for (int b = 0; b < nbody; b++) {
        int a = b, u = 0;
        for (; u < nl; u++) { if (a == 0 || parent[a] == 0) break; a = parent[a]; body_anc[(size_t)b * nl + u] = a; }
        for (; u < nl; u++) body_anc[(size_t)b * nl + u] = b;                   // padding (masked on the device)
      }
    }
    child_adr.assign(nbody + 1, 0);
    child_id.clear();
    for (int b = 0; b < nbody; b++) {
      child_adr[b] = (int)child_id.size();
      for (int ch = 1; ch < nbody; ch++) if (parent[ch] == b) child_id.push_back(ch);
    }
    child_adr[nbody] = (int)child_id.size();
    const auto& dofadr = I("body_dofadr"); const auto& dofnum = I("body_dofnum"); const auto& dofpar = I("dof_parentid");
    body_dofmask.assign(nbody > 0 ? nbody : 1, 0ull);
    for (int b = 1; b < nbody; b++) {
      int p = b;
      while (p > 0) {
        for (int k = 0; k < dofnum[p]; k++) body_dofmask[b] |= 1ull << (dofadr[p] + k);
        p = parent[p];
      }
    }
    dof_ancmask.assign(nv > 0 ? nv : 1, 0ull);
    for (int i = 0; i < nv; i++) for (int j = i; j >= 0; j = dofpar[j]) dof_ancmask[i] |= 1ull << j;
    tri_tab.clear();
    for (int r = 0; r < (nv > 0 ? nv : 1); r++) for (int cc = 0; cc <= r; cc++) tri_tab.push_back((r << 16) | cc);
    {  // actuators grouped by the dof they drive (joint transmission) / list of site transmissions
      const auto& trntype = I("actuator_trntype"); const auto& trnid = I("actuator_trnid"); const auto& jdof = I("jnt_dofadr");
      dofact_adr.assign(nv + 1, 0); dofact_act.clear(); siteact.clear();
      for (int i = 0; i < nv; i++) {
        dofact_adr[i] = (int)dofact_act.size();
        for (int a = 0; a < nu; a++) if (trntype[a] == TRN_JOINT && jdof[trnid[2 * a]] == i) dofact_act.push_back(a);
      }
      dofact_adr[nv] = (int)dofact_act.size();
      for (int a = 0; a < nu; a++) if (trntype[a] != TRN_JOINT) siteact.push_back(a);
      mpair.clear();
      for (int i = 0; i < nv; i++) for (int j = i; j >= 0; j = dofpar[j]) mpair.push_back((i << 8) | j);
    }
    has_damping = 0;
    for (double v : D("dof_damping")) if (v > 0) has_damping = 1;
    has_fluid = (density > 0 || viscosity > 0) ? 1 : 0;
    has_accel = 0;
    for (int t : I("sensor_type")) if (t == SENS_ACCEL) has_accel = 1;
    nvp = 1; nvshift = 0;
    while (nvp < nv) { nvp <<= 1; nvshift++; }
    const auto& g1 = I("pair_geom1"); const auto& g2 = I("pair_geom2"); const auto& gt = I("geom_type");
    int nc = 0;
    for (int p = 0; p < npair; p++) {
      int t1 = gt[g1[p]], t2 = gt[g2[p]];
      nc += ((t1 == G_PLANE && t2 == G_CAPSULE) || (t1 == G_CAPSULE && t2 == G_CAPSULE)) ? 2 : (t1 == G_PLANE && t2 == G_BOX) ? 4 : 1;
    }
    ncon_alloc = nc;
    nefc_alloc = 2 * njnt + 2 * ntendon + 4 * nc;
  }
};

// Fill DevModel<T> through an allocator that returns pointers valid where the kernels run.
//   alloc.putf(const std::vector<T>&) -> const T*;  alloc.puti(...) -> const int*;  alloc.putu(...) -> const unsigned long long*
template <typename T, typename Alloc>
void fill_dev_model(const HostModel& h, Alloc& alloc, int ncon_max, int nefc_max, DevModel<T>& m) {
  std::memset((void*)&m, 0, sizeof(m));
  m.nq = h.nq; m.nv = h.nv; m.nu = h.nu; m.nbody = h.nbody; m.njnt = h.njnt; m.ngeom = h.ngeom; m.nsite = h.nsite;
  m.ntendon = h.ntendon; m.nwrap = h.nwrap; m.nsensor = h.nsensor; m.nsensordata = h.nsensordata; m.nkey = h.nkey; m.npair = h.npair;
  m.nlevel = h.nlevel; m.integrator = h.integrator; m.disableactuator = h.disableactuator; m.iterations = h.iterations;
  m.has_damping = h.has_damping; m.has_fluid = h.has_fluid; m.has_accel = h.has_accel; m.nvp = h.nvp; m.nvshift = h.nvshift;
  m.ncon_max = ncon_max; m.nefc_max = nefc_max; m.nsiteact = (int)h.siteact.size(); m.nmpair = (int)h.mpair.size();
  m.timestep = (T)h.timestep; m.gravity[0] = (T)h.gravity[0]; m.gravity[1] = (T)h.gravity[1]; m.gravity[2] = (T)h.gravity[2];
  m.density = (T)h.density; m.viscosity = (T)h.viscosity; m.meaninertia = (T)h.meaninertia;
  // fp32 cannot resolve MuJoCo's 1e-8 scaled tolerance; floor it at what single precision supports
  m.tolerance = (T)h.tolerance;
  if (sizeof(T) == 4 && m.tolerance < (T)1e-6) m.tolerance = (T)1e-6;
  typedef typename DevModel<T>::FP FP;
  typedef typename DevModel<T>::IP IP;
  typedef typename DevModel<T>::UP UP;
  auto F = [&](const char* k) -> FP {
    const auto& v = h.D(k);
    std::vector<T> tv(v.begin(), v.end());
    return (FP)alloc.putf(tv);
  };
  auto Iq = [&](const char* k) -> IP { return (IP)alloc.puti(h.I(k)); };
  m.body_parentid = Iq("body_parentid"); m.body_depth = Iq("body_depth"); m.body_round = (IP)alloc.puti(h.body_round); m.nround = h.nround; m.nround_inner = h.nround_inner;
  m.body_nsub = (IP)alloc.puti(h.body_nsub); m.body_anc = (IP)alloc.puti(h.body_anc); m.max_nsub = h.max_nsub; m.dfs_ok = h.dfs_ok; m.body_rootid = Iq("body_rootid"); m.body_jntadr = Iq("body_jntadr"); m.body_jntnum = Iq("body_jntnum");
  m.body_dofadr = Iq("body_dofadr"); m.body_dofnum = Iq("body_dofnum");
  m.level_adr = (IP)alloc.puti(h.level_adr); m.level_body = (IP)alloc.puti(h.level_body); m.child_adr = (IP)alloc.puti(h.child_adr); m.child_id = (IP)alloc.puti(h.child_id); m.tri_tab = (IP)alloc.puti(h.tri_tab);
  m.dofact_adr = (IP)alloc.puti(h.dofact_adr); m.dofact_act = (IP)alloc.puti(h.dofact_act); m.siteact = (IP)alloc.puti(h.siteact); m.mpair = (IP)alloc.puti(h.mpair);
  m.body_pos = F("body_pos"); m.body_quat = F("body_quat"); m.body_ipos = F("body_ipos"); m.body_iquat = F("body_iquat"); m.body_mass = F("body_mass");
  m.body_inertia = F("body_inertia"); m.body_subtreemass = F("body_subtreemass"); m.body_invweight0 = F("body_invweight0");
  m.body_dofmask = (UP)alloc.putu(h.body_dofmask); m.dof_ancmask = (UP)alloc.putu(h.dof_ancmask);
  m.jnt_type = Iq("jnt_type"); m.jnt_qposadr = Iq("jnt_qposadr"); m.jnt_dofadr = Iq("jnt_dofadr"); m.jnt_bodyid = Iq("jnt_bodyid"); m.jnt_limited = Iq("jnt_limited");
  m.jnt_pos = F("jnt_pos"); m.jnt_axis = F("jnt_axis"); m.jnt_range = F("jnt_range"); m.jnt_stiffness = F("jnt_stiffness"); m.jnt_margin = F("jnt_margin");
  m.jnt_solref = F("jnt_solref"); m.jnt_solimp = F("jnt_solimp"); m.qpos0 = F("qpos0"); m.qpos_spring = F("qpos_spring");
  m.dof_bodyid = Iq("dof_bodyid"); m.dof_jntid = Iq("dof_jntid"); m.dof_parentid = Iq("dof_parentid");
  m.dof_armature = F("dof_armature"); m.dof_damping = F("dof_damping"); m.dof_invweight0 = F("dof_invweight0");
  m.geom_type = Iq("geom_type"); m.geom_bodyid = Iq("geom_bodyid"); m.geom_pos = F("geom_pos"); m.geom_quat = F("geom_quat"); m.geom_size = F("geom_size");
  m.site_bodyid = Iq("site_bodyid"); m.site_pos = F("site_pos"); m.site_quat = F("site_quat");
  m.tendon_adr = Iq("tendon_adr"); m.tendon_num = Iq("tendon_num"); m.tendon_limited = Iq("tendon_limited"); m.wrap_objid = Iq("wrap_objid");
  m.tendon_range = F("tendon_range"); m.tendon_margin = F("tendon_margin"); m.tendon_solref = F("tendon_solref"); m.tendon_solimp = F("tendon_solimp");
  m.tendon_invweight0 = F("tendon_invweight0"); m.wrap_prm = F("wrap_prm");
  m.actuator_trntype = Iq("actuator_trntype"); m.actuator_trnid = Iq("actuator_trnid"); m.actuator_biastype = Iq("actuator_biastype");
  m.actuator_ctrllimited = Iq("actuator_ctrllimited"); m.actuator_forcelimited = Iq("actuator_forcelimited"); m.actuator_group = Iq("actuator_group");
  m.actuator_gear = F("actuator_gear"); m.actuator_gainprm = F("actuator_gainprm"); m.actuator_biasprm = F("actuator_biasprm");
  m.actuator_ctrlrange = F("actuator_ctrlrange"); m.actuator_forcerange = F("actuator_forcerange");
  m.sensor_type = Iq("sensor_type"); m.sensor_objid = Iq("sensor_objid"); m.sensor_adr = Iq("sensor_adr");
  m.pair_geom1 = Iq("pair_geom1"); m.pair_geom2 = Iq("pair_geom2"); m.pair_condim = Iq("pair_condim");
  m.pair_friction = F("pair_friction"); m.pair_solref = F("pair_solref"); m.pair_solimp = F("pair_solimp"); m.pair_margin = F("pair_margin"); m.pair_gap = F("pair_gap");
  {
    // constraint-row constants folded on the host: K, B of a row depend only on solref, solimp[1] and the timestep
    const double MINIMP = 0.0001, MAXIMP = 0.9999, MINVAL = 1e-15;
    auto KB = [&](const double* solref, double dmax_raw, double& K, double& B) {
      const double dmax = std::min(std::max(dmax_raw, MINIMP), MAXIMP);
      if (solref[0] > 0) {
        const double tc = std::max(solref[0], 2 * h.timestep), dr = solref[1];
        K = 1 / std::max(MINVAL, dmax * dmax * tc * tc * dr * dr);
        B = 2 / std::max(MINVAL, dmax * tc);
      } else {
        K = -solref[0] / std::max(MINVAL, dmax * dmax);
        B = -solref[1] / std::max(MINVAL, dmax);
      }
    };
    const auto& g1 = h.I("pair_geom1"); const auto& g2 = h.I("pair_geom2"); const auto& gb = h.I("geom_bodyid");
    const auto& pm = h.D("pair_margin"); const auto& pg = h.D("pair_gap"); const auto& psr = h.D("pair_solref"); const auto& psi = h.D("pair_solimp");
    const auto& biw = h.D("body_invweight0");
    std::vector<T> kb((size_t)h.npair * 4);
    for (int p = 0; p < h.npair; p++) {
      double K, B;
      KB(&psr[2 * p], psi[5 * p + 1], K, B);
      kb[4 * p] = (T)(pm[p] - pg[p]);
      kb[4 * p + 1] = (T)(biw[2 * gb[g1[p]]] + biw[2 * gb[g2[p]]]);
      kb[4 * p + 2] = (T)K; kb[4 * p + 3] = (T)B;
    }
    m.pair_kb = (FP)alloc.putf(kb);
    std::vector<int> pb((size_t)h.npair * 2);
    for (int p = 0; p < h.npair; p++) { pb[2 * p] = gb[g1[p]]; pb[2 * p + 1] = gb[g2[p]]; }
    m.pair_body = (IP)alloc.puti(pb);
    const int nobj = h.njnt + h.ntendon;
    std::vector<T> lf((size_t)nobj * 12, (T)0);
    std::vector<int> li((size_t)nobj * 2, 0);
    const auto& jl = h.I("jnt_limited"); const auto& jt = h.I("jnt_type"); const auto& jq = h.I("jnt_qposadr"); const auto& jd = h.I("jnt_dofadr");
    const auto& jr = h.D("jnt_range"); const auto& jm = h.D("jnt_margin"); const auto& jsr = h.D("jnt_solref"); const auto& jsi = h.D("jnt_solimp");
    const auto& diw = h.D("dof_invweight0");
    for (int o = 0; o < h.njnt; o++) {
      double K, B;
      KB(&jsr[2 * o], jsi[5 * o + 1], K, B);
      T* r = &lf[(size_t)o * 12];
      r[0] = (T)jr[2 * o]; r[1] = (T)jr[2 * o + 1]; r[2] = (T)jm[o]; r[3] = (T)diw[jd[o]]; r[4] = (T)K; r[5] = (T)B;
      for (int k = 0; k < 5; k++) r[6 + k] = (T)jsi[5 * o + k];
      li[2 * o] = (jl[o] && (jt[o] == JNT_HINGE || jt[o] == JNT_SLIDE)) ? 1 : 0; li[2 * o + 1] = jq[o];
    }
    if (h.ntendon > 0) {
      const auto& tl = h.I("tendon_limited"); const auto& tr = h.D("tendon_range"); const auto& tm = h.D("tendon_margin");
      const auto& tsr = h.D("tendon_solref"); const auto& tsi = h.D("tendon_solimp"); const auto& tiw = h.D("tendon_invweight0");
      for (int t = 0; t < h.ntendon; t++) {
        double K, B;
        KB(&tsr[2 * t], tsi[5 * t + 1], K, B);
        T* r = &lf[(size_t)(h.njnt + t) * 12];
        r[0] = (T)tr[2 * t]; r[1] = (T)tr[2 * t + 1]; r[2] = (T)tm[t]; r[3] = (T)tiw[t]; r[4] = (T)K; r[5] = (T)B;
        for (int k = 0; k < 5; k++) r[6 + k] = (T)tsi[5 * t + k];
        li[2 * (h.njnt + t)] = tl[t] != 0 ? 1 : 0; li[2 * (h.njnt + t) + 1] = t;
      }
    }
    m.lim_f = (FP)alloc.putf(lf);
    m.lim_i = (IP)alloc.puti(li);
    // kinematics records
    const auto& jp = h.D("jnt_pos"); const auto& ja = h.D("jnt_axis"); const auto& q0 = h.D("qpos0");
    std::vector<T> jrec((size_t)h.njnt * 8, (T)0);
    std::vector<int> jirec((size_t)h.njnt * 6, 0);
    const auto& jb = h.I("jnt_bodyid"); const auto& bpar = h.I("body_parentid"); const auto& broot = h.I("body_rootid");
    for (int j = 0; j < h.njnt; j++) {
      for (int k = 0; k < 3; k++) { jrec[8 * j + k] = (T)jp[3 * j + k]; jrec[8 * j + 3 + k] = (T)ja[3 * j + k]; }
      jrec[8 * j + 6] = (T)q0[jq[j]];
      int* r = &jirec[6 * j];
      r[0] = jt[j]; r[1] = jq[j]; r[2] = jd[j]; r[3] = jb[j]; r[4] = bpar[jb[j]]; r[5] = broot[jb[j]];
    }
    {
      const auto& db = h.I("dof_bodyid"); const auto& dj = h.I("dof_jntid"); const auto& bda = h.I("body_dofadr");
      const auto& dd = h.D("dof_damping"); const auto& js = h.D("jnt_stiffness"); const auto& qs = h.D("qpos_spring");
      std::vector<int> dir((size_t)h.nv * 6, 0);
      std::vector<T> dfr((size_t)h.nv * 4, (T)0);
      for (int i = 0; i < h.nv; i++) {
        const int b = db[i], j = dj[i], type = jt[j], first = jd[j];
        int* r = &dir[6 * i];
        r[0] = b; r[1] = bpar[b]; r[2] = bda[b];
        r[3] = type == JNT_FREE ? (i < first + 3 ? -1 : first + 3) : i;
        const bool spring = (type == JNT_HINGE || type == JNT_SLIDE) && js[j] != 0;
        r[4] = spring ? jq[j] : -1;
        dfr[4 * i] = (T)dd[i]; dfr[4 * i + 1] = (T)(spring ? js[j] : 0.0); dfr[4 * i + 2] = (T)(spring ? qs[jq[j]] : 0.0);
      }
      m.dof_irec = (IP)alloc.puti(dir); m.dof_frec = (FP)alloc.putf(dfr);
    }
    const auto& bja = h.I("body_jntadr"); const auto& bjn = h.I("body_jntnum");
    std::vector<int> birec((size_t)h.nbody * 4, 0);
    for (int b = 0; b < h.nbody; b++) { birec[4 * b] = bja[b]; birec[4 * b + 1] = bjn[b]; birec[4 * b + 2] = bjn[b] > 0 ? jt[bja[b]] : -1; }
    m.jnt_rec = (FP)alloc.putf(jrec); m.jnt_irec = (IP)alloc.puti(jirec); m.body_irec = (IP)alloc.puti(birec);
  }
  {
    // bounding radius of every geom about its centre (conservative), folded with the pair margin for the broad phase
    const auto& gt = h.I("geom_type"); const auto& gs = h.D("geom_size");
    const auto& g1 = h.I("pair_geom1"); const auto& g2 = h.I("pair_geom2"); const auto& pm = h.D("pair_margin");
    auto rb = [&](int g) -> double {
      const double a = gs[3 * g], b = gs[3 * g + 1], c = gs[3 * g + 2];
      switch (gt[g]) {
        case G_SPHERE: return a;
        case G_CAPSULE: return a + b;
        case G_ELLIPSOID: return std::max(a, std::max(b, c));
        case G_BOX: return std::sqrt(a * a + b * b + c * c);
        default: return 0.0;                                   // plane: handled through the signed distance
      }
    };
    std::vector<T> cull((size_t)h.npair);
    for (int p = 0; p < h.npair; p++) {
      const bool plane = gt[g1[p]] == G_PLANE;
      const double r = (plane ? 0.0 : rb(g1[p])) + rb(g2[p]) + pm[p];
      cull[p] = (T)(plane ? -std::max(r, 1e-30) : r);
    }
    m.pair_cull = (FP)alloc.putf(cull);
  }
  m.key_qpos = F("key_qpos"); m.key_qvel = F("key_qvel"); m.key_ctrl = F("key_ctrl"); m.key_time = F("key_time");
}

// flat = true: no overlay of temporaries (the two-wave step kernel runs phases of different overlay groups at the same time)
inline Lay make_layout(const HostModel& h, int ncon_max, int nefc_max, size_t sizeofT, bool flat = false) {
  Lay L;
  std::memset(&L, 0, sizeof(L));
  int o = 0;
  auto A = [&](int n) { int r = o; o += n > 0 ? n : 0; return r; };
  int nq = h.nq, nv = h.nv, nu = h.nu, nb = h.nbody, nj = h.njnt, ng = h.ngeom, ns = h.nsite, nt = h.ntendon;
  L.qpos = A(nq); L.qvel = A(nv); L.ctrl = A(nu); L.qacc = A(nv); L.qacc_ws = A(nv); L.qacc_smooth = A(nv);
  L.qfrc_bias = A(nv); L.qfrc_passive = A(nv); L.qfrc_actuator = A(nv); L.qfrc_smooth = A(nv); L.qfrc_constraint = A(nv);
  L.xpos = A(3 * nb); L.xquat = A(4 * nb); L.xipos = A(3 * nb); L.ximat = A(9 * nb);
  L.geom_xpos = A(3 * ng); L.site_xpos = A(3 * ns); L.site_xmat = A(9 * ns);
  L.subtree_com = A(3 * nb); L.cinert = A(10 * nb); L.cdof = A(6 * nv); L.cvel = A(6 * nb); L.bfrc = A(h.has_fluid ? 6 * nb : 0);
  L.M = A(nv * nv); L.W = A(nv * (nv + 1) / 2); L.ten_length = A(nt); L.ten_J = A(nt * nv); L.act_force = A(nu); L.sens = A(h.nsensordata);
  L.con = A(ncon_max * CON_STRIDE);
  L.efc_J = A(nefc_max * nv); L.efc_pos = A(nefc_max); L.efc_D = A(nefc_max); L.efc_aref = A(nefc_max);
  L.tmp = A(nv > 32 ? nv + 1 : 33);          // 1/diag(L) + one dump word; also the zero-padded right-hand side of the 32x32 MFMA solve
  if (h.has_accel) L.cacc = A(6 * nb);          // accelerometers read the bias acceleration after the solve: keep it out of the overlay
  // One overlay region for temporaries with disjoint lifetimes (forward() order: kinematics, com_pos, collision,
  // crb, make_constraint, velocity stage, actuation, solver):
  //   g1 (kinematics .. collision):  xmat, xanchor, xaxis, geom_xmat
  //   g2 (crb):                      crb, dofbuf
  //   g3 (make_constraint):          K, B, imp in jar / jv / force, margins in efc_KBI
  //   g4 (velocity stage):           cdof_dot, cacc, cfrc
  //   g5 (solver .. integrator):     Ma, grad, search, Mv, cholcol, jar, jv, force
  int r0 = o, rend = o;
  L.xmat = A(9 * nb); L.xanchor = A(3 * nj); L.xaxis = A(3 * nj); L.geom_xmat = A(9 * ng);
  if (o > rend) rend = o; if (!flat) o = r0;
  L.crb = A(10 * nb); L.dofbuf = A(6 * nv);
  if (o > rend) rend = o; if (!flat) o = r0;
  L.cdof_dot = A(6 * nv); if (!h.has_accel) L.cacc = A(6 * nb); L.cfrc = A(6 * nb);
  if (o > rend) rend = o; if (!flat) o = r0;
  L.Ma = A(nv); L.grad = A(nv); L.search = A(nv); L.Mv = A(nv); L.cholcol = A(nv + 1);
  L.efc_jar = A(nefc_max); L.efc_jv = A(nefc_max); L.efc_force = A(nefc_max); L.efc_KBI = A(nefc_max);
  if (o > rend) rend = o;
  o = rend;
  L.rk = A(h.integrator == INT_RK4 ? nq + nv + 8 * nv + nv : 0);
  L.nT = o;
  int oi = 0;
  auto AI = [&](int n) { int r = oi; oi += n > 0 ? n : 0; return r; };
  L.i_efc_type = AI(nefc_max); L.i_efc_id = L.i_efc_type; L.i_con_pair = AI(ncon_max); L.i_scal = L.i_con_pair;   // ids are packed into the same words
  L.i_mail = AI(8);
  L.nI = oi;
  size_t bytes = (size_t)L.nT * sizeofT + (size_t)L.nI * sizeof(int);
  L.bytes = (int)((bytes + 15) / 16 * 16);
  return L;
}

}  // namespace mjb
