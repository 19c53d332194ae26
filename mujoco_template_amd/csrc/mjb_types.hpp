// mjb_types.hpp — structures shared by the host API and the device kernels.
//
// DevModel<T>: the compiled model as device pointers (T = float for the product
// path, double for the finite-difference / validation path).  A device copy of the
// struct is read through the constant address space (scalar loads, phase-local).
// Lay: per-environment LDS layout (offsets in elements of T / int), computed on
// the host from the model sizes and the contact/constraint caps.
#pragma once
#include <cstdint>

// Address spaces: model constants are read through the CONSTANT address space (invariant for the
// whole launch -> uniform reads become scalar loads); the host emulation build has no address spaces.
#if defined(MJB_HOST_EMU) || !defined(__HIP_DEVICE_COMPILE__)
#define MJB_CONST
#else
#define MJB_CONST __attribute__((address_space(4)))
#endif

namespace mjb {

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { TRN_JOINT = 0, TRN_SITE = 4 };
enum { INT_EULER = 0, INT_RK4 = 1 };
enum { SENS_JOINTPOS = 0, SENS_GYRO, SENS_ACCEL, SENS_FRAMEQUAT };
enum { EFC_LIMIT_JOINT = 0, EFC_LIMIT_TENDON = 1, EFC_CONTACT_FRICTIONLESS = 2, EFC_CONTACT_PYRAMIDAL = 3 };
enum { CTRL_KEEP = 0, CTRL_ZERO = 1, CTRL_RANDOM = 2, CTRL_FEEDBACK = 3 };

// counters written per environment (int[8])
enum { CNT_NCON = 0, CNT_NEFC, CNT_NITER, CNT_CON_DROPPED, CNT_EFC_DROPPED, CNT_BADQPOS, CNT_BADQVEL, CNT_BADQACC, CNT_N };

template <typename T>
struct DevModel {
  typedef const T MJB_CONST* FP;                     // float-type table
  typedef const int MJB_CONST* IP;                   // int table
  typedef const unsigned long long MJB_CONST* UP;    // 64-bit dof mask table
  int nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, nwrap, nsensor, nsensordata, nkey, npair;
  int nlevel, integrator, disableactuator, iterations, has_damping, has_fluid, has_accel, nvp, nvshift;
  int ncon_max, nefc_max, nsiteact, nmpair, nround, nround_inner, max_nsub, dfs_ok;
  T timestep, gravity[3], density, viscosity, tolerance, meaninertia;
  // kinematic tree
  IP body_parentid, body_rootid, body_jntadr, body_jntnum, body_dofadr, body_dofnum, body_depth, body_round, body_nsub, body_anc;
  IP level_adr, level_body, child_adr, child_id, tri_tab;
  IP dofact_adr, dofact_act, siteact, mpair;   // joint-transmission actuators per dof (CSR), site-transmission actuators, (i<<8|j) ancestor dof pairs of M
  FP body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia, body_subtreemass, body_invweight0;
  UP body_dofmask, dof_ancmask;
  // joints / dofs
  IP jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited;
  FP jnt_pos, jnt_axis, jnt_range, jnt_stiffness, jnt_margin, jnt_solref, jnt_solimp, qpos0, qpos_spring;
  IP dof_bodyid, dof_jntid, dof_parentid;
  FP dof_armature, dof_damping, dof_invweight0;
  // geoms / sites
  IP geom_type, geom_bodyid;
  FP geom_pos, geom_quat, geom_size;
  IP site_bodyid;
  FP site_pos, site_quat;
  // tendons
  IP tendon_adr, tendon_num, tendon_limited, wrap_objid;
  FP tendon_range, tendon_margin, tendon_solref, tendon_solimp, tendon_invweight0, wrap_prm;
  // actuators
  IP actuator_trntype, actuator_trnid, actuator_biastype, actuator_ctrllimited, actuator_forcelimited, actuator_group;
  FP actuator_gear, actuator_gainprm, actuator_biasprm, actuator_ctrlrange, actuator_forcerange;
  // sensors
  IP sensor_type, sensor_objid, sensor_adr;
  // collision pairs
  IP pair_geom1, pair_geom2, pair_condim;
  FP pair_friction, pair_solref, pair_solimp, pair_margin, pair_gap;
  FP pair_kb;        // per pair: margin - gap, body_invweight0 sum (translational), K, B of the contact rows (all model constants)
  FP lim_f;          // per limit object (joints, then tendons) x 12: range lo/hi, margin, diagApprox, K, B, solimp[5], pad
  FP jnt_rec;        // per joint x 8: jnt_pos[3], jnt_axis[3], qpos0[jnt_qposadr], pad   (kinematics: one record, no second hop)
  IP jnt_irec;       // per joint x 6: type, qposadr, dofadr, body, parent of that body, root of that body
  IP dof_irec;       // per dof x 6: body, parent of that body, first dof of the body, dof where its velocity group starts (-1: translational dof of a free joint), qpos address of its spring (-1: none), pad
  FP dof_frec;       // per dof x 4: damping, spring stiffness (0 when not a hinge/slide spring), qpos_spring of that address, pad
  IP body_irec;      // per body x 4: jntadr, jntnum, type of the first joint (-1 if none), pad
  IP pair_body;      // per pair x 2: bodies of geom1 / geom2 (folds geom_bodyid[pair_geom*])
  IP lim_i;          // per limit object x 2: limited-and-limitable flag, index of the value (qpos address / tendon id)
  FP pair_cull;      // broad phase: r1 + r2 + margin (bounding radii); NEGATED when geom1 is a plane (then r2 + margin)
  // keyframes
  FP key_qpos, key_qvel, key_ctrl, key_time;
};

// Per-environment LDS layout.  Offsets of T arrays are in units of T from the
// environment's base; int arrays are in units of int from the int base (placed
// after the nT T-elements).
struct Lay {
  int qpos, qvel, ctrl, qacc, qacc_ws, qacc_smooth;
  int qfrc_bias, qfrc_passive, qfrc_actuator, qfrc_smooth, qfrc_constraint;
  int xpos, xquat, xmat, xipos, ximat, xanchor, xaxis, geom_xpos, geom_xmat, site_xpos, site_xmat;
  int subtree_com, cinert, crb, cdof, cdof_dot, cvel, cacc, cfrc, dofbuf, bfrc;
  int M, W, ten_length, ten_J, act_force, sens;   // sens: sensordata of the last forward pass
  int con;            // contacts: ncon_max * CON_STRIDE
  int efc_J, efc_pos, efc_D, efc_aref, efc_jar, efc_jv, efc_force, efc_KBI;
  int Ma, grad, search, Mv, tmp, cholcol;   // tmp = 1/diag(L); cholcol = scaled pivot column (nv+1)
  int rk;             // RK4 scratch: X0 (nq+nv) + F (4*2*nv) + dX (2*nv)
  int nT;             // total T elements
  int i_efc_type, i_efc_id, i_con_pair, i_scal;  // int arrays
  int i_mail;         // 8 ints: what the two waves of k_step2 tell each other (ncon, nefc, dropped counts, bad-acceleration flag)
  int nI;             // total ints
  int bytes;          // total bytes per environment (rounded to 16)
};

constexpr int CON_STRIDE = 11;  // dist, pos[3], normal[3], tangent1[3], mu (friction[0]); tangent2 = n x t1; pair id in i_con_pair

// Device state of the batch (TS = storage type of the [batch, dof] arrays in HBM).
template <typename TS>
struct DevData {
  int batch;
  TS *qpos, *qvel, *ctrl, *qacc, *qacc_warmstart;
  double* time;
  TS *xpos, *xquat, *xipos, *site_xpos, *geom_xpos, *subtree_com, *sensordata;
  TS *qfrc_inverse, *actuator_moment;   // outputs of the inverse-dynamics mode: [batch, nv], [batch, nu, nv]
  int* counters;
  int* flags;                 // one sticky word for the whole batch: bit 0 contacts dropped, 1 constraint rows dropped, 2 bad-state reset, 3 hand-over timed out
  int* flags_pin;             // the same four bits as four words of PINNED HOST memory (plain stores of 1, no read-modify-write over the bus): after any
                              // stream synchronisation the host reads them without a copy (mjb_engine_flags; MJB_ERR_DEVICE once [3] is set)
  unsigned long long* prof;   // per-phase cycle sums (diagnostic -DMJB_PROFILE build only; null otherwise)
  unsigned* sched;            // ticket mode of k_step: [0] next ticket
  unsigned long long* xfer;   // ticket mode: tagged hand-over buffer [batch, nq + 3 nv + 2] (env_run)
};
enum { PH_KIN = 0, PH_COM, PH_CRB, PH_COLL, PH_CONS, PH_VEL, PH_ACT, PH_SOLVE, PH_INTEG, PH_OTHER, PH_SOL_DIR, PH_SOL_LS, PH_CNT_LS = 12, PH_CNT_DIR, PH_CNT_FACT, PH_SOL_MV, PH_FAC_LOAD = 16, PH_FAC_PANEL, PH_FAC_BACK, PH_FAC_ALL, PH_N = 24 };

// optional per-phase dumps for parity tests (device pointers, may be null)
template <typename TS>
struct DevDebug {
  TS *qM, *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qacc_smooth, *qfrc_constraint;
  TS *efc_J, *efc_aref, *efc_D, *efc_pos, *efc_force, *con;
  TS *cdof, *cinert, *cvel;
  int *efc_type;
};

struct ObsSpecDev {
  int flags;        // bit0 qpos, bit1 qvel, bit2 ctrl, bit3 sensordata, bit4 time, bit5 act(empty), bit6 bodies_inertial
  int nsite, nbody, ngeom, nsubtree, dim;
  const int *site_ids, *body_ids, *geom_ids, *subtree_ids;
};

struct StepArgs {
  int nstep;
  int ctrl_mode;         // CTRL_*
  unsigned seed, step0, env0;   // env0 = global index of this shard's first env (RNG is shard-invariant)
  double ctrl_scale;     // random-ctrl amplitude as a fraction of the ctrl half-range
  double dt;             // model timestep in float64 (time is accumulated in double whatever the state dtype)
  int mode;              // 0 = step, 1 = forward only, 2 = inverse dynamics (mj_inverse) + dense actuator moment
  int write_kin;         // write xpos/xipos/site_xpos/geom_xpos/subtree_com/sensordata of the last forward pass
  int obs_every;         // >0: write flat obs every k steps into obs_out[(step/k), env, dim]
  // CTRL_FEEDBACK: ctrl = clip(u0 - K [differentiatePos(q0, qpos); qvel - v0]) with K [nu, 2nv] row-major (dtype of the arithmetic)
  const void *fb_K, *fb_u0, *fb_q0, *fb_v0;
  // optional ctrl noise of that law (reference lqr.py:160-165): + std[a] * table[(step + env * stride) mod nsteps][a] before the clip
  const void *fb_noise_std, *fb_noise_tab;
  int fb_nsteps, fb_env_stride;
  // host-driven steps of small batches (mjb_step_host): the pinned host mirror block itself, device-visible; the kernel reads the
  // fields the host edited (mirror_mask: bit k = field k of qpos qvel ctrl qacc qacc_warmstart time) from it and writes all six
  // back at the end - no staging copies, no pack / unpack kernels.  nullptr otherwise.
  double* mirror;
  int mirror_mask;
  // != 0: after its state words an environment stores this sequence number into its completion word of the mirror block (system-scope
  // release, behind a system-scope fence): the host polls those words instead of waiting for the end-of-kernel signal (mjb_step_host)
  unsigned long long mirror_seq;
  int fair_bit;          // >0: alternate the issue priority of a SIMD's waves by this bit of the 100 MHz clock (env_run); 0 = leave the hardware's age order
  unsigned ticket_base;  // ticket mode: value of d.sched[0] when this launch starts (the counter is not reset between launches)
  unsigned tagbase;      // ticket mode: tag of this launch's hand-overs (+ the step index at which the hand-over happens), unique among the launches that could still be in the buffer
  int nblk, grid_blocks; // ticket mode: environment blocks of the batch; workgroups launched (the resident ones)
  int chunk_steps;       // >0: TICKET mode of k_step - workgroups draw (environment block, chunk of steps) tickets from d.sched (mjb_kernels.hpp)
  int repeat_phase;      // diagnostic builds only (-DMJB_PHASE_REPEAT, scripts/gpu_phase_pmc.py): the idempotent phase with this index runs TWICE per
                         // step, so that the difference of two PMC passes is that phase's instruction / lane / flop count; -1 / product build: ignored
  unsigned xfer_timeout; // ticket mode: how long a wave waits for a hand-over before it gives up, in ticks of the 100 MHz wall clock (s_memrealtime)
  int xfer_poison_env;   // test hook (MJB_XFER_POISON_ENV): the hand-overs of environment (this - 1) are published with a wrong tag; 0 = off
  int nuniform, nchunk;  // ticket mode, the chunk plan of an environment's nstep steps: `nuniform` chunks of chunk_steps steps, then every
                         // further chunk takes HALF of what is left (guided taper down to single steps: the launch's tail is half of the
                         // LAST chunk); nchunk = all chunks.  chunk_plan() below is the one definition, used by host and device.
};

// Chunk k of the plan (nstep, chunk_steps, nuniform): steps [s0, s1).  Host and device.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void chunk_plan(int nstep, int chunk_steps, int nuniform, int k, int& s0, int& s1) {
  if (k < nuniform) { s0 = k * chunk_steps; s1 = s0 + chunk_steps; return; }
  int at = nuniform * chunk_steps;
  for (int j = nuniform;; j++) {
    const int rem = nstep - at, sz = rem > 1 ? (rem + 1) / 2 : rem;
    if (j == k) { s0 = at; s1 = at + sz; return; }
    at += sz;
  }
}
// the plan for a launch of nstep steps with uniform chunks of c: nuniform, nchunk
inline void chunk_plan_counts(int nstep, int c, int& nuniform, int& nchunk) {
  nuniform = nstep >= 2 * c ? (nstep - c) / c : 0;
  int rem = nstep - nuniform * c, nt = 0;
  while (rem > 0) { rem -= rem > 1 ? (rem + 1) / 2 : rem; nt++; }
  nchunk = nuniform + nt;
}

// References to launch arguments as the device code takes them: objects in the constant address space (the kernarg segment), so
// that every field read is a scalar load that can be redone where it is needed instead of a value kept live (k_step_body).
typedef const StepArgs MJB_CONST& ArgsRef;
typedef const ObsSpecDev MJB_CONST& ObsRef;
template <typename TS> using DataRef = const DevData<TS> MJB_CONST&;
template <typename TS> using DebugRef = const DevDebug<TS> MJB_CONST&;
// the argument list of k_step / mjb_k_step_spec as one struct (same member order, same natural alignment = the kernarg layout)
template <typename T, typename TS> struct StepKernArgs {
  const DevModel<T>* mg; const Lay* lg; DevData<TS> d; DevDebug<TS> dbg; StepArgs a; ObsSpecDev obs; TS* obs_out;
};

}  // namespace mjb
