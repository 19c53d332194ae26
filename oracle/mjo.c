/*
 * mjo.c — CPU ORACLE, float64, one environment.  See mjo.h for the header note
 * ("parity unpinned", who may load this).  Every stage cites the reference call
 * site whose third-party arithmetic it restates; MuJoCo internals are
 * [MJ-KNOWLEDGE].
 */
#include "mjo.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Phase markers of the instrumented build (oracle/mjo_flops.hpp: the same file compiled as C++ with a counting `double`; SURVEY.md
 * §8(d) flop accounting).  No-ops here. */
#ifndef MJO_PHASE
#define MJO_PHASE(k) ((void)0)
#endif

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define MAXVAL 1e10
#define PI 3.14159265358979323846

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { TRN_JOINT = 0, TRN_SITE = 4 };
enum { INT_EULER = 0, INT_RK4 = 1 };
enum { SENS_JOINTPOS = 0, SENS_GYRO, SENS_ACCEL, SENS_FRAMEQUAT };
enum { EFC_LIMIT_JOINT = 0, EFC_LIMIT_TENDON = 1, EFC_CONTACT_FRICTIONLESS = 2, EFC_CONTACT_PYRAMIDAL = 3 };

static char g_err[512];
const char* mjo_last_error(void) { return g_err; }

struct mjoModel {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, nwrap, nsensor, nsensordata, nkey, npair;
  int integrator, disableactuator, iterations;
  int nconmax, nefcmax;           /* caps (0 = unlimited) */
  int ncon_alloc, nefc_alloc;     /* worst case for this pair list */
  double timestep, gravity[3], density, viscosity, impratio, tolerance, meaninertia;
  /* bodies */
  int *body_parentid, *body_rootid, *body_weldid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_subtreemass, *body_invweight0;
  /* joints / dofs */
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  double *jnt_pos, *jnt_axis, *jnt_range, *jnt_stiffness, *jnt_margin, *jnt_solref, *jnt_solimp, *qpos0, *qpos_spring;
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  double *dof_armature, *dof_damping, *dof_invweight0;
  /* geoms / sites */
  int *geom_type, *geom_bodyid;
  double *geom_pos, *geom_quat, *geom_size;
  int* site_bodyid;
  double *site_pos, *site_quat;
  /* tendons */
  int *tendon_adr, *tendon_num, *tendon_limited, *wrap_objid;
  double *tendon_range, *tendon_margin, *tendon_solref, *tendon_solimp, *tendon_invweight0, *wrap_prm;
  /* actuators */
  int *actuator_trntype, *actuator_trnid, *actuator_biastype, *actuator_ctrllimited, *actuator_forcelimited, *actuator_group;
  double *actuator_gear, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  /* sensors */
  int *sensor_type, *sensor_objid, *sensor_adr;
  /* collision pairs */
  int *pair_geom1, *pair_geom2, *pair_condim;
  double *pair_friction, *pair_solref, *pair_solimp, *pair_margin, *pair_gap;
  /* keyframes */
  double *key_qpos, *key_qvel, *key_ctrl, *key_time;
  int has_damping;
  int round_mask;                 /* precision study only (mjo_set_round_mask): which phases' outputs are rounded to fp32; 0 everywhere else */
};

typedef struct {
  double dist, pos[3], frame[9], friction[5], solref[2], solimp[5], includemargin;
  int dim, geom1, geom2, efc_address;
} mjoContact;

struct mjoData {
  const mjoModel* m;
  double time;
  double *qpos, *qvel, *ctrl, *qacc, *qacc_warmstart, *qacc_smooth, *qfrc_applied;
  double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qfrc_constraint, *qfrc_inverse;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *qM, *qL, *qH;             /* dense nv*nv: mass matrix, its Cholesky factor, solver Hessian factor */
  double *ten_length, *ten_J, *ten_velocity;
  double *actuator_length, *actuator_velocity, *actuator_force, *actuator_moment;
  double* sensordata;
  mjoContact* contact;
  int ncon, nefc, solver_niter, ncon_dropped, nefc_dropped;
  int warn_badqpos, warn_badqvel, warn_badqacc;
  int* efc_type; int* efc_id;
  double *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_D, *efc_KBIP, *efc_vel, *efc_aref, *efc_force, *efc_jar;
  int* efc_active;
  /* solver scratch */
  double *s_Ma, *s_grad, *s_Mgrad, *s_search, *s_Mv, *s_jv, *s_tmp;
  double solver_cost;
  double* jacbuf;   /* 6*nv scratch for Jacobians */
  int* iscratch;
};

/* ------------------------------------------------------------------------- */
/* small vector / quaternion helpers                                         */
/* ------------------------------------------------------------------------- */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static inline double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; } else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static inline void copy3(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void addscl3(double* r, const double* a, const double* b, double s) { r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2]; }
static inline void mulmatvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmatTvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2], z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmat3(double* r, const double* a, const double* b) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof t);
}
static inline void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void quat_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static inline void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static inline void axisangle2quat(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* quat <- quat * exp(w*h/2), w in the local frame (mju_quatIntegrate) */
static void quat_integrate(double* q, const double* w, double h) {
  double ax[3] = {w[0], w[1], w[2]};
  double ang = h * normalize3(ax);
  if (norm3(w) < MINVAL) return;
  double qr[4], out[4];
  axisangle2quat(qr, ax, ang);
  quat_mul(out, q, qr);
  quat_normalize(out);
  memcpy(q, out, sizeof out);
}
/* 3D velocity taking qa to qb in unit time, local frame (mju_subQuat) */
static void quat_sub(double* res, const double* qa, const double* qb) {
  double qn[4] = {qa[0], -qa[1], -qa[2], -qa[3]}, qd[4];
  quat_mul(qd, qn, qb);
  double ax[3] = {qd[1], qd[2], qd[3]};
  double sinh_ = normalize3(ax);
  if (norm3(qd + 1) < MINVAL) { res[0] = res[1] = res[2] = 0; return; }
  double ang = 2 * atan2(sinh_, qd[0]);
  if (ang > PI) ang -= 2 * PI;
  res[0] = ax[0] * ang; res[1] = ax[1] * ang; res[2] = ax[2] * ang;
}

/* spatial (6D, [rot; lin]) helpers — conventions of MuJoCo's com-based quantities */
static void inert_com(double* res, const double* inert, const double* mat, const double* dif, double mass) {
  double tmp[9];
  for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) tmp[3 * k + j] = inert[k] * mat[3 * j + k]; /* diag(inert)*mat' */
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
static void mul_inert_vec(double* res, const double* i, const double* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(double* res, const double* vel, const double* v) {
  res[0] = -vel[2] * v[1] + vel[1] * v[2];
  res[1] = vel[2] * v[0] - vel[0] * v[2];
  res[2] = -vel[1] * v[0] + vel[0] * v[1];
  res[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  res[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  res[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
static void cross_force(double* res, const double* vel, const double* f) {
  res[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  res[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  res[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  res[3] = -vel[2] * f[4] + vel[1] * f[5];
  res[4] = vel[2] * f[3] - vel[0] * f[5];
  res[5] = -vel[1] * f[3] + vel[0] * f[4];
}

/* dense Cholesky  A = L L^T (lower, in place), returns min pivot */
static double chol_factor(double* A, int n) {
  double minp = 1e300;
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
    if (s < MINVAL) s = MINVAL;
    if (s < minp) minp = s;
    double l = sqrt(s);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / l;
    }
  }
  return minp;
}
static void chol_solve(const double* L, int n, double* x) {
  for (int i = 0; i < n; i++) {
    double t = x[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}

/* ------------------------------------------------------------------------- */
/* model construction from a table of named arrays                            */
/* ------------------------------------------------------------------------- */
typedef struct { int n; const char* const* names; const void* const* ptrs; const int* dt; const long* cnt; int fail; } Table;

static int tbl_find(Table* t, const char* name) {
  for (int i = 0; i < t->n; i++) if (!strcmp(t->names[i], name)) return i;
  return -1;
}
static double* tbl_d(Table* t, const char* name, long count) {
  int i = tbl_find(t, name);
  if (i < 0 || t->dt[i] != 0 || t->cnt[i] != count) {
    snprintf(g_err, sizeof g_err, "model field '%s': missing or wrong dtype/size (want f64 x %ld, got %ld)", name, count, i < 0 ? -1L : t->cnt[i]);
    t->fail = 1; return NULL;
  }
  double* p = (double*)malloc(sizeof(double) * (count > 0 ? count : 1));
  memcpy(p, t->ptrs[i], sizeof(double) * count);
  return p;
}
static int* tbl_i(Table* t, const char* name, long count) {
  int i = tbl_find(t, name);
  if (i < 0 || t->dt[i] != 1 || t->cnt[i] != count) {
    snprintf(g_err, sizeof g_err, "model field '%s': missing or wrong dtype/size (want i32 x %ld, got %ld)", name, count, i < 0 ? -1L : t->cnt[i]);
    t->fail = 1; return NULL;
  }
  int* p = (int*)malloc(sizeof(int) * (count > 0 ? count : 1));
  memcpy(p, t->ptrs[i], sizeof(int) * count);
  return p;
}
static double tbl_scalar(Table* t, const char* name) { double* p = tbl_d(t, name, 1); double v = p ? *p : 0; free(p); return v; }
static int tbl_iscalar(Table* t, const char* name) { int* p = tbl_i(t, name, 1); int v = p ? *p : 0; free(p); return v; }

static int pair_maxcon(const mjoModel* m, int p) {
  int t1 = m->geom_type[m->pair_geom1[p]], t2 = m->geom_type[m->pair_geom2[p]];
  if (t1 == G_PLANE && t2 == G_CAPSULE) return 2;
  if (t1 == G_PLANE && t2 == G_BOX) return 4;
  if (t1 == G_CAPSULE && t2 == G_CAPSULE) return 2;       /* parallel axes: up to two */
  return 1;
}

mjoModel* mjo_model_create(int nfield, const char* const* names, const void* const* ptrs, const int* dtypes, const long* counts) {
  Table T = {nfield, names, ptrs, dtypes, counts, 0};
  Table* t = &T;
  mjoModel* m = (mjoModel*)calloc(1, sizeof(mjoModel));
  m->nq = tbl_iscalar(t, "nq"); m->nv = tbl_iscalar(t, "nv"); m->nu = tbl_iscalar(t, "nu");
  m->nbody = tbl_iscalar(t, "nbody"); m->njnt = tbl_iscalar(t, "njnt"); m->ngeom = tbl_iscalar(t, "ngeom");
  m->nsite = tbl_iscalar(t, "nsite"); m->ntendon = tbl_iscalar(t, "ntendon"); m->nwrap = tbl_iscalar(t, "nwrap");
  m->nsensor = tbl_iscalar(t, "nsensor"); m->nsensordata = tbl_iscalar(t, "nsensordata");
  m->nkey = tbl_iscalar(t, "nkey"); m->npair = tbl_iscalar(t, "npair");
  m->integrator = tbl_iscalar(t, "integrator"); m->disableactuator = tbl_iscalar(t, "disableactuator");
  m->iterations = tbl_iscalar(t, "iterations");
  m->timestep = tbl_scalar(t, "timestep"); m->density = tbl_scalar(t, "density"); m->viscosity = tbl_scalar(t, "viscosity");
  m->impratio = tbl_scalar(t, "impratio"); m->tolerance = tbl_scalar(t, "tolerance"); m->meaninertia = tbl_scalar(t, "meaninertia");
  if (T.fail) { free(m); return NULL; }
  double* g = tbl_d(t, "gravity", 3); if (g) { memcpy(m->gravity, g, 24); free(g); }
  int nb = m->nbody, nj = m->njnt, nv = m->nv, ng = m->ngeom, ns = m->nsite, nt = m->ntendon, nu = m->nu, np = m->npair;
#define GD(f, c) m->f = tbl_d(t, #f, (long)(c))
#define GI(f, c) m->f = tbl_i(t, #f, (long)(c))
  GI(body_parentid, nb); GI(body_rootid, nb); GI(body_weldid, nb); GI(body_jntadr, nb); GI(body_jntnum, nb); GI(body_dofadr, nb); GI(body_dofnum, nb);
  GD(body_pos, nb * 3); GD(body_quat, nb * 4); GD(body_ipos, nb * 3); GD(body_iquat, nb * 4); GD(body_mass, nb); GD(body_inertia, nb * 3);
  GD(body_subtreemass, nb); GD(body_invweight0, nb * 2);
  GI(jnt_type, nj); GI(jnt_qposadr, nj); GI(jnt_dofadr, nj); GI(jnt_bodyid, nj); GI(jnt_limited, nj);
  GD(jnt_pos, nj * 3); GD(jnt_axis, nj * 3); GD(jnt_range, nj * 2); GD(jnt_stiffness, nj); GD(jnt_margin, nj); GD(jnt_solref, nj * 2); GD(jnt_solimp, nj * 5);
  GD(qpos0, m->nq); GD(qpos_spring, m->nq);
  GI(dof_bodyid, nv); GI(dof_jntid, nv); GI(dof_parentid, nv); GD(dof_armature, nv); GD(dof_damping, nv); GD(dof_invweight0, nv);
  GI(geom_type, ng); GI(geom_bodyid, ng); GD(geom_pos, ng * 3); GD(geom_quat, ng * 4); GD(geom_size, ng * 3);
  GI(site_bodyid, ns); GD(site_pos, ns * 3); GD(site_quat, ns * 4);
  GI(tendon_adr, nt); GI(tendon_num, nt); GI(tendon_limited, nt); GI(wrap_objid, m->nwrap);
  GD(tendon_range, nt * 2); GD(tendon_margin, nt); GD(tendon_solref, nt * 2); GD(tendon_solimp, nt * 5); GD(tendon_invweight0, nt); GD(wrap_prm, m->nwrap);
  GI(actuator_trntype, nu); GI(actuator_trnid, nu * 2); GI(actuator_biastype, nu); GI(actuator_ctrllimited, nu); GI(actuator_forcelimited, nu); GI(actuator_group, nu);
  GD(actuator_gear, nu * 6); GD(actuator_gainprm, nu * 3); GD(actuator_biasprm, nu * 3); GD(actuator_ctrlrange, nu * 2); GD(actuator_forcerange, nu * 2);
  GI(sensor_type, m->nsensor); GI(sensor_objid, m->nsensor); GI(sensor_adr, m->nsensor);
  GI(pair_geom1, np); GI(pair_geom2, np); GI(pair_condim, np);
  GD(pair_friction, np * 5); GD(pair_solref, np * 2); GD(pair_solimp, np * 5); GD(pair_margin, np); GD(pair_gap, np);
  GD(key_qpos, m->nkey * m->nq); GD(key_qvel, m->nkey * nv); GD(key_ctrl, m->nkey * nu); GD(key_time, m->nkey);
#undef GD
#undef GI
  if (T.fail) { mjo_model_free(m); return NULL; }
  m->has_damping = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) m->has_damping = 1;
  int nc = 0;
  for (int p = 0; p < np; p++) nc += pair_maxcon(m, p);
  m->ncon_alloc = nc;
  m->nefc_alloc = 2 * nj + 2 * nt + 4 * nc;
  return m;
}

void mjo_model_free(mjoModel* m) {
  if (!m) return;
  void** fields[] = {
    (void**)&m->body_parentid, (void**)&m->body_rootid, (void**)&m->body_weldid, (void**)&m->body_jntadr, (void**)&m->body_jntnum,
    (void**)&m->body_dofadr, (void**)&m->body_dofnum, (void**)&m->body_pos, (void**)&m->body_quat, (void**)&m->body_ipos,
    (void**)&m->body_iquat, (void**)&m->body_mass, (void**)&m->body_inertia, (void**)&m->body_subtreemass, (void**)&m->body_invweight0,
    (void**)&m->jnt_type, (void**)&m->jnt_qposadr, (void**)&m->jnt_dofadr, (void**)&m->jnt_bodyid, (void**)&m->jnt_limited,
    (void**)&m->jnt_pos, (void**)&m->jnt_axis, (void**)&m->jnt_range, (void**)&m->jnt_stiffness, (void**)&m->jnt_margin,
    (void**)&m->jnt_solref, (void**)&m->jnt_solimp, (void**)&m->qpos0, (void**)&m->qpos_spring, (void**)&m->dof_bodyid,
    (void**)&m->dof_jntid, (void**)&m->dof_parentid, (void**)&m->dof_armature, (void**)&m->dof_damping, (void**)&m->dof_invweight0,
    (void**)&m->geom_type, (void**)&m->geom_bodyid, (void**)&m->geom_pos, (void**)&m->geom_quat, (void**)&m->geom_size,
    (void**)&m->site_bodyid, (void**)&m->site_pos, (void**)&m->site_quat, (void**)&m->tendon_adr, (void**)&m->tendon_num,
    (void**)&m->tendon_limited, (void**)&m->wrap_objid, (void**)&m->tendon_range, (void**)&m->tendon_margin, (void**)&m->tendon_solref,
    (void**)&m->tendon_solimp, (void**)&m->tendon_invweight0, (void**)&m->wrap_prm, (void**)&m->actuator_trntype, (void**)&m->actuator_trnid,
    (void**)&m->actuator_biastype, (void**)&m->actuator_ctrllimited, (void**)&m->actuator_forcelimited, (void**)&m->actuator_group,
    (void**)&m->actuator_gear, (void**)&m->actuator_gainprm, (void**)&m->actuator_biasprm, (void**)&m->actuator_ctrlrange,
    (void**)&m->actuator_forcerange, (void**)&m->sensor_type, (void**)&m->sensor_objid, (void**)&m->sensor_adr, (void**)&m->pair_geom1,
    (void**)&m->pair_geom2, (void**)&m->pair_condim, (void**)&m->pair_friction, (void**)&m->pair_solref, (void**)&m->pair_solimp,
    (void**)&m->pair_margin, (void**)&m->pair_gap, (void**)&m->key_qpos, (void**)&m->key_qvel, (void**)&m->key_ctrl, (void**)&m->key_time};
  for (size_t i = 0; i < sizeof(fields) / sizeof(fields[0]); i++) free(*fields[i]);
  free(m);
}

void mjo_set_disableactuator(mjoModel* m, int mask) { m->disableactuator = mask; }
void mjo_set_limits(mjoModel* m, int nconmax, int nefcmax) { m->nconmax = nconmax; m->nefcmax = nefcmax; }
void mjo_set_solver(mjoModel* m, int iterations, double tolerance) { m->iterations = iterations; m->tolerance = tolerance; }
void mjo_set_round_mask(mjoModel* m, int mask) { m->round_mask = mask; }

/* ------------------------------------------------------------------------- */
/* data                                                                        */
/* ------------------------------------------------------------------------- */
static double* dalloc(long n) { return (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

mjoData* mjo_data_create(const mjoModel* m) {
  mjoData* d = (mjoData*)calloc(1, sizeof(mjoData));
  d->m = m;
  int nq = m->nq, nv = m->nv, nu = m->nu, nb = m->nbody, nj = m->njnt, ng = m->ngeom, ns = m->nsite, nt = m->ntendon;
  d->qpos = dalloc(nq); d->qvel = dalloc(nv); d->ctrl = dalloc(nu); d->qacc = dalloc(nv); d->qacc_warmstart = dalloc(nv);
  d->qacc_smooth = dalloc(nv); d->qfrc_applied = dalloc(nv); d->qfrc_bias = dalloc(nv); d->qfrc_passive = dalloc(nv);
  d->qfrc_actuator = dalloc(nv); d->qfrc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qfrc_inverse = dalloc(nv);
  d->xpos = dalloc(nb * 3); d->xquat = dalloc(nb * 4); d->xmat = dalloc(nb * 9); d->xipos = dalloc(nb * 3); d->ximat = dalloc(nb * 9);
  d->xanchor = dalloc(nj * 3); d->xaxis = dalloc(nj * 3); d->geom_xpos = dalloc(ng * 3); d->geom_xmat = dalloc(ng * 9);
  d->site_xpos = dalloc(ns * 3); d->site_xmat = dalloc(ns * 9); d->subtree_com = dalloc(nb * 3);
  d->cinert = dalloc(nb * 10); d->crb = dalloc(nb * 10); d->cdof = dalloc(nv * 6); d->cdof_dot = dalloc(nv * 6);
  d->cvel = dalloc(nb * 6); d->cacc = dalloc(nb * 6); d->cfrc = dalloc(nb * 6);
  d->qM = dalloc(nv * nv); d->qL = dalloc(nv * nv); d->qH = dalloc(nv * nv);
  d->ten_length = dalloc(nt); d->ten_J = dalloc(nt * nv); d->ten_velocity = dalloc(nt);
  d->actuator_length = dalloc(nu); d->actuator_velocity = dalloc(nu); d->actuator_force = dalloc(nu); d->actuator_moment = dalloc(nu * nv);
  d->sensordata = dalloc(m->nsensordata);
  d->contact = (mjoContact*)calloc((size_t)(m->ncon_alloc > 0 ? m->ncon_alloc : 1), sizeof(mjoContact));
  int ne = m->nefc_alloc > 0 ? m->nefc_alloc : 1;
  d->efc_type = (int*)calloc(ne, sizeof(int)); d->efc_id = (int*)calloc(ne, sizeof(int)); d->efc_active = (int*)calloc(ne, sizeof(int));
  d->efc_J = dalloc((long)ne * nv); d->efc_pos = dalloc(ne); d->efc_margin = dalloc(ne); d->efc_diagApprox = dalloc(ne);
  d->efc_R = dalloc(ne); d->efc_D = dalloc(ne); d->efc_KBIP = dalloc(ne * 4); d->efc_vel = dalloc(ne); d->efc_aref = dalloc(ne);
  d->efc_force = dalloc(ne); d->efc_jar = dalloc(ne);
  d->s_Ma = dalloc(nv); d->s_grad = dalloc(nv); d->s_Mgrad = dalloc(nv); d->s_search = dalloc(nv); d->s_Mv = dalloc(nv);
  d->s_jv = dalloc(ne); d->s_tmp = dalloc(nv > ne ? nv : ne); d->jacbuf = dalloc(6 * nv);
  mjo_reset(m, d);
  return d;
}

void mjo_data_free(mjoData* d) {
  if (!d) return;
  double** f[] = {&d->qpos, &d->qvel, &d->ctrl, &d->qacc, &d->qacc_warmstart, &d->qacc_smooth, &d->qfrc_applied, &d->qfrc_bias,
    &d->qfrc_passive, &d->qfrc_actuator, &d->qfrc_smooth, &d->qfrc_constraint, &d->qfrc_inverse, &d->xpos, &d->xquat, &d->xmat, &d->xipos, &d->ximat,
    &d->xanchor, &d->xaxis, &d->geom_xpos, &d->geom_xmat, &d->site_xpos, &d->site_xmat, &d->subtree_com, &d->cinert, &d->crb,
    &d->cdof, &d->cdof_dot, &d->cvel, &d->cacc, &d->cfrc, &d->qM, &d->qL, &d->qH, &d->ten_length, &d->ten_J, &d->ten_velocity,
    &d->actuator_length, &d->actuator_velocity, &d->actuator_force, &d->actuator_moment, &d->sensordata, &d->efc_J, &d->efc_pos,
    &d->efc_margin, &d->efc_diagApprox, &d->efc_R, &d->efc_D, &d->efc_KBIP, &d->efc_vel, &d->efc_aref, &d->efc_force, &d->efc_jar,
    &d->s_Ma, &d->s_grad, &d->s_Mgrad, &d->s_search, &d->s_Mv, &d->s_jv, &d->s_tmp, &d->jacbuf};
  for (size_t i = 0; i < sizeof(f) / sizeof(f[0]); i++) free(*f[i]);
  free(d->contact); free(d->efc_type); free(d->efc_id); free(d->efc_active);
  free(d);
}

double* mjo_data_array(mjoData* d, const char* name, long* count) {
  const mjoModel* m = d->m;
  int nv = m->nv, nb = m->nbody;
#define F(f, c) if (!strcmp(name, #f)) { if (count) *count = (long)(c); return d->f; }
  F(qpos, m->nq) F(qvel, nv) F(ctrl, m->nu) F(qacc, nv) F(qacc_warmstart, nv) F(qacc_smooth, nv) F(qfrc_applied, nv)
  F(qfrc_bias, nv) F(qfrc_passive, nv) F(qfrc_actuator, nv) F(qfrc_smooth, nv) F(qfrc_constraint, nv) F(qfrc_inverse, nv)
  F(xpos, nb * 3) F(xquat, nb * 4) F(xmat, nb * 9) F(xipos, nb * 3) F(ximat, nb * 9) F(xanchor, m->njnt * 3) F(xaxis, m->njnt * 3)
  F(geom_xpos, m->ngeom * 3) F(geom_xmat, m->ngeom * 9) F(site_xpos, m->nsite * 3) F(site_xmat, m->nsite * 9) F(subtree_com, nb * 3)
  F(cinert, nb * 10) F(crb, nb * 10) F(cdof, nv * 6) F(cdof_dot, nv * 6) F(cvel, nb * 6) F(cacc, nb * 6) F(cfrc, nb * 6)
  F(qM, nv * nv) F(qL, nv * nv) F(ten_length, m->ntendon) F(ten_J, m->ntendon * nv) F(actuator_length, m->nu)
  F(actuator_velocity, m->nu) F(actuator_force, m->nu) F(actuator_moment, m->nu * nv) F(sensordata, m->nsensordata)
  F(efc_J, (long)d->nefc * nv) F(efc_pos, d->nefc) F(efc_D, d->nefc) F(efc_R, d->nefc) F(efc_aref, d->nefc) F(efc_vel, d->nefc)
  F(efc_force, d->nefc) F(efc_diagApprox, d->nefc)
#undef F
  snprintf(g_err, sizeof g_err, "unknown data array '%s'", name);
  if (count) *count = -1;
  return NULL;
}

int* mjo_data_iarray(mjoData* d, const char* name, long* count) {
  static int scratch[8];
  if (!strcmp(name, "counters")) {
    scratch[0] = d->ncon; scratch[1] = d->nefc; scratch[2] = d->solver_niter; scratch[3] = d->ncon_dropped;
    scratch[4] = d->nefc_dropped; scratch[5] = d->warn_badqpos; scratch[6] = d->warn_badqvel; scratch[7] = d->warn_badqacc;
    if (count) *count = 8;
    return scratch;
  }
  if (!strcmp(name, "efc_type")) { if (count) *count = d->nefc; return d->efc_type; }
  if (count) *count = -1;
  return NULL;
}

/* contact geometry export for tests: out[ncon*(1+3+9+2)] = dist,pos,frame,geom1,geom2 */
long mjo_get_contacts(const mjoData* d, double* out, long maxcon) {
  long n = d->ncon < maxcon ? d->ncon : maxcon;
  for (long i = 0; i < n; i++) {
    const mjoContact* c = d->contact + i;
    double* o = out + 15 * i;
    o[0] = c->dist; memcpy(o + 1, c->pos, 24); memcpy(o + 4, c->frame, 72); o[13] = c->geom1; o[14] = c->geom2;
  }
  return d->ncon;
}

double mjo_get_time(const mjoData* d) { return d->time; }
void mjo_set_time(mjoData* d, double t) { d->time = t; }

void mjo_reset(const mjoModel* m, mjoData* d) {
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->qacc, 0, sizeof(double) * m->nv);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  memset(d->qfrc_applied, 0, sizeof(double) * m->nv);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  d->time = 0;
  d->ncon = d->nefc = 0;
}

int mjo_reset_keyframe(const mjoModel* m, mjoData* d, int key) {
  if (key < 0 || key >= m->nkey) return -1;
  mjo_reset(m, d);
  memcpy(d->qpos, m->key_qpos + (size_t)key * m->nq, sizeof(double) * m->nq);
  memcpy(d->qvel, m->key_qvel + (size_t)key * m->nv, sizeof(double) * m->nv);
  memcpy(d->ctrl, m->key_ctrl + (size_t)key * m->nu, sizeof(double) * m->nu);
  d->time = m->key_time[key];
  return 0;
}

/* ------------------------------------------------------------------------- */
/* A1  kinematics (mj_kinematics)                                              */
/* ------------------------------------------------------------------------- */
static void kinematics(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_KIN);
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  copy3(d->xipos, d->xpos); memcpy(d->ximat, d->xmat, 72);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b], jadr = m->body_jntadr[b], jnum = m->body_jntnum[b];
    double pos[3], quat[4];
    if (jnum == 1 && m->jnt_type[jadr] == JNT_FREE) {
      int qa = m->jnt_qposadr[jadr];
      quat_normalize(d->qpos + qa + 3);            /* mj_kinematics normalises qpos quaternions in place */
      copy3(pos, d->qpos + qa); memcpy(quat, d->qpos + qa + 3, 32);
      copy3(d->xanchor + 3 * jadr, pos);
      d->xaxis[3 * jadr] = 0; d->xaxis[3 * jadr + 1] = 0; d->xaxis[3 * jadr + 2] = 1;
    } else {
      double t[3];
      mulmatvec3(t, d->xmat + 9 * p, m->body_pos + 3 * b);
      addscl3(pos, d->xpos + 3 * p, t, 1.0);
      quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
      for (int j = jadr; j < jadr + jnum; j++) {
        double R[9], anchor[3], axis[3];
        quat2mat(R, quat);
        mulmatvec3(anchor, R, m->jnt_pos + 3 * j); addscl3(anchor, anchor, pos, 1.0);
        mulmatvec3(axis, R, m->jnt_axis + 3 * j);
        copy3(d->xanchor + 3 * j, anchor); copy3(d->xaxis + 3 * j, axis);
        int qa = m->jnt_qposadr[j];
        double val = d->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == JNT_SLIDE) {
          addscl3(pos, pos, axis, val);
        } else { /* hinge */
          double ql[4], qn[4], v[3];
          axisangle2quat(ql, m->jnt_axis + 3 * j, val);
          quat_mul(qn, quat, ql); memcpy(quat, qn, 32);
          quat2mat(R, quat);
          mulmatvec3(v, R, m->jnt_pos + 3 * j);
          pos[0] = anchor[0] - v[0]; pos[1] = anchor[1] - v[1]; pos[2] = anchor[2] - v[2];
        }
      }
    }
    quat_normalize(quat);
    copy3(d->xpos + 3 * b, pos); memcpy(d->xquat + 4 * b, quat, 32); quat2mat(d->xmat + 9 * b, quat);
    double t[3], iq[4];
    mulmatvec3(t, d->xmat + 9 * b, m->body_ipos + 3 * b); addscl3(d->xipos + 3 * b, pos, t, 1.0);
    quat_mul(iq, quat, m->body_iquat + 4 * b); quat2mat(d->ximat + 9 * b, iq);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], q[4];
    mulmatvec3(t, d->xmat + 9 * b, m->geom_pos + 3 * g); addscl3(d->geom_xpos + 3 * g, d->xpos + 3 * b, t, 1.0);
    quat_mul(q, d->xquat + 4 * b, m->geom_quat + 4 * g); quat_normalize(q); quat2mat(d->geom_xmat + 9 * g, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double t[3], q[4];
    mulmatvec3(t, d->xmat + 9 * b, m->site_pos + 3 * s); addscl3(d->site_xpos + 3 * s, d->xpos + 3 * b, t, 1.0);
    quat_mul(q, d->xquat + 4 * b, m->site_quat + 4 * s); quat_normalize(q); quat2mat(d->site_xmat + 9 * s, q);
  }
}

/* ------------------------------------------------------------------------- */
/* A2  com-frame quantities (mj_comPos)                                        */
/* ------------------------------------------------------------------------- */
static void com_pos(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_COM);
  int nb = m->nbody;
  for (int b = 0; b < nb; b++) for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = m->body_mass[b] * d->xipos[3 * b + k];
  for (int b = nb - 1; b > 0; b--) { int p = m->body_parentid[b]; for (int k = 0; k < 3; k++) d->subtree_com[3 * p + k] += d->subtree_com[3 * b + k]; }
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < MINVAL) copy3(d->subtree_com + 3 * b, d->xipos + 3 * b);
    else for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] /= m->body_subtreemass[b];
  }
  memset(d->cinert, 0, sizeof(double) * 10);
  for (int b = 1; b < nb; b++) {
    double off[3];
    const double* rc = d->subtree_com + 3 * m->body_rootid[b];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * b + k] - rc[k];
    inert_com(d->cinert + 10 * b, m->body_inertia + 3 * b, d->ximat + 9 * b, off, m->body_mass[b]);
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    double off[3];
    const double* rc = d->subtree_com + 3 * m->body_rootid[b];
    for (int k = 0; k < 3; k++) off[k] = rc[k] - d->xanchor[3 * j + k];
    double* cd = d->cdof + 6 * da;
    switch (m->jnt_type[j]) {
      case JNT_FREE:
        memset(cd, 0, sizeof(double) * 36);
        for (int i = 0; i < 3; i++) cd[6 * i + 3 + i] = 1;
        for (int i = 0; i < 3; i++) {
          double ax[3] = {d->xmat[9 * b + i], d->xmat[9 * b + 3 + i], d->xmat[9 * b + 6 + i]};
          double* c = cd + 6 * (3 + i);
          copy3(c, ax); cross3(c + 3, ax, off);
        }
        break;
      case JNT_SLIDE:
        cd[0] = cd[1] = cd[2] = 0; copy3(cd + 3, d->xaxis + 3 * j);
        break;
      default: /* hinge */
        copy3(cd, d->xaxis + 3 * j); cross3(cd + 3, d->xaxis + 3 * j, off);
    }
  }
}

/* A3  fixed tendons + actuator transmission (mj_tendon, mj_transmission) */
static void jac_point(const mjoModel* m, const mjoData* d, int body, const double* point, double* jacp, double* jacr);

static void tendon_transmission(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_TENDON);
  int nv = m->nv;
  for (int t = 0; t < m->ntendon; t++) {
    double len = 0;
    memset(d->ten_J + (size_t)t * nv, 0, sizeof(double) * nv);
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
      int j = m->wrap_objid[w];
      len += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[j]];
      d->ten_J[(size_t)t * nv + m->jnt_dofadr[j]] = m->wrap_prm[w];
    }
    d->ten_length[t] = len;
  }
  double* jp = d->jacbuf;
  double* jr = jp + 3 * nv;
  for (int a = 0; a < m->nu; a++) {
    double* mom = d->actuator_moment + (size_t)a * nv;
    memset(mom, 0, sizeof(double) * nv);
    const double* gear = m->actuator_gear + 6 * a;
    int id = m->actuator_trnid[2 * a];
    if (m->actuator_trntype[a] == TRN_JOINT) {
      d->actuator_length[a] = gear[0] * d->qpos[m->jnt_qposadr[id]];
      mom[m->jnt_dofadr[id]] = gear[0];
    } else { /* site transmission, no refsite: wrench = gear expressed in the site frame */
      double f[3], tq[3];
      d->actuator_length[a] = 0;
      mulmatvec3(f, d->site_xmat + 9 * id, gear); mulmatvec3(tq, d->site_xmat + 9 * id, gear + 3);
      jac_point(m, d, m->site_bodyid[id], d->site_xpos + 3 * id, jp, jr);
      for (int i = 0; i < nv; i++)
        mom[i] = jp[i] * f[0] + jp[nv + i] * f[1] + jp[2 * nv + i] * f[2] + jr[i] * tq[0] + jr[nv + i] * tq[1] + jr[2 * nv + i] * tq[2];
    }
  }
}

/* A4  composite rigid body + dense factor (mj_crb, mj_factorM) */
static void crb_factor(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_CRB);
  int nv = m->nv, nb = m->nbody;
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * nb);
  for (int b = nb - 1; b > 0; b--) { int p = m->body_parentid[b]; if (p > 0) for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k]; }
  memset(d->qM, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += d->cdof[6 * j + k] * buf[k];
      d->qM[i * nv + j] = d->qM[j * nv + i] = v;
    }
    d->qM[i * nv + i] += m->dof_armature[i];
  }
  memcpy(d->qL, d->qM, sizeof(double) * nv * nv);
  chol_factor(d->qL, nv);
}

/* Jacobian of a world point attached to `body` (mj_jac): 3 x nv each, from cdof */
static void jac_point(const mjoModel* m, const mjoData* d, int body, const double* point, double* jacp, double* jacr) {
  int nv = m->nv;
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
  if (body <= 0) return;
  double off[3];
  const double* rc = d->subtree_com + 3 * m->body_rootid[body];
  for (int k = 0; k < 3; k++) off[k] = point[k] - rc[k];
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* c = d->cdof + 6 * i;
    if (jacr) { jacr[i] = c[0]; jacr[nv + i] = c[1]; jacr[2 * nv + i] = c[2]; }
    if (jacp) {
      double t[3];
      cross3(t, c, off);
      jacp[i] = c[3] + t[0]; jacp[nv + i] = c[4] + t[1]; jacp[2 * nv + i] = c[5] + t[2];
    }
  }
}

/* ------------------------------------------------------------------------- */
/* A5  collision: static pair list + narrow phase                              */
/* ------------------------------------------------------------------------- */
static void make_frame(double* f) {
  normalize3(f);
  if (norm3(f + 3) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double t = dot3(f, f + 3);
  f[3] -= t * f[0]; f[4] -= t * f[1]; f[5] -= t * f[2];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

typedef struct { double dist, pos[3], frame[9]; } RawCon;

static int sphere_sphere(const double* p1, double r1, const double* p2, double r2, double margin, RawCon* c) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double cd = norm3(dif);
  double dist = cd - r1 - r2;
  if (dist > margin) return 0;
  memset(c->frame, 0, sizeof c->frame);
  if (cd < MINVAL) { c->frame[0] = 1; } else { c->frame[0] = dif[0] / cd; c->frame[1] = dif[1] / cd; c->frame[2] = dif[2] / cd; }
  c->dist = dist;
  addscl3(c->pos, p1, c->frame, r1 + 0.5 * dist);
  return 1;
}
static int plane_sphere(const double* pp, const double* pm, const double* sp, double r, double margin, RawCon* c) {
  double n[3] = {pm[2], pm[5], pm[8]}, dif[3] = {sp[0] - pp[0], sp[1] - pp[1], sp[2] - pp[2]};
  double dist = dot3(dif, n) - r;
  if (dist > margin) return 0;
  memset(c->frame, 0, sizeof c->frame);
  copy3(c->frame, n);
  c->dist = dist;
  addscl3(c->pos, sp, n, -(r + 0.5 * dist));
  return 1;
}
static int plane_capsule(const double* pp, const double* pm, const double* cp, const double* cm, const double* size, double margin, RawCon* c) {
  double axis[3] = {cm[2], cm[5], cm[8]}, end[3];
  int n = 0;
  for (int s = 0; s < 2; s++) {
    addscl3(end, cp, axis, s == 0 ? size[1] : -size[1]);
    if (plane_sphere(pp, pm, end, size[0], margin, c + n)) { copy3(c[n].frame + 3, axis); n++; }  /* y-hint = capsule axis */
  }
  return n;
}
static int plane_box(const double* pp, const double* pm, const double* bp, const double* bm, const double* size, double margin, RawCon* c) {
  double n[3] = {pm[2], pm[5], pm[8]}, dif[3] = {bp[0] - pp[0], bp[1] - pp[1], bp[2] - pp[2]};
  double dist = dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double v[3] = {(i & 1 ? size[0] : -size[0]), (i & 2 ? size[1] : -size[1]), (i & 4 ? size[2] : -size[2])}, corner[3];
    mulmatvec3(corner, bm, v);
    double ld = dot3(n, corner);
    if (dist + ld > margin || ld > 0) continue;
    c[cnt].dist = dist + ld;
    memset(c[cnt].frame, 0, sizeof c[cnt].frame); copy3(c[cnt].frame, n);
    double w[3]; addscl3(w, corner, bp, 1.0);
    addscl3(c[cnt].pos, w, n, -0.5 * c[cnt].dist);
    if (++cnt >= 4) break;
  }
  return cnt;
}
static int plane_ellipsoid(const double* pp, const double* pm, const double* ep, const double* em, const double* size, double margin, RawCon* c) {
  /* support point of the ellipsoid in direction -n (mjc_PlaneConvex restricted to one support point) */
  double n[3] = {pm[2], pm[5], pm[8]}, dl[3], s[3], w[3];
  double nn[3] = {-n[0], -n[1], -n[2]};
  mulmatTvec3(dl, em, nn);
  double den = sqrt(size[0] * size[0] * dl[0] * dl[0] + size[1] * size[1] * dl[1] * dl[1] + size[2] * size[2] * dl[2] * dl[2]);
  if (den < MINVAL) den = MINVAL;
  for (int k = 0; k < 3; k++) s[k] = size[k] * size[k] * dl[k] / den;
  mulmatvec3(w, em, s); addscl3(w, w, ep, 1.0);
  double dif[3] = {w[0] - pp[0], w[1] - pp[1], w[2] - pp[2]};
  double dist = dot3(dif, n);
  if (dist > margin) return 0;
  memset(c->frame, 0, sizeof c->frame); copy3(c->frame, n);
  c->dist = dist;
  addscl3(c->pos, w, n, -0.5 * dist);
  return 1;
}
static int sphere_capsule(const double* sp, double sr, const double* cp, const double* cm, const double* size, double margin, RawCon* c) {
  double axis[3] = {cm[2], cm[5], cm[8]}, dif[3] = {sp[0] - cp[0], sp[1] - cp[1], sp[2] - cp[2]}, pt[3];
  double x = dot3(axis, dif);
  if (x > size[1]) x = size[1]; if (x < -size[1]) x = -size[1];
  addscl3(pt, cp, axis, x);
  return sphere_sphere(sp, sr, pt, size[0], margin, c);
}
/* mjraw_CapsuleCapsule [MJ-KNOWLEDGE: engine_collision_primitive.c]: nearest points of the two axis segments -> sphere-sphere.
 * Parallel axes (|det| < mjMINVAL): MuJoCo tests the four end caps in the order (+end of 1, -end of 1, +end of 2, -end of 2);
 * every end whose projection onto the OTHER segment falls inside it gives a sphere-sphere contact, at most two in total. */
static int capsule_capsule(const double* p1, const double* m1, const double* s1, const double* p2, const double* m2, const double* s2, double margin, RawCon* c) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  double det = ma * mc - mb * mb, x1, x2, v1[3], v2[3];
  if (fabs(det) >= MINVAL) {
    x1 = (mc * u - mb * v) / det; x2 = (ma * v - mb * u) / det;
    if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb * s1[1]) / mc; }
    else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v + mb * s1[1]) / mc; }
    if (x2 > s2[1]) { x2 = s2[1]; x1 = (u - mb * s2[1]) / ma; if (x1 > s1[1]) x1 = s1[1]; if (x1 < -s1[1]) x1 = -s1[1]; }
    else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = (u + mb * s2[1]) / ma; if (x1 > s1[1]) x1 = s1[1]; if (x1 < -s1[1]) x1 = -s1[1]; }
    addscl3(v1, p1, a1, x1); addscl3(v2, p2, a2, x2);
    return sphere_sphere(v1, s1[0], v2, s2[0], margin, c);
  }
  int n = 0;
  for (int k = 0; k < 4 && n < 2; k++) {
    if (k < 2) { x1 = k == 0 ? s1[1] : -s1[1]; x2 = (v - mb * x1) / mc; if (x2 < -s2[1] || x2 > s2[1]) continue; }
    else { x2 = k == 2 ? s2[1] : -s2[1]; x1 = (u - mb * x2) / ma; if (x1 < -s1[1] || x1 > s1[1]) continue; }
    addscl3(v1, p1, a1, x1); addscl3(v2, p2, a2, x2);
    n += sphere_sphere(v1, s1[0], v2, s2[0], margin, c + n);
  }
  return n;
}

static void collision(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_COLL);
  d->ncon = 0; d->ncon_dropped = 0;
  RawCon rc[4];
  int cap = m->nconmax > 0 ? m->nconmax : m->ncon_alloc;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], t1 = m->geom_type[g1], t2 = m->geom_type[g2], n = 0;
    const double *p1 = d->geom_xpos + 3 * g1, *m1 = d->geom_xmat + 9 * g1, *s1 = m->geom_size + 3 * g1;
    const double *p2 = d->geom_xpos + 3 * g2, *m2 = d->geom_xmat + 9 * g2, *s2 = m->geom_size + 3 * g2;
    double margin = m->pair_margin[p];
    if (t1 == G_PLANE && t2 == G_SPHERE) n = plane_sphere(p1, m1, p2, s2[0], margin, rc);
    else if (t1 == G_PLANE && t2 == G_CAPSULE) n = plane_capsule(p1, m1, p2, m2, s2, margin, rc);
    else if (t1 == G_PLANE && t2 == G_BOX) n = plane_box(p1, m1, p2, m2, s2, margin, rc);
    else if (t1 == G_PLANE && t2 == G_ELLIPSOID) n = plane_ellipsoid(p1, m1, p2, m2, s2, margin, rc);
    else if (t1 == G_SPHERE && t2 == G_SPHERE) n = sphere_sphere(p1, s1[0], p2, s2[0], margin, rc);
    else if (t1 == G_SPHERE && t2 == G_CAPSULE) n = sphere_capsule(p1, s1[0], p2, m2, s2, margin, rc);
    else if (t1 == G_CAPSULE && t2 == G_CAPSULE) n = capsule_capsule(p1, m1, s1, p2, m2, s2, margin, rc);
    for (int k = 0; k < n; k++) {
      if (d->ncon >= cap) { d->ncon_dropped++; continue; }
      mjoContact* c = d->contact + d->ncon++;
      c->dist = rc[k].dist; copy3(c->pos, rc[k].pos); memcpy(c->frame, rc[k].frame, 72);
      make_frame(c->frame);
      c->dim = m->pair_condim[p]; c->geom1 = g1; c->geom2 = g2;
      memcpy(c->friction, m->pair_friction + 5 * p, 40); memcpy(c->solref, m->pair_solref + 2 * p, 16); memcpy(c->solimp, m->pair_solimp + 5 * p, 40);
      c->includemargin = m->pair_margin[p] - m->pair_gap[p];
      c->efc_address = -1;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* A6  constraint assembly (mj_makeConstraint + impedance / reference)        */
/* ------------------------------------------------------------------------- */
static int add_row(const mjoModel* m, mjoData* d, int type, int id, double pos, double margin, const double* solref, const double* solimp, double diagApprox) {
  int i = d->nefc++;
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin; d->efc_diagApprox[i] = diagApprox;
  /* impedance (getimpedance) */
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4], imp;
  if (dmin < MINIMP) dmin = MINIMP; if (dmin > MAXIMP) dmin = MAXIMP;
  if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
  if (width < 0) width = 0; if (mid < MINIMP) mid = MINIMP; if (mid > MAXIMP) mid = MAXIMP; if (power < 1) power = 1;
  if (dmin == dmax || width <= MINVAL) imp = 0.5 * (dmin + dmax);
  else {
    double x = fabs(pos - margin) / width;
    if (x >= 1) imp = dmax;
    else if (x <= 0) imp = dmin;
    else {
      double y;
      if (power == 1) y = x;
      else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
      else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
      imp = dmin + y * (dmax - dmin);
    }
  }
  /* stiffness / damping from solref (mj_makeImpedance, refsafe enabled) */
  double K, B;
  if (solref[0] > 0) {
    double tc = solref[0], dr = solref[1];
    if (tc < 2 * m->timestep) tc = 2 * m->timestep;
    double dm = solimp[1]; if (dm < MINIMP) dm = MINIMP; if (dm > MAXIMP) dm = MAXIMP;
    K = 1 / fmax(MINVAL, dm * dm * tc * tc * dr * dr);
    B = 2 / fmax(MINVAL, dm * tc);
  } else {
    double dm = solimp[1]; if (dm < MINIMP) dm = MINIMP; if (dm > MAXIMP) dm = MAXIMP;
    K = -solref[0] / fmax(MINVAL, dm * dm); B = -solref[1] / fmax(MINVAL, dm);
  }
  d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B; d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = 0;
  d->efc_R[i] = fmax(MINVAL, (1 - imp) * diagApprox / imp);
  return i;
}

static void make_constraint(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_CONS);
  int nv = m->nv;
  d->nefc = 0; d->nefc_dropped = 0;
  int cap = m->nefcmax > 0 ? m->nefcmax : m->nefc_alloc;
  /* joint limits (hinge / slide) */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || (m->jnt_type[j] != JNT_HINGE && m->jnt_type[j] != JNT_SLIDE)) continue;
    double value = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
      if (dist < m->jnt_margin[j]) {
        if (d->nefc >= cap) { d->nefc_dropped++; continue; }
        int da = m->jnt_dofadr[j];
        int i = add_row(m, d, EFC_LIMIT_JOINT, j, dist, m->jnt_margin[j], m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, m->dof_invweight0[da]);
        memset(d->efc_J + (size_t)i * nv, 0, sizeof(double) * nv);
        d->efc_J[(size_t)i * nv + da] = -side;
      }
    }
  }
  /* tendon limits */
  for (int t = 0; t < m->ntendon; t++) {
    if (!m->tendon_limited[t]) continue;
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->tendon_range[2 * t + (side + 1) / 2] - d->ten_length[t]);
      if (dist < m->tendon_margin[t]) {
        if (d->nefc >= cap) { d->nefc_dropped++; continue; }
        int i = add_row(m, d, EFC_LIMIT_TENDON, t, dist, m->tendon_margin[t], m->tendon_solref + 2 * t, m->tendon_solimp + 5 * t, m->tendon_invweight0[t]);
        for (int k = 0; k < nv; k++) d->efc_J[(size_t)i * nv + k] = -side * d->ten_J[(size_t)t * nv + k];
      }
    }
  }
  /* contacts: frictionless (condim 1) or pyramidal (condim 3) */
  double* jp1 = d->jacbuf;
  double* jp2 = jp1 + 3 * nv;
  int overflow = 0;
  for (int c = 0; c < d->ncon; c++) {
    mjoContact* con = d->contact + c;
    con->efc_address = -1;
    if (con->dist >= con->includemargin) continue;
    int rows = con->dim == 1 ? 1 : 2 * (con->dim - 1);
    if (overflow || d->nefc + rows > cap) { overflow = 1; d->nefc_dropped += rows; continue; }
    int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
    jac_point(m, d, b1, con->pos, jp1, NULL);
    jac_point(m, d, b2, con->pos, jp2, NULL);
    double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    con->efc_address = d->nefc;
    if (con->dim == 1) {
      int i = add_row(m, d, EFC_CONTACT_FRICTIONLESS, c, con->dist, con->includemargin, con->solref, con->solimp, tran);
      for (int k = 0; k < nv; k++) {
        double v = 0;
        for (int a = 0; a < 3; a++) v += con->frame[a] * (jp2[a * nv + k] - jp1[a * nv + k]);
        d->efc_J[(size_t)i * nv + k] = v;
      }
    } else {
      int first = -1;
      for (int r = 0; r < rows; r++) {
        double mu = con->friction[r / 2], sgn = (r & 1) ? -1.0 : 1.0;
        const double* tdir = con->frame + 3 * (1 + r / 2);
        int i = add_row(m, d, EFC_CONTACT_PYRAMIDAL, c, con->dist, con->includemargin, con->solref, con->solimp, tran + mu * mu * tran);
        if (first < 0) first = i;
        for (int k = 0; k < nv; k++) {
          double v = 0;
          for (int a = 0; a < 3; a++) v += (con->frame[a] + sgn * mu * tdir[a]) * (jp2[a * nv + k] - jp1[a * nv + k]);
          d->efc_J[(size_t)i * nv + k] = v;
        }
      }
      /* pyramidal: every edge gets Rpy = 2 mu^2 R(first edge), mu = friction[0] (impratio = 1) */
      double Rpy = 2 * con->friction[0] * con->friction[0] * d->efc_R[first];
      if (Rpy < MINVAL) Rpy = MINVAL;
      for (int r = 0; r < rows; r++) d->efc_R[first + r] = Rpy;
    }
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
}

/* mj_referenceConstraint: efc_vel = J qvel, aref = -B vel - K imp (pos - margin) */
static void reference_constraint(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_CONS);
  int nv = m->nv;
  for (int i = 0; i < d->nefc; i++) {
    double v = 0;
    for (int k = 0; k < nv; k++) v += d->efc_J[(size_t)i * nv + k] * d->qvel[k];
    d->efc_vel[i] = v;
    d->efc_aref[i] = -d->efc_KBIP[4 * i + 1] * v - d->efc_KBIP[4 * i] * d->efc_KBIP[4 * i + 2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* ------------------------------------------------------------------------- */
/* A7  velocity stage: com velocities, passive forces, bias forces             */
/* ------------------------------------------------------------------------- */
static void com_vel(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_VEL);
  memset(d->cvel, 0, sizeof(double) * 6);
  for (int b = 1; b < m->nbody; b++) {
    double cvel[6], tmp[6];
    memcpy(cvel, d->cvel + 6 * m->body_parentid[b], 48);
    int da = m->body_dofadr[b], dn = m->body_dofnum[b];
    for (int j = da; j < da + dn;) {
      int jt = m->jnt_type[m->dof_jntid[j]];
      if (jt == JNT_FREE) {
        memset(d->cdof_dot + 6 * j, 0, sizeof(double) * 18);
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (j + k) + c] * d->qvel[j + k];
        j += 3;
        for (int k = 0; k < 3; k++) cross_motion(d->cdof_dot + 6 * (j + k), cvel, d->cdof + 6 * (j + k));
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (j + k) + c] * d->qvel[j + k];
        j += 3;
      } else {
        cross_motion(tmp, cvel, d->cdof + 6 * j);
        memcpy(d->cdof_dot + 6 * j, tmp, 48);
        for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * j + c] * d->qvel[j];
        j++;
      }
    }
    memcpy(d->cvel + 6 * b, cvel, 48);
  }
}

/* apply a force/torque at a world point of `body` to generalized forces (mj_applyFT) */
static void apply_ft(const mjoModel* m, const mjoData* d, const double* force, const double* torque, const double* point, int body, double* qfrc) {
  int nv = m->nv;
  double* jp = d->jacbuf;
  double* jr = jp + 3 * nv;
  jac_point(m, d, body, point, jp, jr);
  for (int i = 0; i < nv; i++)
    qfrc[i] += jp[i] * force[0] + jp[nv + i] * force[1] + jp[2 * nv + i] * force[2] + jr[i] * torque[0] + jr[nv + i] * torque[1] + jr[2 * nv + i] * torque[2];
}

static void passive(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_PASSIVE);
  int nv = m->nv;
  memset(d->qfrc_passive, 0, sizeof(double) * nv);
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_stiffness[j] == 0) continue;
    if (m->jnt_type[j] == JNT_HINGE || m->jnt_type[j] == JNT_SLIDE) {
      int qa = m->jnt_qposadr[j];
      d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
    }
  }
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
  /* fluid: inertia-box model (mj_inertiaBoxFluidModel) */
  if (m->density > 0 || m->viscosity > 0) {
    for (int b = 1; b < m->nbody; b++) {
      double mass = m->body_mass[b];
      if (mass < MINVAL) continue;
      const double* I = m->body_inertia + 3 * b;
      double box[3] = {sqrt(fmax(MINVAL, I[1] + I[2] - I[0]) / mass * 6.0), sqrt(fmax(MINVAL, I[0] + I[2] - I[1]) / mass * 6.0), sqrt(fmax(MINVAL, I[0] + I[1] - I[2]) / mass * 6.0)};
      /* 6D velocity at the body com in the inertial frame orientation (mj_objectVelocity, flg_local) */
      const double* cv = d->cvel + 6 * b;
      double off[3], lin[3], t[3], lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0}, bfrc[6];
      const double* rc = d->subtree_com + 3 * m->body_rootid[b];
      for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * b + k] - rc[k];
      cross3(t, cv, off);                       /* w x (p - ref) */
      for (int k = 0; k < 3; k++) lin[k] = cv[3 + k] + t[k];
      mulmatTvec3(lvel, d->ximat + 9 * b, cv);
      mulmatTvec3(lvel + 3, d->ximat + 9 * b, lin);
      if (m->viscosity > 0) {
        double diam = (box[0] + box[1] + box[2]) / 3.0;
        for (int k = 0; k < 3; k++) { lfrc[k] = -PI * diam * diam * diam * m->viscosity * lvel[k]; lfrc[3 + k] = -3.0 * PI * diam * m->viscosity * lvel[3 + k]; }
      }
      if (m->density > 0) {
        lfrc[3] -= 0.5 * m->density * box[1] * box[2] * fabs(lvel[3]) * lvel[3];
        lfrc[4] -= 0.5 * m->density * box[0] * box[2] * fabs(lvel[4]) * lvel[4];
        lfrc[5] -= 0.5 * m->density * box[0] * box[1] * fabs(lvel[5]) * lvel[5];
        lfrc[0] -= m->density * box[0] * (pow(box[1], 4) + pow(box[2], 4)) * fabs(lvel[0]) * lvel[0] / 64.0;
        lfrc[1] -= m->density * box[1] * (pow(box[0], 4) + pow(box[2], 4)) * fabs(lvel[1]) * lvel[1] / 64.0;
        lfrc[2] -= m->density * box[2] * (pow(box[0], 4) + pow(box[1], 4)) * fabs(lvel[2]) * lvel[2] / 64.0;
      }
      mulmatvec3(bfrc, d->ximat + 9 * b, lfrc); mulmatvec3(bfrc + 3, d->ximat + 9 * b, lfrc + 3);
      apply_ft(m, d, bfrc + 3, bfrc, d->xipos + 3 * b, b, d->qfrc_passive);
    }
  }
}

/* mj_rne with flg_acc = 0: Coriolis/centrifugal + gravity */
static void rne_bias(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_RNE);
  int nb = m->nbody, nv = m->nv;
  memset(d->cacc, 0, sizeof(double) * 6);
  d->cacc[3] = -m->gravity[0]; d->cacc[4] = -m->gravity[1]; d->cacc[5] = -m->gravity[2];
  memset(d->cfrc, 0, sizeof(double) * 6);
  for (int b = 1; b < nb; b++) {
    int da = m->body_dofadr[b], dn = m->body_dofnum[b];
    double* ca = d->cacc + 6 * b;
    memcpy(ca, d->cacc + 6 * m->body_parentid[b], 48);
    for (int j = da; j < da + dn; j++) for (int c = 0; c < 6; c++) ca[c] += d->cdof_dot[6 * j + c] * d->qvel[j];
    double t[6], t1[6];
    mul_inert_vec(d->cfrc + 6 * b, d->cinert + 10 * b, ca);
    mul_inert_vec(t, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(t1, d->cvel + 6 * b, t);
    for (int c = 0; c < 6; c++) d->cfrc[6 * b + c] += t1[c];
  }
  for (int b = nb - 1; b > 0; b--) { int p = m->body_parentid[b]; if (p > 0) for (int c = 0; c < 6; c++) d->cfrc[6 * p + c] += d->cfrc[6 * b + c]; }
  for (int i = 0; i < nv; i++) {
    double v = 0;
    for (int c = 0; c < 6; c++) v += d->cdof[6 * i + c] * d->cfrc[6 * m->dof_bodyid[i] + c];
    d->qfrc_bias[i] = v;
  }
}

/* A8  actuation (mj_fwdActuation) */
static void actuation(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_ACT);
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, sizeof(double) * nv);
  for (int a = 0; a < m->nu; a++) {
    const double* mom = d->actuator_moment + (size_t)a * nv;
    double vel = 0;
    for (int i = 0; i < nv; i++) vel += mom[i] * d->qvel[i];
    d->actuator_velocity[a] = vel;
    int grp = m->actuator_group[a];
    if (grp >= 0 && grp < 31 && ((m->disableactuator >> grp) & 1)) { d->actuator_force[a] = 0; continue; }
    double ctrl = d->ctrl[a];
    if (m->actuator_ctrllimited[a]) { if (ctrl < m->actuator_ctrlrange[2 * a]) ctrl = m->actuator_ctrlrange[2 * a]; if (ctrl > m->actuator_ctrlrange[2 * a + 1]) ctrl = m->actuator_ctrlrange[2 * a + 1]; }
    double force = m->actuator_gainprm[3 * a] * ctrl;
    if (m->actuator_biastype[a] == 1)
      force += m->actuator_biasprm[3 * a] + m->actuator_biasprm[3 * a + 1] * d->actuator_length[a] + m->actuator_biasprm[3 * a + 2] * vel;
    if (m->actuator_forcelimited[a]) { if (force < m->actuator_forcerange[2 * a]) force = m->actuator_forcerange[2 * a]; if (force > m->actuator_forcerange[2 * a + 1]) force = m->actuator_forcerange[2 * a + 1]; }
    d->actuator_force[a] = force;
    for (int i = 0; i < nv; i++) d->qfrc_actuator[i] += mom[i] * force;
  }
}

/* A9  unconstrained acceleration (mj_fwdAcceleration) */
static void acceleration(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_ACC);
  int nv = m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * nv);
  chol_solve(d->qL, nv, d->qacc_smooth);
}

/* ------------------------------------------------------------------------- */
/* A10  constraint solve: Newton, pyramidal/frictionless/limit rows (one-sided quadratic) */
/* ------------------------------------------------------------------------- */
static double constraint_update(const mjoModel* m, mjoData* d, const double* jar, int* changed) {
  double cost = 0;
  int ch = 0;
  for (int i = 0; i < d->nefc; i++) {
    int act = jar[i] < 0;
    if (act != d->efc_active[i]) ch = 1;
    d->efc_active[i] = act;
    if (act) { d->efc_force[i] = -d->efc_D[i] * jar[i]; cost += 0.5 * d->efc_D[i] * jar[i] * jar[i]; }
    else d->efc_force[i] = 0;
  }
  (void)m;
  if (changed) *changed = ch;
  return cost;
}
static void mul_M(const mjoModel* m, const mjoData* d, double* res, const double* v) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * v[k]; res[i] = s; }
}
static void mul_J(const mjoModel* m, const mjoData* d, double* res, const double* v) {
  int nv = m->nv;
  for (int i = 0; i < d->nefc; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * v[k]; res[i] = s; }
}
static double total_cost(const mjoModel* m, mjoData* d, const double* qacc, double* Ma, double* jar, int* changed) {
  int nv = m->nv;
  mul_M(m, d, Ma, qacc);
  mul_J(m, d, jar, qacc);
  for (int i = 0; i < d->nefc; i++) jar[i] -= d->efc_aref[i];
  double cost = constraint_update(m, d, jar, changed), gauss = 0;
  for (int i = 0; i < nv; i++) gauss += (Ma[i] - d->qfrc_smooth[i]) * (qacc[i] - d->qacc_smooth[i]);
  return cost + 0.5 * gauss;
}
static void newton_direction(const mjoModel* m, mjoData* d) {
  int nv = m->nv;
  MJO_PHASE(MJO_PH_SOL_HESS);
  for (int i = 0; i < nv; i++) {
    double g = d->s_Ma[i] - d->qfrc_smooth[i];
    for (int r = 0; r < d->nefc; r++) g -= d->efc_J[(size_t)r * nv + i] * d->efc_force[r];
    d->s_grad[i] = g;
  }
  memcpy(d->qH, d->qM, sizeof(double) * nv * nv);
  for (int r = 0; r < d->nefc; r++) {
    if (!d->efc_active[r]) continue;
    const double* J = d->efc_J + (size_t)r * nv;
    double D = d->efc_D[r];
    for (int i = 0; i < nv; i++) { if (J[i] == 0) continue; double s = D * J[i]; for (int k = 0; k <= i; k++) d->qH[i * nv + k] += s * J[k]; }
  }
  for (int i = 0; i < nv; i++) for (int k = i + 1; k < nv; k++) d->qH[i * nv + k] = d->qH[k * nv + i];
  MJO_PHASE(MJO_PH_SOL_FACTOR);
  chol_factor(d->qH, nv);
  memcpy(d->s_Mgrad, d->s_grad, sizeof(double) * nv);
  chol_solve(d->qH, nv, d->s_Mgrad);
  for (int i = 0; i < nv; i++) d->s_search[i] = -d->s_Mgrad[i];
  MJO_PHASE(MJO_PH_SOL_UPDATE);
}
/* exact minimiser of the 1-D convex piecewise-quadratic cost along `search` */
static double line_search(const mjoModel* m, mjoData* d) {
  int nv = m->nv;
  double g1 = 0, g2 = 0;
  for (int i = 0; i < nv; i++) { g1 += d->s_search[i] * (d->s_Ma[i] - d->qfrc_smooth[i]); g2 += d->s_search[i] * d->s_Mv[i]; }
  double alpha = 0, lo = 0, hi = -1;
  for (int it = 0; it < 50; it++) {
    double d1 = g1 + alpha * g2, d2 = g2;
    for (int i = 0; i < d->nefc; i++) {
      double x = d->efc_jar[i] + alpha * d->s_jv[i];
      if (x < 0) { d1 += d->efc_D[i] * x * d->s_jv[i]; d2 += d->efc_D[i] * d->s_jv[i] * d->s_jv[i]; }
    }
    if (it == 0 && d1 >= 0) return 0;           /* not a descent direction */
    if (d2 < MINVAL) break;
    if (fabs(d1) < 1e-14 * (fabs(g1) + MINVAL)) break;
    if (d1 < 0) lo = alpha; else hi = alpha;
    double an = alpha - d1 / d2;
    if (an <= lo || (hi >= 0 && an >= hi)) an = hi >= 0 ? 0.5 * (lo + hi) : 2 * alpha + 1e-3;
    if (an == alpha) break;
    alpha = an;
  }
  return alpha;
}

static void solve_constraints(const mjoModel* m, mjoData* d) {
  int nv = m->nv, nefc = d->nefc;
  MJO_PHASE(MJO_PH_SOL_SETUP);
  d->solver_niter = 0;
  if (nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    return;
  }
  /* warmstart(): best of (qacc_warmstart, qacc_smooth) */
  for (int i = 0; i < nefc; i++) d->efc_active[i] = 0;
  double cost_ws = total_cost(m, d, d->qacc_warmstart, d->s_Ma, d->efc_jar, NULL);
  double cost_sm = total_cost(m, d, d->qacc_smooth, d->s_Ma, d->efc_jar, NULL);
  if (cost_ws < cost_sm) memcpy(d->qacc, d->qacc_warmstart, sizeof(double) * nv);
  else memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
  double cost = total_cost(m, d, d->qacc, d->s_Ma, d->efc_jar, NULL);
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  newton_direction(m, d);
  for (int iter = 0; iter < m->iterations; iter++) {
    double gn = 0;
    for (int i = 0; i < nv; i++) gn += d->s_grad[i] * d->s_grad[i];
    if (scale * sqrt(gn) < m->tolerance) break;
    MJO_PHASE(MJO_PH_SOL_MATVEC);
    mul_M(m, d, d->s_Mv, d->s_search);
    mul_J(m, d, d->s_jv, d->s_search);
    MJO_PHASE(MJO_PH_SOL_LS);
    double alpha = line_search(m, d);
    MJO_PHASE(MJO_PH_SOL_UPDATE);
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) { d->qacc[i] += alpha * d->s_search[i]; d->s_Ma[i] += alpha * d->s_Mv[i]; }
    for (int i = 0; i < nefc; i++) d->efc_jar[i] += alpha * d->s_jv[i];
    double old = cost, gauss = 0;
    cost = constraint_update(m, d, d->efc_jar, NULL);
    for (int i = 0; i < nv; i++) gauss += (d->s_Ma[i] - d->qfrc_smooth[i]) * (d->qacc[i] - d->qacc_smooth[i]);
    cost += 0.5 * gauss;
    d->solver_niter = iter + 1;
    newton_direction(m, d);
    if (scale * (old - cost) < m->tolerance) break;
  }
  d->solver_cost = cost;
  for (int i = 0; i < nv; i++) { double s = 0; for (int r = 0; r < nefc; r++) s += d->efc_J[(size_t)r * nv + i] * d->efc_force[r]; d->qfrc_constraint[i] = s; }
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}

/* A12 sensors: jointpos, gyro, framequat, accelerometer (needs cacc with qacc: mj_rnePostConstraint) */
static void sensors(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_SENSORS);
  int need_acc = 0;
  for (int s = 0; s < m->nsensor; s++) if (m->sensor_type[s] == SENS_ACCEL) need_acc = 1;
  if (need_acc) {   /* cacc <- [0; -g] + sum over ancestors of (cdof_dot qvel + cdof qacc) */
    memset(d->cacc, 0, sizeof(double) * 6);
    d->cacc[3] = -m->gravity[0]; d->cacc[4] = -m->gravity[1]; d->cacc[5] = -m->gravity[2];
    for (int b = 1; b < m->nbody; b++) {
      double* ca = d->cacc + 6 * b;
      memcpy(ca, d->cacc + 6 * m->body_parentid[b], 48);
      for (int j = m->body_dofadr[b]; j < m->body_dofadr[b] + m->body_dofnum[b]; j++)
        for (int c = 0; c < 6; c++) ca[c] += d->cdof_dot[6 * j + c] * d->qvel[j] + d->cdof[6 * j + c] * d->qacc[j];
    }
  }
  for (int s = 0; s < m->nsensor; s++) {
    double* out = d->sensordata + m->sensor_adr[s];
    int id = m->sensor_objid[s];
    switch (m->sensor_type[s]) {
      case SENS_JOINTPOS: out[0] = d->qpos[m->jnt_qposadr[id]]; break;
      case SENS_GYRO: mulmatTvec3(out, d->site_xmat + 9 * id, d->cvel + 6 * m->site_bodyid[id]); break;
      case SENS_FRAMEQUAT: { double q[4]; quat_mul(q, d->xquat + 4 * m->site_bodyid[id], m->site_quat + 4 * id); quat_normalize(q); memcpy(out, q, 32); } break;
      case SENS_ACCEL: {   /* mj_objectAcceleration(site, local): transport com-based cvel / cacc to the site, rotate, add w x v */
        int b = m->site_bodyid[id];
        const double *R = d->site_xmat + 9 * id, *cv = d->cvel + 6 * b, *ca = d->cacc + 6 * b;
        const double* rc = d->subtree_com + 3 * m->body_rootid[b];
        double dif[3] = {d->site_xpos[3 * id] - rc[0], d->site_xpos[3 * id + 1] - rc[1], d->site_xpos[3 * id + 2] - rc[2]};
        double t[3], vlin[3], alin[3], wl[3], vl[3], al[3], cr[3];
        cross3(t, cv, dif); for (int k = 0; k < 3; k++) vlin[k] = cv[3 + k] + t[k];
        cross3(t, ca, dif); for (int k = 0; k < 3; k++) alin[k] = ca[3 + k] + t[k];
        mulmatTvec3(wl, R, cv); mulmatTvec3(vl, R, vlin); mulmatTvec3(al, R, alin);
        cross3(cr, wl, vl);
        out[0] = al[0] + cr[0]; out[1] = al[1] + cr[1]; out[2] = al[2] + cr[2];
      } break;
      default: out[0] = 0;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* mj_forward                                                                   */
/* ------------------------------------------------------------------------- */
/* Precision study (scripts/precision_study.py): the outputs of the phases named in m->round_mask are rounded to fp32
 * where they are handed to the next phase - an fp32-STORAGE model of a mixed-precision kernel (the arithmetic inside a phase stays
 * float64, so the error it shows is a lower bound for a real fp32 phase).  round_mask is 0 in every other use of the oracle. */
static void r32(double* x, long n) { for (long i = 0; i < n; i++) x[i] = (double)(float)x[i]; }
enum { RM_KIN = 1, RM_COM = 2, RM_CRB = 4, RM_CONS = 8, RM_VEL = 16, RM_RNE = 32, RM_FRC = 64, RM_ACC = 128, RM_SOLVER = 256, RM_STATE = 512 };

void mjo_forward(const mjoModel* m, mjoData* d) {
  const int rm = m->round_mask, nv = m->nv, nb = m->nbody;
  kinematics(m, d);
  if (rm & RM_KIN) {
    r32(d->xpos, 3L * nb); r32(d->xquat, 4L * nb); r32(d->xmat, 9L * nb); r32(d->xipos, 3L * nb); r32(d->ximat, 9L * nb);
    r32(d->xanchor, 3L * m->njnt); r32(d->xaxis, 3L * m->njnt); r32(d->geom_xpos, 3L * m->ngeom); r32(d->geom_xmat, 9L * m->ngeom);
    r32(d->site_xpos, 3L * m->nsite); r32(d->site_xmat, 9L * m->nsite);
  }
  com_pos(m, d);
  if (rm & RM_COM) { r32(d->subtree_com, 3L * nb); r32(d->cinert, 10L * nb); r32(d->cdof, 6L * nv); }
  tendon_transmission(m, d);
  crb_factor(m, d);
  if (rm & RM_CRB) { r32(d->crb, 10L * nb); r32(d->qM, (long)nv * nv); memcpy(d->qL, d->qM, sizeof(double) * nv * nv); chol_factor(d->qL, nv); r32(d->qL, (long)nv * nv); }
  collision(m, d);
  make_constraint(m, d);
  com_vel(m, d);
  if (rm & RM_VEL) { r32(d->cvel, 6L * nb); r32(d->cdof_dot, 6L * nv); }
  passive(m, d);
  reference_constraint(m, d);
  if (rm & RM_CONS) {
    for (int i = 0; i < d->ncon; i++) { r32(&d->contact[i].dist, 1); r32(d->contact[i].pos, 3); r32(d->contact[i].frame, 9); }
    r32(d->efc_J, (long)d->nefc * nv); r32(d->efc_pos, d->nefc); r32(d->efc_D, d->nefc); r32(d->efc_R, d->nefc); r32(d->efc_aref, d->nefc);
  }
  rne_bias(m, d);
  if (rm & RM_RNE) { r32(d->cacc, 6L * nb); r32(d->cfrc, 6L * nb); r32(d->qfrc_bias, nv); }
  actuation(m, d);
  if (rm & RM_FRC) { r32(d->qfrc_passive, nv); r32(d->qfrc_actuator, nv); }
  acceleration(m, d);
  if (rm & RM_ACC) { r32(d->qfrc_smooth, nv); r32(d->qacc_smooth, nv); }
  solve_constraints(m, d);
  if (rm & RM_SOLVER) { r32(d->qacc, nv); r32(d->qacc_warmstart, nv); r32(d->qfrc_constraint, nv); r32(d->efc_force, d->nefc); }
  sensors(m, d);
}

/* mj_inverse [MJ-KNOWLEDGE engine_inverse.c]: inverse dynamics at the current (qpos, qvel, qacc), continuous-time form
 * (mjENBL_INVDISCRETE off).  Position and velocity stages as in mj_forward; the constraint force follows in closed form
 * from jar = J qacc - aref (mj_invConstraint -> mj_constraintUpdate: one-sided quadratic rows, force = -D jar where
 * jar < 0); qfrc_inverse = M qacc + qfrc_bias - qfrc_passive - qfrc_constraint.
 * Reference call sites: mujoco_template/setpoints.py:29-31, examples/humanoid/controllers/lqr.py:57-70. */
void mjo_inverse(const mjoModel* m, mjoData* d) {
  int nv = m->nv;
  kinematics(m, d);
  com_pos(m, d);
  tendon_transmission(m, d);
  crb_factor(m, d);
  collision(m, d);
  make_constraint(m, d);
  com_vel(m, d);
  passive(m, d);
  reference_constraint(m, d);
  rne_bias(m, d);
  mul_J(m, d, d->efc_jar, d->qacc);
  for (int i = 0; i < d->nefc; i++) d->efc_jar[i] -= d->efc_aref[i];
  constraint_update(m, d, d->efc_jar, NULL);
  for (int k = 0; k < nv; k++) {
    double s = 0;
    for (int i = 0; i < d->nefc; i++) s += d->efc_J[(size_t)i * nv + k] * d->efc_force[i];
    d->qfrc_constraint[k] = s;
  }
  mul_M(m, d, d->s_Ma, d->qacc);
  for (int k = 0; k < nv; k++) d->qfrc_inverse[k] = d->s_Ma[k] + d->qfrc_bias[k] - d->qfrc_passive[k] - d->qfrc_constraint[k];
}

/* A16 mj_integratePos / mj_differentiatePos */
void mjo_integrate_pos(const mjoModel* m, double* qpos, const double* qvel, double dt) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += dt * qvel[da + k];
      quat_integrate(qpos + qa + 3, qvel + da + 3, dt);
    } else qpos[qa] += dt * qvel[da];
  }
}
void mjo_differentiate_pos(const mjoModel* m, double* qvel, double dt, const double* qpos1, const double* qpos2) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) qvel[da + k] = (qpos2[qa + k] - qpos1[qa + k]) / dt;
      quat_sub(qvel + da + 3, qpos1 + qa + 3, qpos2 + qa + 3);
      for (int k = 0; k < 3; k++) qvel[da + 3 + k] /= dt;
    } else qvel[da] = (qpos2[qa] - qpos1[qa]) / dt;
  }
}

/* A13 bad-state guard (mj_checkPos/Vel/Acc): NaN or |x| > 1e10 -> reset to qpos0 */
static int bad(const double* x, int n) { for (int i = 0; i < n; i++) if (!(fabs(x[i]) <= MAXVAL)) return 1; return 0; }

/* A11 integrators */
static void euler(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_INTEG);
  int nv = m->nv;
  double h = m->timestep;
  double* qacc = d->s_tmp;
  if (m->has_damping) {
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    memcpy(d->qH, d->qM, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) d->qH[i * nv + i] += h * m->dof_damping[i];
    chol_factor(d->qH, nv);
    chol_solve(d->qH, nv, qacc);
  } else memcpy(qacc, d->qacc, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  mjo_integrate_pos(m, d->qpos, d->qvel, h);
  d->time += h;
}

static void rk4(const mjoModel* m, mjoData* d) {
  int nq = m->nq, nv = m->nv;
  MJO_PHASE(MJO_PH_INTEG);
  double h = m->timestep, time0 = d->time;
  static const double A[9] = {0.5, 0, 0, 0, 0.5, 0, 0, 0, 1}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  double* X0q = dalloc(nq); double* X0v = dalloc(nv);
  double* Fv = dalloc(4 * nv); double* Fa = dalloc(4 * nv); double* dv = dalloc(nv); double* da = dalloc(nv);
  memcpy(X0q, d->qpos, sizeof(double) * nq); memcpy(X0v, d->qvel, sizeof(double) * nv);
  memcpy(Fv, d->qvel, sizeof(double) * nv); memcpy(Fa, d->qacc, sizeof(double) * nv);
  for (int i = 1; i < 4; i++) {
    memset(dv, 0, sizeof(double) * nv); memset(da, 0, sizeof(double) * nv);
    double C = 0;
    for (int j = 0; j < 3; j++) {
      double a = A[(i - 1) * 3 + j];
      C += a;
      if (a == 0) continue;
      for (int k = 0; k < nv; k++) { dv[k] += a * Fv[j * nv + k]; da[k] += a * Fa[j * nv + k]; }
    }
    memcpy(d->qpos, X0q, sizeof(double) * nq);
    mjo_integrate_pos(m, d->qpos, dv, h);
    for (int k = 0; k < nv; k++) d->qvel[k] = X0v[k] + h * da[k];
    d->time = time0 + C * h;
    mjo_forward(m, d);
    MJO_PHASE(MJO_PH_INTEG);
    memcpy(Fv + i * nv, d->qvel, sizeof(double) * nv); memcpy(Fa + i * nv, d->qacc, sizeof(double) * nv);
  }
  memset(dv, 0, sizeof(double) * nv); memset(da, 0, sizeof(double) * nv);
  for (int j = 0; j < 4; j++) for (int k = 0; k < nv; k++) { dv[k] += B[j] * Fv[j * nv + k]; da[k] += B[j] * Fa[j * nv + k]; }
  memcpy(d->qpos, X0q, sizeof(double) * nq);
  for (int k = 0; k < nv; k++) d->qvel[k] = X0v[k] + h * da[k];
  mjo_integrate_pos(m, d->qpos, dv, h);
  d->time = time0 + h;
  free(X0q); free(X0v); free(Fv); free(Fa); free(dv); free(da);
}

void mjo_step(const mjoModel* m, mjoData* d) {
  MJO_PHASE(MJO_PH_OTHER);
  if (bad(d->qpos, m->nq)) { d->warn_badqpos++; double t = d->time; mjo_reset(m, d); (void)t; }
  if (bad(d->qvel, m->nv)) { d->warn_badqvel++; mjo_reset(m, d); }
  mjo_forward(m, d);
  if (bad(d->qacc, m->nv)) { d->warn_badqacc++; mjo_reset(m, d); mjo_forward(m, d); }
  if (m->integrator == INT_RK4) rk4(m, d); else euler(m, d);
  if (m->round_mask & RM_STATE) { r32(d->qpos, m->nq); r32(d->qvel, m->nv); }      /* precision study: fp32 state storage */
}

/* ------------------------------------------------------------------------- */
/* counter-based uniform random ctrl (Philox4x32-10), identical on CPU and GPU */
/* ------------------------------------------------------------------------- */
static void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
  for (int r = 0; r < 10; r++) {
    unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void mjo_random_ctrl(const mjoModel* m, double* ctrl, unsigned seed, unsigned env, unsigned step, double scale) {
  for (int a = 0; a < m->nu; a++) {
    unsigned r[4];
    philox4x32(env, step, (unsigned)a, 0u, seed, 0x5EEDu, r);
    double u = (double)(r[0] >> 8) * (1.0 / 16777216.0);   /* [0,1), exact in fp32 */
    double lo = -1, hi = 1;
    if (m->actuator_ctrllimited[a]) { lo = m->actuator_ctrlrange[2 * a]; hi = m->actuator_ctrlrange[2 * a + 1]; }
    double mid = 0.5 * (lo + hi), half = 0.5 * (hi - lo);
    ctrl[a] = mid + half * scale * (2 * u - 1);
  }
}
void mjo_rollout_random(const mjoModel* m, mjoData* d, int nstep, unsigned seed, unsigned env, unsigned step0, double scale) {
  for (int s = 0; s < nstep; s++) {
    mjo_random_ctrl(m, d->ctrl, seed, env, step0 + (unsigned)s, scale);
    mjo_step(m, d);
  }
}

/* ------------------------------------------------------------------------- */
/* A14 mjd_transitionFD (centred or forward), tangent-space state x = (dq, qvel) */
/* ------------------------------------------------------------------------- */
void mjo_transition_fd(const mjoModel* m, mjoData* d, double eps, int centered, double* A, double* B) {
  int nq = m->nq, nv = m->nv, nu = m->nu, nx = 2 * nv;
  double* q0 = dalloc(nq); double* v0 = dalloc(nv); double* u0 = dalloc(nu); double* w0 = dalloc(nv);
  double* yq0 = dalloc(nq); double* yv0 = dalloc(nv);
  double* yqp = dalloc(nq); double* yvp = dalloc(nv); double* yqm = dalloc(nq); double* yvm = dalloc(nv);
  double* dq = dalloc(nv);
  double t0 = d->time;
  memcpy(q0, d->qpos, sizeof(double) * nq); memcpy(v0, d->qvel, sizeof(double) * nv);
  memcpy(u0, d->ctrl, sizeof(double) * nu); memcpy(w0, d->qacc_warmstart, sizeof(double) * nv);
#define RESTORE() do { memcpy(d->qpos, q0, sizeof(double) * nq); memcpy(d->qvel, v0, sizeof(double) * nv); \
    memcpy(d->ctrl, u0, sizeof(double) * nu); memcpy(d->qacc_warmstart, w0, sizeof(double) * nv); d->time = t0; } while (0)
  mjo_step(m, d);
  memcpy(yq0, d->qpos, sizeof(double) * nq); memcpy(yv0, d->qvel, sizeof(double) * nv);
  int ncol = 2 * nv + nu;
  for (int col = 0; col < ncol; col++) {
    int have_p = 1, have_m = centered;
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      if (sgn == -1 && !centered) break;
      RESTORE();
      if (col < nv) { memset(dq, 0, sizeof(double) * nv); dq[col] = 1; mjo_integrate_pos(m, d->qpos, dq, sgn * eps); }
      else if (col < 2 * nv) d->qvel[col - nv] += sgn * eps;
      else {
        int a = col - 2 * nv;
        double v = u0[a] + sgn * eps;
        if (m->actuator_ctrllimited[a] && (v < m->actuator_ctrlrange[2 * a] || v > m->actuator_ctrlrange[2 * a + 1])) {
          if (sgn == 1) have_p = 0; else have_m = 0;       /* nudge would leave ctrlrange: one-sided difference */
          continue;
        }
        d->ctrl[a] = v;
      }
      mjo_step(m, d);
      if (sgn == 1) { memcpy(yqp, d->qpos, sizeof(double) * nq); memcpy(yvp, d->qvel, sizeof(double) * nv); }
      else { memcpy(yqm, d->qpos, sizeof(double) * nq); memcpy(yvm, d->qvel, sizeof(double) * nv); }
    }
    const double *aq, *av, *bq, *bv; double den;
    if (have_p && have_m) { aq = yqm; av = yvm; bq = yqp; bv = yvp; den = 2 * eps; }
    else if (have_p) { aq = yq0; av = yv0; bq = yqp; bv = yvp; den = eps; }
    else if (have_m) { aq = yqm; av = yvm; bq = yq0; bv = yv0; den = eps; }
    else { aq = bq = yq0; av = bv = yv0; den = 1; }
    mjo_differentiate_pos(m, dq, den, aq, bq);
    for (int r = 0; r < nv; r++) {
      double dvv = (bv[r] - av[r]) / den;
      if (col < 2 * nv) { if (A) { A[r * nx + col] = dq[r]; A[(nv + r) * nx + col] = dvv; } }
      else if (B) { B[r * nu + (col - 2 * nv)] = dq[r]; B[(nv + r) * nu + (col - 2 * nv)] = dvv; }
    }
  }
  RESTORE();
#undef RESTORE
  mjo_forward(m, d);
  memcpy(d->qacc_warmstart, w0, sizeof(double) * nv);
  free(q0); free(v0); free(u0); free(w0); free(yq0); free(yv0); free(yqp); free(yvp); free(yqm); free(yvm); free(dq);
}

/* A15 Jacobians: kind 0 site, 1 body origin, 2 body com, 3 subtree com */
void mjo_jac(const mjoModel* m, const mjoData* d, int kind, int id, double* jacp, double* jacr) {
  int nv = m->nv;
  if (kind == 0) jac_point(m, d, m->site_bodyid[id], d->site_xpos + 3 * id, jacp, jacr);
  else if (kind == 1) jac_point(m, d, id, d->xpos + 3 * id, jacp, jacr);
  else if (kind == 2) jac_point(m, d, id, d->xipos + 3 * id, jacp, jacr);
  else {
    double* tmp = dalloc(3 * nv);
    if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
    if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
    for (int b = id; b < m->nbody; b++) {
      int p = b, inside = 0;
      while (p > 0) { if (p == id) { inside = 1; break; } p = m->body_parentid[p]; }
      if (id == 0) inside = 1;
      if (!inside || m->body_mass[b] <= 0) continue;
      jac_point(m, d, b, d->xipos + 3 * b, tmp, NULL);
      if (jacp) for (int k = 0; k < 3 * nv; k++) jacp[k] += tmp[k] * m->body_mass[b];
    }
    if (jacp && m->body_subtreemass[id] > MINVAL) for (int k = 0; k < 3 * nv; k++) jacp[k] /= m->body_subtreemass[id];
    free(tmp);
  }
}

/* ------------------------------------------------------------------------- */
/* CPU-baseline helper: nenv independent random-ctrl rollouts, one env per OpenMP task. */
/* qpos_out (may be NULL): [nenv, nq] final positions.  Returns env-steps executed.      */
/* ------------------------------------------------------------------------- */
#ifdef _OPENMP
#include <omp.h>
#endif
long mjo_rollout_batch(const mjoModel* m, int nenv, int nstep, unsigned seed, unsigned env0, double scale, int nthreads,
                       const double* qpos_init, const double* qvel_init, double* qpos_out, double* qvel_out) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int e = 0; e < nenv; e++) {
    mjoData* d = mjo_data_create(m);
    if (qpos_init) memcpy(d->qpos, qpos_init + (size_t)e * m->nq, sizeof(double) * m->nq);
    if (qvel_init) memcpy(d->qvel, qvel_init + (size_t)e * m->nv, sizeof(double) * m->nv);
    mjo_rollout_random(m, d, nstep, seed, env0 + (unsigned)e, 0u, scale);
    if (qpos_out) memcpy(qpos_out + (size_t)e * m->nq, d->qpos, sizeof(double) * m->nq);
    if (qvel_out) memcpy(qvel_out + (size_t)e * m->nv, d->qvel, sizeof(double) * m->nv);
    mjo_data_free(d);
  }
  return (long)nenv * nstep;
}
