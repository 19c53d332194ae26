/*
 * mjo.h — CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C, float64, single-environment restatement of the arithmetic the
 * reference delegates to the third-party `mujoco` library at these call
 * sites:
 *   mj_step            reference mujoco_template/model.py:56-57
 *   mj_forward         reference mujoco_template/model.py:53-54
 *   mj_resetData(+Keyframe)   model.py:59-71
 *   mjd_transitionFD   reference mujoco_template/linearization.py:16-35
 *   mj_jacSite/Body/BodyCom/SubtreeCom   reference mujoco_template/jacobians.py:44-79
 *   mj_integratePos / mj_differentiatePos   linearization.py:10-13,67,77
 *
 * PARITY UNPINNED: the `mujoco` wheel (pyproject.toml:11, ">=3.1", no lock) is
 * absent from /root/reference and from this image, and the reference's tests
 * hold no numeric physics vectors.  The algorithm below is a
 * restatement of MuJoCo 3.x's published computation model from memory
 * [MJ-KNOWLEDGE]; it is anchored by analytic known-answer tests
 * (tests/test_oracle_anchors.py), not by golden vectors of the real library.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#ifndef MJO_H
#define MJO_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mjoModel mjoModel;
typedef struct mjoData mjoData;

/* Model is built from a table of named arrays (dtype 0 = float64, 1 = int32). */
mjoModel* mjo_model_create(int nfield, const char* const* names, const void* const* ptrs,
                           const int* dtypes, const long* counts);
void mjo_model_free(mjoModel* m);
const char* mjo_last_error(void);
void mjo_set_disableactuator(mjoModel* m, int mask);
void mjo_set_limits(mjoModel* m, int nconmax, int nefcmax);   /* 0 = unlimited (default) */
void mjo_set_solver(mjoModel* m, int iterations, double tolerance);
/* precision study only: bit k set = the outputs of phase k are rounded to fp32 (fp32-storage model); 0 = the oracle proper */
void mjo_set_round_mask(mjoModel* m, int mask);

mjoData* mjo_data_create(const mjoModel* m);
void mjo_data_free(mjoData* d);
/* pointer to a named per-env array (qpos, qvel, ctrl, qacc, qacc_warmstart, xpos, ...); count returned */
double* mjo_data_array(mjoData* d, const char* name, long* count);
int* mjo_data_iarray(mjoData* d, const char* name, long* count);
/* contacts of the last forward pass: out[i*15] = dist, pos[3], frame[9], geom1, geom2; returns ncon */
long mjo_get_contacts(const mjoData* d, double* out, long maxcon);
double mjo_get_time(const mjoData* d);
void mjo_set_time(mjoData* d, double t);

void mjo_reset(const mjoModel* m, mjoData* d);
int mjo_reset_keyframe(const mjoModel* m, mjoData* d, int key);
void mjo_forward(const mjoModel* m, mjoData* d);
/* mj_inverse: fills qfrc_inverse from (qpos, qvel, qacc); reference mujoco_template/setpoints.py:29-31 */
void mjo_inverse(const mjoModel* m, mjoData* d);
void mjo_step(const mjoModel* m, mjoData* d);

/* random-ctrl rollout used by the CPU baseline / parity tests:
 * ctrl[a] = mid + half*scale*(2u-1), u = philox(seed, env, step, a) */
void mjo_random_ctrl(const mjoModel* m, double* ctrl, unsigned seed, unsigned env, unsigned step, double scale);
void mjo_rollout_random(const mjoModel* m, mjoData* d, int nstep, unsigned seed, unsigned env,
                        unsigned step0, double scale);

/* CPU-baseline helper: nenv independent random-ctrl rollouts across OpenMP threads; returns env-steps executed */
long mjo_rollout_batch(const mjoModel* m, int nenv, int nstep, unsigned seed, unsigned env0, double scale, int nthreads,
                       const double* qpos_init, const double* qvel_init, double* qpos_out, double* qvel_out);

void mjo_transition_fd(const mjoModel* m, mjoData* d, double eps, int centered, double* A, double* B);
/* kind: 0 site, 1 body (frame origin), 2 body com, 3 subtree com.  jacp/jacr may be NULL. */
void mjo_jac(const mjoModel* m, const mjoData* d, int kind, int id, double* jacp, double* jacr);
void mjo_integrate_pos(const mjoModel* m, double* qpos, const double* qvel, double dt);
void mjo_differentiate_pos(const mjoModel* m, double* qvel, double dt, const double* qpos1, const double* qpos2);

#ifdef __cplusplus
}
#endif
#endif
