/*
 * mjbatch.h — C ABI of the MI355X-native batched MuJoCo step/rollout engine.
 *
 * This is the drop-in boundary for the reference's per-step physics path.  The
 * reference (pure Python) reaches its physics through the pybind11 surface of the
 * third-party `mujoco` package (stub: reference mujoco_template/mujoco.pyi:1-75);
 * each entry point below names the reference call site it replaces.  Handles are
 * opaque, arguments are plain pointers and sizes, every function returns 0 on
 * success or a negative status and leaves a message for mjb_last_error().
 * All functions on one mjbData are NOT re-entrant (reference semantics are
 * single-threaded); launches go to the stream set with mjb_set_stream().
 *
 * There is no CPU fallback: without a HIP device mjb_data_create() fails.
 */
#ifndef MJBATCH_H
#define MJBATCH_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mjbModel mjbModel;
typedef struct mjbData mjbData;
typedef struct mjbObsSpec mjbObsSpec;

#define MJB_OK 0
#define MJB_ERR_ARG (-1)     /* bad argument / unknown name        -> ConfigError      */
#define MJB_ERR_MODEL (-2)   /* model table rejected                -> ValueError       */
#define MJB_ERR_DEVICE (-3)  /* HIP error / no device               -> TemplateError    */
#define MJB_ERR_LOOKUP (-4)  /* index out of range                  -> NameLookupError  */

#define MJB_F32 0
#define MJB_F64 1

/* ctrl source of a fused rollout */
#define MJB_CTRL_KEEP 0    /* use data.ctrl as is (host controller wrote it)  */
#define MJB_CTRL_ZERO 1    /* ZeroController, reference controllers.py:12-25 */
#define MJB_CTRL_RANDOM 2  /* uniform random ctrl, Philox(seed; env, step, actuator) */
#define MJB_CTRL_FEEDBACK 3 /* ctrl = clip(u0 - K [q (-) q0; qvel - v0]): the LQR law of the reference's examples
                               (examples/humanoid/controllers/lqr.py:147-170), gains set by mjb_set_feedback */

const char* mjb_last_error(void);
int mjb_device_count(void);

/* ---- model: replaces MjModel.from_xml_* (reference model.py:22-37).  mjb_model_load_xml* compile MJCF inside the library;
 * mjb_model_create takes an already compiled model as a table of named arrays (dtype 0 = float64, 1 = int32, 2 = bytes) whose
 * names follow mjModel (what mjb_model_load_xml* and mjb_model_load build internally). ---- */
int mjb_model_create(int nfield, const char* const* names, const void* const* ptrs, const int* dtypes,
                     const long* counts, mjbModel** out);
/* MjModel.from_xml_path / from_xml_string (reference model.py:22-27) in the library itself: the MJCF-subset compiler
 * (csrc/mjb_mjcf.cpp, host C++) -> the same table -> mjb_model_create.  Anything outside the supported subset is rejected with
 * MJB_ERR_MODEL and a message naming it (the host front raises ValueError, as mujoco's compiler does).  base_dir resolves
 * <include file=...>. */
int mjb_model_load_xml(const char* path, mjbModel** out);
int mjb_model_load_xml_string(const char* xml_text, const char* base_dir, mjbModel** out);
void mjb_model_free(mjbModel* m);
/* model.opt.disableactuator bit mask, reference model.py:88-93 */
int mjb_model_set_disableactuator(mjbModel* m, int mask);
/* solver knobs (mjOption.iterations / tolerance) */
int mjb_model_set_solver(mjbModel* m, int iterations, double tolerance);

/* object names: mj_name2id / mj_id2name (reference observations.py:80-84, jacobians.py:39-75, logging.py:88-126, model.py:64).
 * objtype = MuJoCo's mjtObj code (1 body, 2 xbody, 3 joint, 5 geom, 6 site, 18 tendon, 19 actuator, 20 sensor, 24 key);
 * name2id returns -1 when the name is unknown (the host front raises NameLookupError), id2name NULL for unnamed / out of range.
 * The names cross mjb_model_create as table fields "names_<objtype>" of dtype 2 (bytes: NUL-terminated names in id order). */
int mjb_model_name2id(const mjbModel* m, int objtype, const char* name);
const char* mjb_model_id2name(const mjbModel* m, int objtype, int id);
/* any field of the compiled model by its mjModel name ("nq", "nv", "actuator_ctrlrange", "jnt_type", ...): host pointer valid for the
 * model's lifetime, element count and dtype (0 float64, 1 int32, 2 bytes) — the model.nq / model.actuator_* / model.jnt_* attribute
 * reads of the reference (mujoco.pyi:7-49, compat.py:64-121) for a host that is not Python */
int mjb_model_field(const mjbModel* m, const char* name, const void** ptr, long* count, int* dtype);
/* enumeration of the same fields: 0 and the field's name / pointer / count / dtype for 0 <= index < n, otherwise n (the field count) */
int mjb_model_field_at(const mjbModel* m, int index, const char** name, const void** ptr, long* count, int* dtype);
/* MjModel.from_binary_path / mj_saveModel (reference model.py:28-31, :49): the compiled table in a flat binary file ("MJBM0001").
 * A host without the Python MJCF compiler loads a model compiled once elsewhere. */
int mjb_model_save(const mjbModel* m, const char* path);
int mjb_model_load(const char* path, mjbModel** out);

/* mj_integratePos(m, qpos, qvel, dt) / mj_differentiatePos(m, qvel, dt, qpos1, qpos2) (reference linearization.py:12,67,77,
 * examples/humanoid/controllers/lqr.py:153, examples/drone2/main.py:404-406): in place on CALLER-OWNED HOST vectors, float64,
 * batched: qpos [batch, nq], qvel [batch, nv] row-major.  qvel_out = (qpos2 (-) qpos1) / dt in the tangent space. */
int mjb_integrate_pos(const mjbModel* m, int batch, double* qpos, const double* qvel, double dt);
int mjb_differentiate_pos(const mjbModel* m, int batch, double* qvel_out, double dt, const double* qpos1, const double* qpos2);

/* ---- data: replaces MjData(model) (reference model.py:16-19) for `batch` independent replicas.
 * dtype: MJB_F32 (product path) or MJB_F64.  lanes: lanes per environment (8, 16, 64; 0 = auto).
 * nconmax / nefcmax: per-environment contact / constraint-row caps held in LDS (0 = default).
 * env0: global index of this shard's first environment (keeps random ctrl shard-invariant). ---- */
int mjb_data_create(mjbModel* m, int batch, int dtype, int lanes, int nconmax, int nefcmax, int device, int env0, mjbData** out);
void mjb_data_free(mjbData* d);
int mjb_set_stream(mjbData* d, void* hip_stream);
int mjb_sync(mjbData* d);
/* Engine failures cross the ABI (reference convention: mujoco raises FatalError / mujoco_template raises TemplateError,
 * exceptions.py:4-21 - nothing is ever silently wrong).  Sticky flag word of the batch, bit 0 contacts dropped, 1 constraint rows
 * dropped (per-environment LDS caps exceeded: truncated physics, COUNTED per environment by mjb_get_counters), 2 bad-state auto-reset
 * (mj_checkPos/Vel/Acc), 3 a ticket-mode launch timed out waiting for a state hand-over.  mjb_engine_flags waits for the stream and
 * reports the word (the kernels keep it in pinned host memory: no copy).  Bit 3 is an ERROR: from the first synchronising call
 * after it was raised (mjb_sync, mjb_get_array, mjb_get_counters, mjb_sync_to_host, mjb_step_host*) and at the entry of every
 * further launch (mjb_step, mjb_rollout, mjb_forward ...) the library returns MJB_ERR_DEVICE until mjb_reset clears the flags; the
 * environments concerned were NOT advanced (their arrays hold the state the failed launch started from). */
int mjb_engine_flags(mjbData* d, int* flags_out);
int mjb_data_info(mjbData* d, int* batch, int* dtype, int* lanes, int* nconmax, int* nefcmax, int* lds_bytes_per_env);

/* device pointer of a [batch, n] state array: qpos qvel ctrl qacc qacc_warmstart (dtype of the data),
 * time (float64 [batch]), xpos xquat xipos site_xpos geom_xpos subtree_com sensordata qfrc_inverse actuator_moment, counters (int32 [batch, 8]) */
int mjb_array_ptr(mjbData* d, const char* name, void** dev_ptr, long* per_env, int* dtype);
/* host <-> device copies with conversion to/from float64 (state snapshot/restore, reference state_utils.py:9-31) */
int mjb_get_array(mjbData* d, const char* name, double* host_out);
int mjb_set_array(mjbData* d, const char* name, const double* host_in);
int mjb_get_counters(mjbData* d, int* host_out /* [batch, 8] */);

/* ---- host mirror ("host_view"): the zero-copy numpy views data.qpos / qvel / ctrl / qacc / qacc_warmstart / time of the reference
 * (mujoco.pyi:51-75; observations with copy=False alias them, tests/test_mujoco_template.py:241-252) as ONE pinned float64 block
 * per data object.  mjb_host_view returns the address of one field's [batch, n] slice (valid for the data's lifetime);
 * mjb_sync_to_host refreshes the whole block with one pack kernel + ONE device-to-host copy + one stream sync;
 * mjb_sync_to_device uploads the fields named in field_mask (bit 0 qpos, 1 qvel, 2 ctrl, 3 qacc, 4 qacc_warmstart, 5 time) after the
 * host edited them in place (data.ctrl[:] = ..., what every reference controller does);
 * mjb_step_host = sync_to_device(field_mask) + nstep x mj_step (nstep = 0: mj_forward) + sync_to_host in one call: the body of the
 * reference's host-driven loop (env.py:186-190, runtime.py:631-663) costs one library call per step. ---- */
/* name "engine_flags": ONE double behind the state, refreshed by the same copy: sticky bits 0 contacts dropped, 1 constraint rows dropped
 * (per-environment LDS caps exceeded), 2 bad-state auto-reset, 3 hand-over timed out (see mjb_engine_flags) — the whole-batch OR of what
 * mjb_get_counters details per environment */
int mjb_host_view(mjbData* d, const char* name, double** host_ptr, long* per_env);
int mjb_sync_to_host(mjbData* d);
int mjb_sync_to_device(mjbData* d, int field_mask);
int mjb_step_host(mjbData* d, int nstep, int field_mask);
/* Edit detection inside the library: the reference's controllers write data.ctrl / qpos / qvel IN PLACE through the numpy views
 * (control.py:26-32, examples/drone2/main.py:393-397), so "what did the host change since the block was last refreshed" is a comparison
 * of the pinned block with a library-owned shadow copy.  mjb_mirror_edited_mask: bit k set = field k differs from the shadow;
 * mjb_mirror_commit: the shadow takes the block's current content for the fields in field_mask (after a refresh or an upload);
 * mjb_step_host_auto = [compare != 0: edited mask, else 0] + mjb_step_host(nstep, mask) + commit(all): ONE call per reference-style
 * Env.step; *mask_out (may be NULL) receives the mask that was uploaded. */
int mjb_mirror_edited_mask(mjbData* d, int* mask_out);
int mjb_mirror_commit(mjbData* d, int field_mask);
int mjb_step_host_auto(mjbData* d, int nstep, int compare, int* mask_out);

/* ---- per-model specialisation of the fp32 step kernel (no reference counterpart: the reference's MjModel is interpreted by
 * one pre-built C library; here the structural sizes of the compiled model and the LDS layout offsets can be folded into the
 * kernel).  mjb_model_spec_source / mjb_spec_source write the translation unit (returns its length; call with buf = NULL to
 * size it); compile it for gfx950 with `hipcc --genco -I <csrc>` (mujoco_template_amd/_capi.py does, cached in-tree) and hand
 * the code object to mjb_spec_load: every later launch on this data object uses it (same arguments, same results).
 * The generic kernel stays the default and the fallback. ---- */
long mjb_model_spec_source(mjbModel* m, int dtype, int lanes, int nconmax, int nefcmax, char* buf, long cap);
long mjb_spec_source(mjbData* d, char* buf, long cap);
int mjb_spec_load(mjbData* d, const void* code_object, long nbytes);
int mjb_spec_unload(mjbData* d);
/* The same for the TWO-WAVE step kernel that stepping launches use on small batches (one environment per 128-thread workgroup, the
 * independent phases of a step side by side on its two wavefronts; fp32, one wave per environment, nv <= 32, Euler): results bitwise
 * those of the one-wave kernel.  mjb_step_schedule()[5] tells whether a launch used it; MJB_TWO_WAVE=0 / 1 forces it off / on. */
long mjb_model_step2_spec_source(mjbModel* m, int lanes, int nconmax, int nefcmax, char* buf, long cap);
long mjb_step2_spec_source(mjbData* d, char* buf, long cap);
int mjb_step2_spec_load(mjbData* d, const void* code_object, long nbytes);
int mjb_step2_spec_unload(mjbData* d);
/* The same for the float64 finite-difference kernel behind mjb_transition_fd (k_fd<double, TS, G> of this data object: float64 layout,
 * model baked in as float64 constants): source -> `hipcc --genco` -> mjb_fd_spec_load; results bitwise those of the generic kernel. */
long mjb_model_fd_spec_source(mjbModel* m, int dtype, int lanes, int nconmax, int nefcmax, char* buf, long cap);   /* without a data object (build step) */
long mjb_fd_spec_source(mjbData* d, char* buf, long cap);
int mjb_fd_spec_load(mjbData* d, const void* code_object, long nbytes);
int mjb_fd_spec_unload(mjbData* d);

/* mj_resetData / mj_resetDataKeyframe (reference model.py:59-71); key < 0 = qpos0 */
int mjb_reset(mjbData* d, int key);
/* mj_forward (reference model.py:53-54): fills qacc and the kinematic outputs */
int mjb_forward(mjbData* d);
/* mj_inverse (reference setpoints.py:29-31 steady_ctrl0, examples/humanoid/controllers/lqr.py:57-70): inverse dynamics at the
 * current (qpos, qvel, qacc) of every environment -> array "qfrc_inverse" [batch, nv]; the same pass fills
 * "actuator_moment" [batch, nu, nv] (dense form of data.actuator_moment, which setpoints.py:40-47 densifies).  State is not advanced. */
int mjb_inverse(mjbData* d);
/* nstep x mj_step (reference model.py:56-57, env.py:190), ctrl taken from data.ctrl */
int mjb_step(mjbData* d, int nstep);
/* fused rollout = the body of runtime.iterate_passive (reference runtime.py:631-663) for a device-side
 * controller: nstep x [ctrl <- mode, mj_step]; when spec != NULL the flat observation of every
 * `obs_every`-th step is written to obs_out_dev[(nstep/obs_every), batch, dim] (dtype of the data). */
int mjb_rollout(mjbData* d, int nstep, int ctrl_mode, unsigned seed, unsigned step0, double ctrl_scale,
                const mjbObsSpec* spec, void* obs_out_dev, int obs_every);

/* gains of MJB_CTRL_FEEDBACK, host float64: K [nu, 2nv] row-major, u0 [nu], q0 [nq], v0 [nv] (NULL = zeros); shared by all environments */
int mjb_set_feedback(mjbData* d, const double* K, const double* u0, const double* q0, const double* v0);

/* the same law as a STANDALONE batched kernel for the host-driven loop (one K for many environments: K dx is a GEMM [batch, 2nv] x
 * [2nv, nu], on MFMA in fp32): writes data.ctrl on the device from the current qpos / qvel, then the host calls mjb_step / mjb_step_host.
 * Optional ctrl noise of the reference law (lqr.py:160-165): ctrl += std[a] * table[(step + env * env_stride) mod nsteps][a] before the
 * clip (also inside MJB_CTRL_FEEDBACK rollouts, step = the rollout's step counter); mjb_set_feedback_noise(d, NULL, NULL, 0, 0) switches it off.  Host float64 inputs: std [nu], table [nsteps, nu]. */
int mjb_set_feedback_noise(mjbData* d, const double* noise_std, const double* noise_table, int nsteps, int env_stride);
int mjb_feedback_ctrl(mjbData* d, int step);

/* ---- observations: ObservationExtractor.__call__ with as_dict=False (reference observations.py:98-174) ---- */
int mjb_obs_spec_create(mjbData* d, int flags, int nsite, const int* site_ids, int nbody, const int* body_ids,
                        int ngeom, const int* geom_ids, int nsubtree, const int* subtree_ids, mjbObsSpec** out);
void mjb_obs_spec_free(mjbObsSpec* s);
int mjb_obs_dim(const mjbObsSpec* s);
int mjb_obs_gather(mjbData* d, const mjbObsSpec* s, void* out_dev /* [batch, dim], dtype of the data */);

/* the ONE collective of the path: all-gather of the flat observation block over RCCL / xGMI for a host that owns
 * an ncclComm_t (one process per GPU): send_dev [count_per_rank] -> recv_dev [nranks * count_per_rank], dtype MJB_F32 / MJB_F64, on
 * hip_stream.  The Python front uses torch.distributed's all_gather_into_tensor instead (distributed.all_gather_obs); this entry
 * point is the same ncclAllGather for a non-Python host.  The library does not link RCCL: it resolves ncclAllGather from the RCCL
 * already in the process, else from librccl.so.1. */
int mjb_allgather_obs(void* nccl_comm, const void* send_dev, void* recv_dev, long count_per_rank, int dtype, void* hip_stream);

/* ---- mjd_transitionFD (reference linearization.py:16-35): float64 on device.
 * A_host [batch, 2nv, 2nv], B_host [batch, 2nv, nu], row-major ---- */
int mjb_transition_fd(mjbData* d, double eps, int centered, double* A_host, double* B_host);
/* the same without the final host copy: pointers to the library's PINNED result blocks (same layouts), valid until the next
 * mjb_transition_fd* call on this data object — at humanoid batch 512 the two blocks are 16.5 MB */
int mjb_transition_fd_pinned(mjbData* d, double eps, int centered, const double** A_pinned, const double** B_pinned);

/* ---- mj_jacSite / mj_jacBody / mj_jacBodyCom / mj_jacSubtreeCom (reference jacobians.py:44-79).
 * kinds[i]: 0 site, 1 body, 2 bodycom, 3 subtreecom.  jacp/jacr host [batch, nreq, 3, nv] float64 (jacr may be NULL) ---- */
int mjb_jac(mjbData* d, int nreq, const int* kinds, const int* ids, double* jacp_host, double* jacr_host);

/* per-phase dumps of the last mjb_forward for parity tests: name in
 * qM qfrc_bias qfrc_passive qfrc_actuator qacc_smooth qfrc_constraint efc_J efc_aref efc_D efc_pos efc_force con cdof cinert cvel (float64 out)
 * and efc_type (int32 out).  Call mjb_debug_forward() first. */
/* diagnostic build (-DMJB_PROFILE) only: per-phase shader-cycle sums since the last call, host_out[24]; zeros otherwise */
int mjb_profile_get(mjbData* d, unsigned long long* host_out);
/* How the last stepping launch (mjb_step / mjb_rollout / mjb_step_host) mapped work to workgroups: out6 = { steps of the launch,
 * environment blocks, resident workgroup slots of the step kernel on this device (0 = unknown), chunk_steps (0 = static map: one
 * workgroup per block for all steps; > 0 = the resident workgroups drew (block, chunk) tickets), fair_bit (0 = hardware age order), 1 if the launch used two wavefronts per environment (small batches) }.
 * The engine picks the map itself (more blocks than slots -> tickets); MJB_CHUNK_STEPS / MJB_FAIR_BIT override it for experiments. */
int mjb_step_schedule(mjbData* d, int* out6);
/* diagnostic kernel (-DMJB_TIMELINE) only: per environment [start, end] of its wave in the last launch (100 MHz clock), HW_ID, XCC_ID */
int mjb_profile_env_get(mjbData* d, unsigned long long* host_out /* [batch, 4] */);
int mjb_debug_forward(mjbData* d);
int mjb_debug_get(mjbData* d, const char* name, void* host_out, long capacity_elems);

#ifdef __cplusplus
}
#endif
#endif
