/*
 * mbd_hip.h — C ABI of libmbd_hip.so: the MI355X-native reverse-diffusion sampling loop of
 * LeCAR-Lab/model-based-diffusion (mbd/planners/mbd_planner.py:84-148,179-180).
 *
 * Every entry point below replaces one piece of the reference's (pure Python/JAX) plugin surface; the
 * reference file:line it stands in for is cited on each declaration.  The reference has no FFI of its own —
 * the binding a maintainer would add is the ctypes stub shown in INTEGRATION.md (and shipped as
 * model-based-diffusion_amd/mbd_hip/_capi.py).
 *
 * Conventions
 *   - plain C types only; every function returns int (0 = MBD_OK, <0 = mbd_status); no exceptions/aborts
 *     cross the ABI; mbd_last_error() returns a thread-local message for the last failure.
 *   - pointers named d_* are DEVICE pointers (HBM, caller-owned, e.g. torch.Tensor.data_ptr()); all other
 *     pointers are HOST pointers. The library never keeps a caller pointer after return.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Functions taking a stream are
 *     asynchronous w.r.t. the host unless documented otherwise.
 *   - all floating point data is IEEE float32, all PRNG words uint32 (reference: everything is f32,
 *     mbd_planner.py:13-14 keeps x64 disabled).
 *   - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef MBD_HIP_H
#define MBD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------ */
/* status codes                                                                                      */
/* ------------------------------------------------------------------------------------------------ */
typedef enum mbd_status {
  MBD_OK = 0,
  MBD_ERR_INVALID = -1,     /* bad argument (shape, NULL, range)                                     */
  MBD_ERR_UNSUPPORTED = -2, /* env name / model feature outside the hot-path scope (-> ValueError)   */
  MBD_ERR_HIP = -3,         /* a HIP runtime call failed; message carries hipGetErrorString           */
  MBD_ERR_NO_DEVICE = -4,   /* no gfx950 device visible: the product path NEVER falls back to a CPU  */
  MBD_ERR_STATE = -5        /* call sequence error (e.g. plan used after destroy)                    */
} mbd_status;

/* ------------------------------------------------------------------------------------------------ */
/* compiled rigid-body model ("sys")                                                                 */
/* ------------------------------------------------------------------------------------------------ */
/* What brax.io.mjcf.load returns as a `System` pytree (call sites mbd/envs/humanoidrun.py:15,
 * hopper.py:14, humanoidtrack.py:16) flattened to one POD struct.  It is produced on the host by
 * mbd_hip/mjcf.py (or loaded from a committed .json) and handed to mbd_env_create_model().
 * Frames: every link carries a centre-of-mass frame whose ORIENTATION equals the link frame
 * (offset `com` only), so inverse inertia is a full symmetric body-frame tensor. */
#define MBD_MAX_LINKS 16
#define MBD_MAX_Q 40
#define MBD_MAX_ACT 24
#define MBD_MAX_COL 16
#define MBD_MAX_TRACK 8

enum mbd_reward_kind {
  MBD_REW_HUMANOIDRUN = 0,   /* mbd/envs/humanoidrun.py:46-51                                        */
  MBD_REW_HOPPER = 1,        /* mbd/envs/hopper.py:57-65 and walker2d.py:57-62: x - clip(|z - p0|,-1,1)*p1,
                                reward_params = (p0, p1) = (1.0, 0.5) hopper / (1.1, 0.5) walker2d     */
  MBD_REW_HALFCHEETAH = 2,   /* brax.envs.half_cheetah (absent from the reference tree)              */
  MBD_REW_HUMANOIDTRACK = 3, /* mbd/envs/humanoidtrack.py:87-96 (computed from the INCOMING state)   */
  MBD_REW_HUMANOIDSTANDUP = 4, /* mbd/envs/humanoidstandup.py:50-56                                   */
  MBD_REW_ANT = 6,           /* brax.envs.ant (absent): forward_reward + healthy_reward - ctrl_cost:
                                p0*(x1-x0)/dt + (p5 != 0 || p2 <= z <= p3 ? p4 : 0) - p1*|a|^2, reward_params =
                                (1, 0.5, 0.2, 1.0, 1.0, 1) — p5 = terminate_when_unhealthy (stock: on, the
                                healthy term is then a constant); recollection, unpinned                 */
  MBD_REW_CARTPOLE = 5       /* mbd/envs/cartpole.py:45: cos(q[1]) - |qd[0]| (hinge of link 1, slide of link 0) */
};

enum mbd_model_flags {
  MBD_FLAG_RESET_QUAT_RAW = 1, /* reset(): leave the noise-perturbed root quaternion un-normalised (humanoidrun.py:24-26
                                  perturbs all 7 root coordinates; whether kinematics.forward renormalises is unverified).
                                  Default 0: normalised.                                                          */
  MBD_FLAG_PLANAR = 2,         /* the model moves in the x-z plane (every hinge about the world y axis, slides and offsets
                                  in the plane, no free joint: hopper, walker2d, halfcheetah, cartpole) and is simulated by
                                  the planar restatement of the same six stages — in-plane coordinates only — instead
                                  of the general 3-D arithmetic, whose float round-off leaks 1e-5..1e-3 out of the
                                  plane over a rollout.  Set by mbd_hip/mjcf.py when the model qualifies (planar=False
                                  keeps the 3-D path); a specification of its own for these models. */
  /* ---- SPECIFICATION SWITCHES: the places where this engine had to guess at CODE level what Brax's positional
   * pipeline does.  Each bit selects the named form, in the checker and in the kernels alike (bit-exact
   * against each other either way), so that a golden vector of the real reference flips a flag instead of forcing a rewrite
   * (tools/compare_golden.py --search tries every combination).  The DEFAULT word is MBD_DEFAULT_SPEC (below): what
   * mbd_hip.mjcf.load gives a model, what the built-in models carry and what the shipped library's tuned kernels compile in
   * (mbd_tuned_spec()); a model with another word runs the general `spec` kernel instantiations. */
  MBD_FLAG_CONTACT_AVG = 4,        /* several ACTIVE contacts on one link: the link's position correction (stage 4) and —
                                      unless CONTACT6_GAUSS_SEIDEL — its velocity change (stage 6) are the AVERAGE over
                                      them (sum * 1/n, n >= 2) instead of the sum; single contacts are untouched.
                                      SET BY DEFAULT since round 6: contacts solved independently (Jacobi, below) and
                                      SUMMED blow up when a link's contacts sit close to its centre of mass — four
                                      spheres 1 cm around it turn -0.5 m/s into -9.5 m/s in ONE substep
                                      (tests/test_oracle_invariants.py) — so an engine that vmaps its contacts has to
                                      average them (Brax v1's colliders divided by the contact count); bit clear: the
                                      sum of rounds 1-5                                                                    */
  MBD_FLAG_CONTACT6_GAUSS_SEIDEL = 8, /* stage (6), collisions.resolve_velocity.  DEFAULT (bit clear, since round 5): every
                                      contact of a link computes its impulse from the SAME velocities (those stage (5)
                                      left) and the changes are added in collider order (Jacobi) — the only form Brax's code
                                      structure allows: it vmaps its contacts and segment-sums their changes per link.
                                      Bit set: one contact after the other, each seeing what the link's previous contacts
                                      left (Gauss-Seidel per link — the default of rounds 1-4, kept as the alternative)    */
  MBD_FLAG_FRICTION_VEL_BOUND = 16, /* stage (6) dynamic friction: |dv_t| = min(mu lambda_n / h, |v_t|) (Mueller et al.
                                      2020, eq. 30, literally: the bound is a velocity) instead of
                                      min(mu lambda_n / h * w_t, |v_t|) (the bound is an impulse)                         */
  MBD_FLAG_RESTITUTION_MIN = 32,   /* stage (6): the literal min(-e vn_prev, 0) of eq. 34 (Brax's sign convention unknown;
                                      with this engine's +z normal it makes elasticity a no-op) instead of max(.., 0)     */
  MBD_FLAG_EULER_EXTRINSIC = 64,   /* joints with 2 or 3 hinge dofs: q composes as rotations about the FIXED joint-frame
                                      axes (R = Rz(q2) Ry(q1) Rx(q0): gimbal axes Xc, Zp x Xc, Zp) instead of the moving
                                      ones (R = Rx(q0) Ry(q1) Rz(q2): Xp, Zc x Xp, Zc) — forward kinematics at reset, the
                                      angles / axes of stage (1) torques and stage (3) limits, observations               */
  MBD_FLAG_GYROSCOPIC = 128        /* stage (2): angular acceleration includes -I^-1 (w x I w) (Mueller et al., eq. for
                                      the velocity update); identically zero for isotropic tensors and planar models,
                                      which ignore the bit                                                                */
};
#define MBD_SPEC_FLAGS (4 | 8 | 16 | 32 | 64 | 128)
#define MBD_DEFAULT_SPEC 4 /* Jacobi per link + average over a link's active contacts */

typedef struct mbd_model {
  /* sizes */
  int32_t n_links, n_q, n_qd, n_act, n_col, n_track, n_frames, reward_kind;
  int32_t iso_inertia; /* 1: every inv_inertia is s*identity (spring_inertia_scale = 1 models)    */
  int32_t flags;       /* mbd_model_flags: named switches for the places where this engine had to GUESS what
                          Brax does — flipping one is a recompile of the model, not of the code */
  int32_t reserved_i[2];
  /* solver scalars */
  float dt;            /* physics substep (opt.timestep); control dt = dt * n_frames                */
  float vel_fac;       /* exp(vel_damping * dt)                                                     */
  float ang_fac;       /* exp(ang_damping * dt)                                                     */
  float joint_scale_pos, joint_scale_ang, collide_scale;
  float friction, elasticity;
  float gravity[3];
  float reset_noise;   /* U(-reset_noise, reset_noise) on q and qd at reset; 0 = deterministic     */
  float reward_params[8];
  /* per link */
  int32_t parent[MBD_MAX_LINKS];  /* -1 = world                                                    */
  int32_t n_rot[MBD_MAX_LINKS];   /* hinge dofs of the link's joint (0..3); -1 = free joint        */
  int32_t n_slide[MBD_MAX_LINKS]; /* slide dofs (0..3), listed before the hinges in q              */
  int32_t q_idx[MBD_MAX_LINKS], qd_idx[MBD_MAX_LINKS];
  float inv_mass[MBD_MAX_LINKS];
  float inv_inertia[MBD_MAX_LINKS][6]; /* xx yy zz xy xz yz, link frame, about the COM             */
  float com[MBD_MAX_LINKS][3];         /* COM in the link frame                                    */
  float ap_pos[MBD_MAX_LINKS][3], ap_rot[MBD_MAX_LINKS][4]; /* joint frame on the parent, in the
                                                               parent's COM frame (world if -1)   */
  float ac_pos[MBD_MAX_LINKS][3], ac_rot[MBD_MAX_LINKS][4]; /* joint frame on the child, in the
                                                               child's COM frame                  */
  float ang_damp[MBD_MAX_LINKS], vel_damp[MBD_MAX_LINKS];   /* constraint_{ang,vel}_damping        */
  float rot_lo[MBD_MAX_LINKS][3], rot_hi[MBD_MAX_LINKS][3]; /* limits on the joint-frame Euler
                                                               angles (x, y', z''), radians       */
  float rot_stiff[MBD_MAX_LINKS][3], rot_damp[MBD_MAX_LINKS][3];
  float rot_sign[MBD_MAX_LINKS][3]; /* q_k = rot_sign_k * euler_k (handedness of the MJCF axes)    */
  float slide_axis[MBD_MAX_LINKS][3][3]; /* slide axes in the joint frame                          */
  float slide_lo[MBD_MAX_LINKS][3], slide_hi[MBD_MAX_LINKS][3]; /* slide limits (metres)            */
  float slide_damp[MBD_MAX_LINKS][3];    /* MJCF joint damping of the slide dofs                    */
  /* actuators (brax.actuator.to_tau: clip to ctrlrange, * gear, scatter to the dof)               */
  int32_t act_link[MBD_MAX_ACT];
  int32_t act_slot[MBD_MAX_ACT]; /* 0..2 hinge k, 3..5 slide k                                     */
  float act_gear[MBD_MAX_ACT], act_lo[MBD_MAX_ACT], act_hi[MBD_MAX_ACT];
  /* sphere colliders vs the z = 0 plane                                                            */
  int32_t col_link[MBD_MAX_COL];
  float col_pos[MBD_MAX_COL][3]; /* sphere centre in the link's COM frame                          */
  float col_radius[MBD_MAX_COL];
  /* forward kinematics (reset only; kinematics.forward)                                            */
  float link_pos[MBD_MAX_LINKS][3], link_rot[MBD_MAX_LINKS][4]; /* link frame in the parent frame  */
  float joint_pos[MBD_MAX_LINKS][3];     /* joint anchor in the link frame                         */
  float rot_axis[MBD_MAX_LINKS][3][3];   /* hinge axes in the link frame                           */
  float slide_axis_body[MBD_MAX_LINKS][3][3];
  float init_q[MBD_MAX_Q];
  /* demo tracking (humanoidtrack.py:26-28)                                                         */
  int32_t track_link[MBD_MAX_TRACK];
} mbd_model_t;

/* Dynamic state of one environment = per link 13 floats: COM position p[3], orientation quaternion
 * r[4] = (w,x,y,z), linear velocity v[3], angular velocity w[3] (world frame).  This is brax's
 * positional State.x_i / State.xd_i — the only quantities integrated from step to step. Layout of a
 * state buffer: float[n_links][13].  car2d's state is float[3] = (x, y, theta) (car2d.py:35-40). */
#define MBD_LINK_STATE 13

/* ------------------------------------------------------------------------------------------------ */
/* library                                                                                           */
/* ------------------------------------------------------------------------------------------------ */
const char* mbd_last_error(void);
int mbd_version(void);
/* The word of specification switches (mbd_model_flags, MBD_SPEC_FLAGS) this BUILD's tuned kernels compile in: models whose
 * switches equal it run them, any other word runs the general instantiations that read the switches per launch (same results,
 * a third to a half of the speed).  MBD_DEFAULT_SPEC in the shipped library; -DMBD_TUNED_SPEC=<word> rebuilds it (no
 * counterpart in the reference, whose physics has ONE specification — Brax's, which no vector pins yet). */
int mbd_tuned_spec(void);
/* number of usable gfx950 devices (0 on a box without a GPU — every compute entry then returns
 * MBD_ERR_NO_DEVICE; there is deliberately no CPU fallback in this library). */
int mbd_device_count(int* count);

/* ------------------------------------------------------------------------------------------------ */
/* JAX PRNG (threefry2x32) — replaces jax.random.{PRNGKey,split} at mbd_planner.py:40,79,103,150    */
/* host-side, pure integer arithmetic, no device needed.                                             */
/* ------------------------------------------------------------------------------------------------ */
enum mbd_prng_impl {
  MBD_PRNG_LEGACY = 0,       /* jax_threefry_partitionable = False (JAX < 0.5.0 default)            */
  MBD_PRNG_PARTITIONABLE = 1 /* jax_threefry_partitionable = True  (JAX >= 0.5.0 default)           */
};
int mbd_prng_key(uint64_t seed, uint32_t key_out[2]);
int mbd_prng_split(const uint32_t key[2], int num, int impl, uint32_t* keys_out /* [num][2] */);

/* ------------------------------------------------------------------------------------------------ */
/* environments — replaces mbd.envs.get_env (mbd/envs/__init__.py:13-33) and the env methods the    */
/* planner uses: reset (mbd_planner.py:75,80), step (:74), action_size/observation_size (:71-72),   */
/* eval_xref_logpd (:118), rew_xref (:121)                                                           */
/* ------------------------------------------------------------------------------------------------ */
typedef struct mbd_env mbd_env;

/* get_env(env_name) (mbd/envs/__init__.py:13-33): a string in, an env out.  The compiled models ("sys" of
 * humanoidrun.py:15, hopper.py:14, humanoidtrack.py:16, ...) and the demo trajectories (car2d.py:66,
 * humanoidtrack.py:33-43) are constant data inside the library, so a C caller needs no Python and no MJCF
 * compiler.  Names: car2d, hopper, halfcheetah, humanoidrun, humanoidtrack, walker2d, humanoidstandup, cartpole,
 * ant.  "pushT" (generalized backend) and unknown names return MBD_ERR_UNSUPPORTED — the Python shim raises
 * ValueError for both, like :33. */
int mbd_env_create(const char* env_name, int device, mbd_env** out);
/* the names mbd_env_create accepts: index 0, 1, ... until NULL (host only) */
const char* mbd_env_name(int index);
/* the embedded compiled model of a built-in rigid-body env (host only, no device needed) */
int mbd_builtin_model(const char* env_name, mbd_model_t* model_out);

/* The two constructors below take caller-supplied data instead (custom MJCF models compiled by
 * mbd_hip/mjcf.py, other demo paths).
 * car2d needs no model; `env_name` must be "car2d" (mbd/envs/car2d.py:43-71).  xref = demo path
 * [50][2] float32 (car2d_xref.npy cast to f32) or NULL when demos are not used. */
int mbd_env_create_car2d(int device, const float* xref, mbd_env** out);
/* rigid-body envs: humanoidrun / humanoidtrack / hopper / halfcheetah — the caller passes the
 * compiled model. xref = [n_track][50][3] float32 demo body positions or NULL. rew_xref as
 * humanoidtrack.py:44. */
int mbd_env_create_model(const char* env_name, int device, const mbd_model_t* model,
                         const float* xref, float rew_xref, mbd_env** out);
int mbd_env_destroy(mbd_env* env);

/* action_size / observation_size / state record size (floats) / H limit of the demo (0 = none)   */
int mbd_env_info(const mbd_env* env, int* action_size, int* observation_size, int* state_size,
                 int* n_links, int* n_frames, float* dt);
/* reset(rng) -> state (humanoidrun.py:19-32, hopper.py:20-34, humanoidtrack.py:48-61, car2d.py:73-75).
 * Host computation (forward kinematics once per run). state_out: float[state_size] HOST. */
int mbd_env_reset(const mbd_env* env, const uint32_t key[2], int prng_impl, float* state_out);
/* PipelineEnv.pipeline_init(q, qd) -> pipeline_state (the call every wrapper's reset ends in: humanoidrun.py:29,
 * hopper.py:30, walker2d.py:29, humanoidstandup.py:29, cartpole.py:29, humanoidtrack.py:54): forward kinematics of the
 * generalized coordinates — a state to plan FROM that is not a reset (receding-horizon use).  q: float[n_q] (free root:
 * position, quaternion w-first — normalised like reset's unless MBD_FLAG_RESET_QUAT_RAW), qd: float[n_qd]; n_q / n_qd must
 * be the model's (mbd_env_get_model).  car2d: q = (x, y, theta), n_q = 3, qd ignored (may be NULL, n_qd = 0).  Host
 * arithmetic, no device needed; all pointers HOST. */
int mbd_env_pipeline_init(const mbd_env* env, const float* q, int n_q, const float* qd, int n_qd, float* state_out);
/* step(state, action) -> (state', reward, obs) for ONE environment (rendering / verification path,
 * mbd_planner.py:163; utils.py:23-33).  Runs the same HIP rollout kernel with B=1,H=1; synchronous.
 * All pointers HOST. reward_out / obs_out (float[observation_size], see mbd_env_observe) may be NULL. */
int mbd_env_step(mbd_env* env, const float* state_in, const float* action, float* state_out,
                 float* reward_out, float* obs_out);
/* rew_xref (car2d.py:71, humanoidtrack.py:44) */
int mbd_env_rew_xref(const mbd_env* env, float* out);
/* env.sys (mbd_planner.py:174; humanoidrun.py:15): a copy of the env's compiled model */
int mbd_env_get_model(const mbd_env* env, mbd_model_t* model_out);
/* env.xref (mbd_planner.py:167; car2d.py:66 [50][2], humanoidtrack.py:36-43 [n_track][50][3]) copied to the HOST
 * buffer xref_out[capacity]; count_out = number of floats (0: the env has no demonstration). xref_out may be NULL. */
int mbd_env_xref(const mbd_env* env, float* xref_out, int capacity, int* count_out);
/* jax.vmap(env.eval_xref_logpd)(qs) (mbd_planner.py:118; humanoidtrack.py:98-106, car2d.py:95-102):
 *   d_xpos      : [B][H][K][3] tracked link positions after every control step (car2d: [B][H][3] = q) — the
 *                 d_xpos output of mbd_env_rollout
 *   d_logpd_out : [B]
 * H must be 50 (the demos have 50 rows). asynchronous on `stream`. */
int mbd_env_xref_logpd(const mbd_env* env, const float* d_xpos, int B, int H, float* d_logpd_out, void* stream);
/* _get_obs (humanoidrun.py:43-44, hopper.py:49-55, car2d.py:86; brax ant / half_cheetah) of ONE state:
 * kinematics.inverse on the host (the planner never reads observations, mbd_planner.py:71 — API parity only).
 * state: float[state_size] HOST; obs_out: float[observation_size] HOST. */
int mbd_env_observe(const mbd_env* env, const float* state, float* obs_out);
/* the same from a bare model, plus the generalized coordinates: q_out[n_q], qd_out[n_qd], obs_out — any may be
 * NULL.  Pure host arithmetic, no device needed. */
int mbd_model_observe(const mbd_model_t* model, const float* state, float* q_out, float* qd_out, float* obs_out);

/* pipeline_init from a bare model (the inverse of mbd_model_observe's q_out / qd_out): q[model->n_q], qd[model->n_qd] ->
 * state_out[13 * n_links].  Pure host arithmetic, no device needed. */
int mbd_model_forward(const mbd_model_t* model, const float* q, const float* qd, float* state_out);

/* Batched rollout = jax.vmap(rollout_us, in_axes=(None,0)) (mbd_planner.py:109, utils.py:14-20).
 *   d_state0 : [state_size]        one initial state shared by all B candidates
 *   d_us     : [B][H][Nu]          action sequences
 *   d_rewss  : [B][H]              reward after every control step
 *   d_xpos   : [B][H][K][3] or NULL  world positions of the K tracked links after every step
 *                                  (car2d: [B][H][3] = q).  Only what eval_xref_logpd consumes of the
 *                                  reference's `pipline_states` output.
 *   d_state_final : [B][state_size] or NULL
 * asynchronous on `stream`. */
int mbd_env_rollout(mbd_env* env, const float* d_state0, const float* d_us, int B, int H,
                    float* d_rewss, float* d_xpos, float* d_state_final, void* stream);

/* ------------------------------------------------------------------------------------------------ */
/* planner fast path — replaces reverse_once / reverse / run_diffusion (mbd_planner.py:97-151,179)  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct mbd_plan mbd_plan;

typedef struct mbd_plan_config {
  int32_t Nsample;     /* GLOBAL number of candidates N (Args.Nsample, mbd_planner.py:29)           */
  int32_t Hsample;     /* horizon H (Args.Hsample :30)                                              */
  int32_t Ndiffuse;    /* diffusion steps (Args.Ndiffuse :31)                                       */
  float temp_sample;   /* (:32)                                                                     */
  float beta0, betaT;  /* (:33-34)                                                                  */
  int32_t enable_demo; /* (:35)                                                                     */
  int32_t prng_impl;   /* mbd_prng_impl                                                             */
  int32_t shard_begin; /* this process owns candidates [shard_begin, shard_begin + shard_count)     */
  int32_t shard_count; /* = Nsample on one GPU                                                      */
  int32_t literal_score; /* 1: evaluate score/Yim1/Ybar_im1 literally (:130-133); 0: use identity   */
  int32_t update_method; /* 0 = MBD (mbd_planner.py); path-integral baselines of
                            mbd/planners/path_integral.py:33-52 on the same rollout kernel:
                            1 = mppi (softmax_update), 2 = cma-es, 3 = cem. For these Ndiffuse plays
                            Nrefine (:28), sigma is a carried scalar starting at 1.0 (:131), the
                            standardisation has no zero-std guard (:123) and demos are not used.      */
  int32_t shares_device; /* 1: other plans run on this GPU at the same time (plans of a sweep on separate streams):
                            the plan then generates each step's normals in front of its rollout instead of one step
                            ahead in workgroups / on a stream the other plans need (results identical either way)  */
  int32_t reserved[3];
} mbd_plan_config;

int mbd_plan_create(mbd_env* env, const mbd_plan_config* cfg, mbd_plan** out);
int mbd_plan_destroy(mbd_plan* plan);
/* noise schedule (mbd_planner.py:84-87): copies alphas, alphas_bar, sigmas ([Ndiffuse] each, HOST) */
int mbd_plan_schedule(const mbd_plan* plan, float* alphas, float* alphas_bar, float* sigmas);
/* state_init = reset(rng_reset) (mbd_planner.py:79-80): uploads a HOST state to the plan */
int mbd_plan_set_state0(mbd_plan* plan, const float* state0);

/* ---- one reverse-diffusion step, split at the (only) exchange point so that N can be sharded ---- */
/* phase 1 (mbd_planner.py:103-110): eps -> Y0s = clip(eps*sigma_i + Ybar_i) for the local shard,
 * rollout, rews = mean_H(rewss) [+ demo log-densities].  (MBD plans on rigid-body envs keep eps and form the
 * candidate values where they are consumed — the rollout's action fetch, the weighted mean, mbd_plan_peek — with the
 * same two roundings; d_Ybar_i must stay unchanged until phase 2 of the step has run.)  d_Ybar_i [H][Nu] (device, read), key_sample
 * is Y0s_rng of (:103).  Writes d_rews_local [shard_count] and, with demos, d_logpd_local
 * [shard_count] (else may be NULL).  async on stream. */
int mbd_plan_sample_rollout(mbd_plan* plan, int i, const uint32_t key_sample[2],
                            const float* d_Ybar_i, float* d_rews_local, float* d_logpd_local,
                            void* stream);
/* Optional hint BEFORE phase 1: declares key_next, the Y0s_rng of the diffusion step AFTER the one the next
 * mbd_plan_sample_rollout runs.  Its normals (jax.random.normal(Y0s_rng, ...), mbd_planner.py:104 — they depend on that
 * step's key only) are then generated beside that rollout: in spare workgroups of the rollout launch itself when it
 * leaves CUs idle, on the plan's second stream otherwise; the declared step finds them ready when its key_sample
 * equals key_next and generates its own otherwise.  Results are bit-identical with and without the hint.  Plans that
 * materialise Y0s (car2d, path-integral updates) ignore it.  `stream` is unused (kept for ABI stability). */
int mbd_plan_prefetch_noise(mbd_plan* plan, const uint32_t key_next[2], void* stream);
/* phase 2 (mbd_planner.py:111-135): from ALL N rewards (after the all-gather) standardise, demo
 * blend, softmax, weighted mean over all N candidates (noise regenerated from the counter-based PRNG,
 * so the result is bit-identical on every rank and for every shard layout), score update.
 * Writes d_Ybar_im1 [H][Nu] and d_rew_mean [1] (= rews.mean(), :135). async on stream.
 * (The phases of a plan may be issued on different streams: the library orders each call behind the plan's previous
 * one with an event when the stream changes — the normals prepared by one step's launch are read by the next.) */
int mbd_plan_score_update(mbd_plan* plan, int i, const uint32_t key_sample[2], const float* d_Ybar_i,
                          const float* d_rews_all, const float* d_logpd_all, float* d_Ybar_im1,
                          float* d_rew_mean, void* stream);
/* path-integral plans: the carried sampling sigma (path_integral.py:113,131). set before the first
 * step (mbd_plan_run does it itself); synchronous. */
int mbd_plan_set_sigma(mbd_plan* plan, float sigma);
int mbd_plan_get_sigma(mbd_plan* plan, float* sigma_out);
/* single-GPU convenience = reverse_once (mbd_planner.py:97-135): phase 1 + phase 2 on `stream`;
 * key_inout is advanced exactly as `rng, Y0s_rng = split(rng)` (:103). async; d_Ybar updated in
 * place; d_rew_mean [1]. */
int mbd_plan_reverse_once(mbd_plan* plan, int i, uint32_t key_inout[2], float* d_Ybar,
                          float* d_rew_mean, void* stream);
/* Host blocking of the phase calls: a plan whose next step's normals are generated on its second stream (plans that
 * fill the chip) keeps the host at most one step ahead of the device — mbd_plan_sample_rollout may then sleep up to 5 ms
 * (mbd_plan_run / mbd_sweep_run: 20 ms) for the previous rollout to START; when the stream is slower than that (a shared
 * or time-sliced GPU, earlier work on the caller's stream, a profiler) the call falls back to ordering its two streams
 * with an event and returns — a slow stream is never an error. */
/* whole reverse loop = reverse() (mbd_planner.py:138-148) + final evaluation (:179-180).
 * key = rng_exp of (:150).  mu_0ts_out HOST [Ndiffuse-1][H][Nu] (the array saved at :156),
 * rew_means_out HOST [Ndiffuse-1] (the tqdm postfix values, :147) — either may be NULL.
 * Synchronous; one device->host copy at the end instead of the reference's per-step sync. */
int mbd_plan_run(mbd_plan* plan, const uint32_t key[2], float* mu_0ts_out, float* rew_means_out,
                 float* rew_final_out, double* loop_seconds_out);
/* rew_final = rollout_us(state_init, Y).mean() for one plan Y [H][Nu] HOST (mbd_planner.py:179-180) */
int mbd_plan_eval(mbd_plan* plan, const float* Y, float* rew_final_out);

/* what the last step worked on, copied to HOST buffers (inspection / parity tests; synchronises the device): the
 * candidates Y0s [Nsample][H][Nu] (plans that keep normals instead form them here, from the normals, sigma_i and the
 * Ybar_i of the last step — between phase 1 and phase 2 the caller's d_Ybar_i must still be unchanged), the shard's
 * rewss [shard_count][H], the softmax weights [Nsample].  Any pointer may be NULL. */
int mbd_plan_peek(mbd_plan* plan, float* Y0s_out, float* rewss_out, float* weights_out);
/* timing of the dominant (rollout) kernel measured with hipEvents on the launch stream:
 * average milliseconds per launch since the last reset; count = launches. reset != 0 clears. */
int mbd_plan_kernel_time(mbd_plan* plan, float* avg_ms_out, int* count_out, int reset);
int mbd_plan_enable_timing(mbd_plan* plan, int enable);

/* ------------------------------------------------------------------------------------------------ */
/* sweeps — replaces the loops of mbd/scripts/run_mbd.py (:17-39 eight seeds, :42-64 eight           */
/* temperatures): P independent MBD plans of ONE env with the same (Nsample, Hsample, Ndiffuse,      */
/* beta0, betaT, enable_demo) advanced in lockstep, ONE rollout launch over the P * Nsample          */
/* candidates and ONE score + weighted-mean launch per diffusion step.  Seeds (keys, start states)   */
/* and temperatures may differ per plan.  Every plan's result is bit-identical to mbd_plan_run on    */
/* that plan alone.                                                                                  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct mbd_sweep mbd_sweep;
#define MBD_SWEEP_MAX_PLANS 32
/* cfg: as for mbd_plan_create (unsharded; Nsample * 4 bytes <= 48 KB: larger plans fill the chip on their own — run
 * them as plans).  update_method 0: MBD plans; 1 / 2 / 3: the path-integral baselines mppi / cma-es / cem
 * (run_mbd.py:22-26,46-50 over path_integral.py:111-127; Ndiffuse plays Nrefine, no demos): one sampling launch, one rollout
 * launch and the update rule's kernels over all plans per refinement step.  temps: [n_plans] temp_sample per plan, or
 * NULL: cfg->temp_sample. */
int mbd_sweep_create(mbd_env* env, const mbd_plan_config* cfg, int n_plans, const float* temps, mbd_sweep** out);
int mbd_sweep_destroy(mbd_sweep* sweep);
/* state_init of plan k (HOST, state_size floats) */
int mbd_sweep_set_state0(mbd_sweep* sweep, int k, const float* state0);
/* the P reverse loops + final evaluations (mbd_planner.py:138-148,179-180).  keys: [n_plans][2] = rng_exp of each
 * plan (:150).  HOST outputs, any may be NULL: mu_0ts_out [n_plans][Ndiffuse-1][H][Nu], rew_means_out
 * [n_plans][Ndiffuse-1], rew_final_out [n_plans]; loop_seconds_out: wall time of the lockstep loop.  Synchronous. */
int mbd_sweep_run(mbd_sweep* sweep, const uint32_t* keys, float* mu_0ts_out, float* rew_means_out,
                  float* rew_final_out, double* loop_seconds_out);
/* path-integral sweeps: the carried sampling sigma of every plan after the last run (path_integral.py:113,131); HOST [n_plans] */
int mbd_sweep_get_sigmas(mbd_sweep* sweep, float* sigmas_out);
/* average milliseconds of the sweep's rollout launches since the last reset (hipEvents on the launch stream) */
int mbd_sweep_kernel_time(mbd_sweep* sweep, int enable, float* avg_ms_out, int* count_out);

/* ------------------------------------------------------------------------------------------------ */
/* in-library exchange — the ONE collective of a sharded diffusion step (the all-gather of the       */
/* per-candidate mean rewards between mbd_plan_sample_rollout and mbd_plan_score_update,             */
/* mbd_planner.py:109-111 with N sharded) WITHOUT a collective library: every rank owns a receive     */
/* window in device memory, peers write their slice into it directly (xGMI peer stores through       */
/* hipIpc mappings) and raise a flag per rank and epoch; a consumer kernel on the caller's stream    */
/* waits for the world's flags and hands over the gathered [rows][N] values.  Messages are <= 64 KB: */
/* the step pays two short kernels instead of a collective launch.  Results are the all-gather's.    */
/* ------------------------------------------------------------------------------------------------ */
typedef struct mbd_exchange mbd_exchange;
#define MBD_IPC_HANDLE_BYTES 64
#define MBD_EXCHANGE_MAX_RANKS 16
/* rank `rank` of `world` (<= MBD_EXCHANGE_MAX_RANKS) on `device`: rows x shard floats per rank and step.  The window is
 * FINE-GRAINED device memory (peers' stores and the owner's loads meet at system scope); a runtime without such a pool
 * gets MBD_ERR_UNSUPPORTED — keep the collective library's all-gather then (a coarse-grained window can hand the owner
 * a flag beside stale rewards). */
int mbd_exchange_create(int device, int rank, int world, int rows, int shard, mbd_exchange** out);
/* 1: the window is fine-grained memory (always, unless the test lever MBD_EXCHANGE_COARSE_OK allowed otherwise) */
int mbd_exchange_fine_grained(const mbd_exchange* x, int* out);
int mbd_exchange_destroy(mbd_exchange* x);
/* this rank's window as an IPC handle (MBD_IPC_HANDLE_BYTES bytes, HOST).  The caller passes the handles around by
 * any host channel (torch.distributed.all_gather_object, MPI, a file) and hands all of them to _connect. */
int mbd_exchange_local_handle(mbd_exchange* x, void* handle_out);
/* handles: [world][MBD_IPC_HANDLE_BYTES] HOST, rank-major (the own entry is not opened) */
int mbd_exchange_connect(mbd_exchange* x, const void* handles);
/* d_local [rows][shard] (device) -> every rank's window; *d_all_out: device pointer to [rows][world * shard], rank-major
 * = candidate order, valid until the next call.  Asynchronous on `stream`; every rank must call it once per step. */
int mbd_exchange_all_gather(mbd_exchange* x, const float* d_local, const float** d_all_out, void* stream);
/* MBD_OK, or MBD_ERR_STATE when a wait ran into its time limit (a peer that never arrived); synchronises the device */
int mbd_exchange_status(mbd_exchange* x);

#ifdef __cplusplus
}
#endif
#endif /* MBD_HIP_H */
