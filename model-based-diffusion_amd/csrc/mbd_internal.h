// mbd_internal.h — what the translation units of libmbd_hip.so's host side share (NOT part of the boundary: the C ABI is
// include/mbd_hip.h): the handle structs, error reporting, the test levers, and the launch functions one unit defines for
// the others.  Units: mbd_env.hip (library, levers, envs, the rollout launch), mbd_plan.hip (plans: one reverse-diffusion
// step and the loops over it), mbd_sweep.hip (sweeps: several plans per launch), mbd_exchange.hip (the in-library exchange).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mbd_hip.h"
#include "mbd_kernels.h"
#include "mbd_step_kernels.h"

using namespace mbd;

// error reporting (mbd_env.hip): sets the thread-local message of mbd_last_error() and returns `code`
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(MBD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)


inline int device_count_quiet() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---- host PRNG (jax.random.split) --------------------------------------------------------------------
inline void host_split(const uint32_t key[2], int num, int impl, uint32_t* keys) {
  if (impl == MBD_PRNG_PARTITIONABLE) {
    for (int j = 0; j < num; ++j) threefry2x32(key[0], key[1], 0u, (uint32_t)j, keys[2 * j], keys[2 * j + 1]);
    return;
  }
  for (int e = 0; e < 2 * num; ++e) keys[e] = random_bits32(key[0], key[1], 0, (uint64_t)e, (uint64_t)(2 * num));
}

enum EnvKind { ENV_CAR2D = 0, ENV_MODEL = 1 };
constexpr int kLdsN = 36 * 1024;  // candidates whose logp0 fits the score kernel's LDS (144 KB of the CU's 160)


struct mbd_env {
  int kind = ENV_MODEL;
  int device = 0;
  std::string name;
  mbd_model_t model;
  mbd_model_t* d_model = nullptr;
  float* d_xref = nullptr;
  bool has_xref = false;
  float rew_xref = 0.0f;
  int lps = 16, max_children = 0, max_col = 0, max_rot = 0;
  int n_cus = 256;  // compute units of the device (hipDeviceProp_t::multiProcessorCount; four SIMDs each)
  bool diag_inertia = true;  // every body-frame inverse-inertia tensor is exactly diagonal
  bool axisym = true;        // ... with two equal entries: axisymmetric about a link axis (AXI instantiations)
  bool slides = false;
  bool slide_limits = false;  // any slide dof with a finite range
  int max_slide = 0;          // largest slide-dof count of a joint
  bool slides_world_only = true;  // every joint with a slide dof hangs off the world
  bool has_weld = false;          // some joint has no hinge dof
  bool planar = false;            // MBD_FLAG_PLANAR: the planar restatement (mbd_planar.h)
  bool any_stiff = false;         // some hinge has a joint spring
  // DPP layout (kernels.h "lane exchange without the LDS"): lane <-> link tables when the tree fits the shifts
  int dpp_family = -1;  // index into kDppFamilies, -1: shuffles
  signed char lane_tab[32];
  signed char* d_lane_tab = nullptr;
  LaneRec3* d_lane_rec = nullptr;  // [3][16]: per-lane constants of the 3-D kernels (lane = link; the DPP layout; the
                                   // DPP layout with helper lanes)
  bool helpers = false;  // one link with 3..5 colliders and two idle lanes to lend them to (HELP instantiations)
  bool spec = false;     // the model carries specification switches (MBD_SPEC_FLAGS): the general SPEC instantiations, 16 lanes
  unsigned long long* dbg_clock = nullptr;  // per env, caller-owned device buffer (mbd_debug_set_clock_buffer; probes only)
  // scratch for the single-env step path
  float *d_s_in = nullptr, *d_act = nullptr, *d_s_out = nullptr, *d_rew = nullptr;
  mbd_env() = default;
  mbd_env(const mbd_env&) = delete;
  mbd_env& operator=(const mbd_env&) = delete;
  ~mbd_env() {  // owns its device buffers: every exit of the create functions, early or not, releases them
    (void)hipSetDevice(device);
    (void)hipFree(d_model); (void)hipFree(d_xref); (void)hipFree(d_lane_tab); (void)hipFree(d_lane_rec);
    (void)hipFree(d_s_in); (void)hipFree(d_s_out); (void)hipFree(d_act); (void)hipFree(d_rew);
  }
  int state_size() const { return kind == ENV_CAR2D ? 3 : model.n_links * MBD_LINK_STATE; }
  int action_size() const { return kind == ENV_CAR2D ? 2 : model.n_act; }
  int observation_size() const;
};

struct mbd_plan {
  mbd_env* env = nullptr;
  hipStream_t last_stream = nullptr;  // stream of the plan's previous phase call (plan_enter orders a change of stream)
  bool last_stream_set = false;
  hipEvent_t ev_xs = nullptr;
  mbd_plan_config cfg;
  int HNu = 0;
  std::vector<float> alphas, alphas_bar, sigmas;
  hipStream_t stream = nullptr;
  // second stream: the non-lazy sharded sampler's other-rank rows, and the next step's normals of lazy plans whose
  // rollout fills the chip (smaller rollouts generate them in spare workgroups of their own launch)
  hipStream_t aux = nullptr;
  hipEvent_t ev_in = nullptr, ev_aux = nullptr;
  bool aux_pending = false;
  float *d_state0 = nullptr, *d_Y0s = nullptr, *d_rewss = nullptr, *d_rews = nullptr, *d_lp = nullptr;
  float *d_xpos = nullptr, *d_weights = nullptr, *d_Ybar = nullptr, *d_mu = nullptr, *d_rewmeans = nullptr;
  float *d_scratch = nullptr;
  float* d_wm_partial = nullptr;  // [64][HNu] partials of the split weighted mean (plans of >= 4096 candidates)
  float* d_lg = nullptr;          // [N] logp0 scratch of the score kernel for plans beyond kLdsN candidates
  // LAZY plans (the MBD update on a rigid-body env): the candidates are never materialised.  d_eps[b] holds the normals
  // eps [N][HNu] of a diffusion step; the rollout's action fetch and the weighted mean form clip(eps sigma_i + Ybar_i)
  // on the fly (RolloutParams).  A ring of buffers: while step k reads one, the normals of step k+1 (they depend on that
  // step's key only) are generated into the next — by spare workgroups of step k's rollout launch, or on the aux
  // stream when that launch fills the chip (mbd_plan_prefetch_noise declares the key).  THREE buffers, so that the aux
  // stream needs no event from the step's stream while the caller keeps in step with the device: the buffer step k+1's
  // normals go into was last read by step k-2's weighted mean, which has finished once the rollout of step k-1 has
  // STARTED — every rollout launch of the plan stores its sequence number into h_progress (pinned host memory) as it
  // starts, and the host looks there.  A caller that runs ahead of the device (mbd_plan_run's loop, the async leg of the
  // bench) gets the event-ordered form: a mark on the step's stream in front of the rollout, a wait on the aux stream.
  bool lazy = false;
  float* d_eps[3] = {nullptr, nullptr, nullptr};
  int eps_cur = 0;                 // buffer of the step in flight (set by sample_rollout, read by score_update / peek)
  uint32_t eps_key[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  bool eps_valid[3] = {false, false, false};   // d_eps[b] holds normal(eps_key[b])
  bool eps_on_aux[3] = {false, false, false};  // ... generated on the aux stream: the reader checks ev_noise[b] first
  int eps_read_seq[3] = {0, 0, 0};             // sequence number of the last rollout launch that read d_eps[b]
  int* h_progress = nullptr;
  int seq = 0;
  bool in_step = false;        // the last sample_rollout found the host in step with the device (a per-step host read)
  bool wm_mark_valid = false;  // ev_wm was recorded behind the latest weighted mean
  bool kept_in_step = false;   // plan_keep_in_step held the host back for the coming sample_rollout (the queue is NOT draining)
  uint32_t hint_key[2] = {0, 0};        // mbd_plan_prefetch_noise: key of the step after the next sample_rollout
  bool hint_valid = false;
  float* d_ybar_keep = nullptr;    // [HNu] Ybar_i of the last finished step (mbd_plan_peek materialises Y0s from it)
  const float* peek_ybar = nullptr;  // the caller's d_Ybar_i between phase 1 and phase 2 of a step, d_ybar_keep after
  float sigma_last = 0.0f;
  hipEvent_t ev_noise[3] = {nullptr, nullptr, nullptr}, ev_wm = nullptr;
  float *d_sigma = nullptr, *d_spread = nullptr;  // path-integral plans
  int* d_idx = nullptr;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  mbd_plan() = default;
  mbd_plan(const mbd_plan&) = delete;
  mbd_plan& operator=(const mbd_plan&) = delete;
  ~mbd_plan() {  // owns its device buffers, streams and events
    if (env) (void)hipSetDevice(env->device);
    if (ev_xs) (void)hipEventDestroy(ev_xs);
    for (auto& ev : events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipFree(d_state0); (void)hipFree(d_Y0s); (void)hipFree(d_rewss); (void)hipFree(d_rews);
    (void)hipFree(d_lp); (void)hipFree(d_xpos); (void)hipFree(d_weights); (void)hipFree(d_Ybar);
    (void)hipFree(d_mu); (void)hipFree(d_rewmeans); (void)hipFree(d_scratch);
    (void)hipFree(d_sigma); (void)hipFree(d_spread); (void)hipFree(d_idx); (void)hipFree(d_wm_partial); (void)hipFree(d_lg);
    (void)hipFree(d_eps[0]); (void)hipFree(d_eps[1]); (void)hipFree(d_eps[2]); (void)hipFree(d_ybar_keep);
    if (h_progress) (void)hipHostFree(h_progress);
    for (int b = 0; b < 3; ++b)
      if (ev_noise[b]) (void)hipEventDestroy(ev_noise[b]);
    if (ev_wm) (void)hipEventDestroy(ev_wm);
    if (stream) (void)hipStreamDestroy(stream);
    if (aux) (void)hipStreamDestroy(aux);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_aux) (void)hipEventDestroy(ev_aux);
  }
};

// Lazy candidates of a plan's rollout (RolloutParams): d_us holds normals, actions are formed at the fetch; and the
// optional noise job for the NEXT step, which rides in spare workgroups of the launch when the rollout leaves CUs idle.
struct LazyArgs {
  const float* ybar = nullptr;  // Ybar_i [H][Nu]
  float sigma = 0.0f;
  float* nz_out = nullptr;      // [nz_N][nz_HNu] normals of the next step, or nullptr: no job
  uint32_t nz_key[2] = {0, 0};
  int nz_impl = 0, nz_N = 0, nz_HNu = 0;
  bool nz_fused = false;        // out: the job went into this launch (false: the caller runs it elsewhere)
  int* progress = nullptr;      // RolloutParams.progress / progress_val
  int progress_val = 0;
};
// the host's view of a progress word (pinned host memory the device stores into)
static inline int progress_read(const int* h) { return __atomic_load_n(h, __ATOMIC_ACQUIRE); }
// how long a host loop that keeps step with the device through the progress word waits before it falls back to ordering
// the streams with an event (a slow stream is not an error; steady_clock also counts system pauses)
constexpr int kInStepWaitMs = 20;
// ---- defined in mbd_env.hip ------------------------------------------------------------------------------------------
// test / A-B levers (one process-wide table, include/mbd_hip_debug.h): -1 = not set
int lever(const char* name);
bool env_flag(const char* name);
// launch of the env's rollout instantiation; sweep = {plan_N, plan_state_stride, plan_ybar_stride} (RolloutParams), or nullptr
// d_lp: the demo log-densities [B] accumulated inside the rollout (RolloutParams::lp) — only where rollout_fuses_logpd says so
int launch_rollout(mbd_env* env, const float* d_state0, const float* d_us, int B, int H, float* d_rewss, float* d_rews,
                   float* d_xpos, float* d_state_final, hipStream_t stream, LazyArgs* lz = nullptr, const int* sweep = nullptr,
                   float* d_lp = nullptr);
// whether the instantiation such a launch runs accumulates HumanoidTrack.eval_xref_logpd itself (the tracking reward compiled
// in, one candidate per lane): the caller then passes d_lp instead of d_xpos and skips launch_logpd.  MBD_NO_FUSED_LOGPD = 1: never.
bool rollout_fuses_logpd(const mbd_env* env, int B, int H, const int* sweep = nullptr);
// whether the device is a whole 8-XCD part (the premise of the XCD-pinned launch forms)
bool device_has_eight_xcds(const mbd_env* env);
// whether a rollout launch of B candidates takes the next step's normals into spare workgroups
bool rollout_fuses_noise(const mbd_env* env, int B, bool allow_pk2 = true);  // (allow_pk2: false for sweeps whose plans hold an odd candidate count — launch_rollout)
int launch_logpd(const mbd_env* e, const float* d_xpos, int B, int H, float* d_out, hipStream_t s);
// ---- defined in mbd_plan.hip -----------------------------------------------------------------------------------------
// noise schedule (mbd_planner.py:84-87)
void host_schedule(float beta0, float betaT, int Nd, std::vector<float>& alphas, std::vector<float>& alphas_bar,
                   std::vector<float>& sigmas);
