// mbd_capi.hip — C ABI of libmbd_hip.so (include/mbd_hip.h): env / plan handles, host-side reset
// (forward kinematics, once per run) and the launch sequence of one reverse-diffusion step.
// There is NO CPU fallback: without a gfx950 device every compute entry returns MBD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mbd_hip.h"
#include "mbd_kernels.h"

using namespace mbd;

namespace {

thread_local std::string g_err;
unsigned long long* g_dbg_clock = nullptr;  // set by mbd_debug_set_clock_buffer (tools/probes only)

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(MBD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

int device_count_quiet() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---- host PRNG (jax.random.split) --------------------------------------------------------------------
void host_split(const uint32_t key[2], int num, int impl, uint32_t* keys) {
  if (impl == MBD_PRNG_PARTITIONABLE) {
    for (int j = 0; j < num; ++j) threefry2x32(key[0], key[1], 0u, (uint32_t)j, keys[2 * j], keys[2 * j + 1]);
    return;
  }
  for (int e = 0; e < 2 * num; ++e) keys[e] = random_bits32(key[0], key[1], 0, (uint64_t)e, (uint64_t)(2 * num));
}

enum EnvKind { ENV_CAR2D = 0, ENV_MODEL = 1 };

}  // namespace

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
  bool diag_inertia = true;  // every body-frame inverse-inertia tensor is exactly diagonal
  bool slides = false;
  bool slide_limits = false;  // any slide dof with a finite range
  // DPP layout (kernels.h "lane exchange without the LDS"): lane <-> link tables when the tree fits the shifts
  int dpp_family = -1;  // index into kDppFamilies, -1: shuffles
  signed char lane_tab[32];
  signed char* d_lane_tab = nullptr;
  // scratch for the single-env step path
  float *d_s_in = nullptr, *d_act = nullptr, *d_s_out = nullptr, *d_rew = nullptr;
  mbd_env() = default;
  mbd_env(const mbd_env&) = delete;
  mbd_env& operator=(const mbd_env&) = delete;
  ~mbd_env() {  // owns its device buffers: every exit of the create functions, early or not, releases them
    (void)hipSetDevice(device);
    (void)hipFree(d_model); (void)hipFree(d_xref); (void)hipFree(d_lane_tab);
    (void)hipFree(d_s_in); (void)hipFree(d_s_out); (void)hipFree(d_act); (void)hipFree(d_rew);
  }
  int state_size() const { return kind == ENV_CAR2D ? 3 : model.n_links * MBD_LINK_STATE; }
  int action_size() const { return kind == ENV_CAR2D ? 2 : model.n_act; }
  int observation_size() const { return kind == ENV_CAR2D ? 3 : model.n_q + model.n_qd; }
};

struct mbd_plan {
  mbd_env* env = nullptr;
  mbd_plan_config cfg;
  int HNu = 0;
  std::vector<float> alphas, alphas_bar, sigmas;
  hipStream_t stream = nullptr;
  // sharded plans: the other ranks' candidate rows are sampled on a second stream while the rollout runs
  hipStream_t aux = nullptr;
  hipEvent_t ev_in = nullptr, ev_aux = nullptr;
  bool aux_pending = false;
  float *d_state0 = nullptr, *d_Y0s = nullptr, *d_rewss = nullptr, *d_rews = nullptr, *d_lp = nullptr;
  float *d_xpos = nullptr, *d_weights = nullptr, *d_Ybar = nullptr, *d_mu = nullptr, *d_rewmeans = nullptr;
  float *d_scratch = nullptr;
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
    for (auto& ev : events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipFree(d_state0); (void)hipFree(d_Y0s); (void)hipFree(d_rewss); (void)hipFree(d_rews);
    (void)hipFree(d_lp); (void)hipFree(d_xpos); (void)hipFree(d_weights); (void)hipFree(d_Ybar);
    (void)hipFree(d_mu); (void)hipFree(d_rewmeans); (void)hipFree(d_scratch);
    (void)hipFree(d_sigma); (void)hipFree(d_spread); (void)hipFree(d_idx);
    if (stream) (void)hipStreamDestroy(stream);
    if (aux) (void)hipStreamDestroy(aux);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_aux) (void)hipEventDestroy(ev_aux);
  }
};

namespace {

// DPP layouts the kernels are instantiated for: lane(parent) = lane(s-th child) + D[s] (0: the family has no such
// slot).  A model qualifies for a family if some root lane puts every link on a distinct lane of its LPS-lane group.
constexpr int kDppD0 = 1, kDppD1 = -4, kDppD2 = -6;  // humanoid family: up to three children per link
constexpr int kDppFamilies[4][4] = {{kDppD0, kDppD1, kDppD2, 0},
                                    {1, -3, 0, 0} /* two-legged planar */,
                                    {1, 0, 0, 0} /* chain */,
                                    {1, -2, -4, -6} /* ant: four two-link legs */};
bool find_dpp_layout(const mbd_model_t& m, int lps, const int D[4], signed char tab[32]) {
  const int L = m.n_links;
  if (L > lps) return false;
  for (int root = 0; root < lps; ++root) {
    int lane[MBD_MAX_LINKS];
    bool used[16] = {false}, ok = true;
    for (int l = 0; l < L && ok; ++l) {
      if (m.parent[l] < 0) {
        if (l != 0) { ok = false; break; }  // one tree, rooted at link 0
        lane[l] = root;
      } else {
        if (m.parent[l] >= l) { ok = false; break; }
        int slot = 0;
        for (int c = 0; c < l; ++c) slot += m.parent[c] == m.parent[l] ? 1 : 0;
        if (slot > 3 || D[slot] == 0) { ok = false; break; }
        lane[l] = lane[m.parent[l]] - D[slot];
      }
      if (lane[l] < 0 || lane[l] >= lps || used[lane[l]]) { ok = false; break; }
      used[lane[l]] = true;
    }
    if (!ok) continue;
    for (int i = 0; i < 32; ++i) tab[i] = -1;
    for (int l = 0; l < L; ++l) { tab[lane[l]] = (signed char)l; tab[16 + l] = (signed char)lane[l]; }
    return true;
  }
  return false;
}

// One launch site for every instantiation.  lds > 0 reserves dynamic LDS the kernel never touches: more than half
// of a CU's 160 KB keeps a second workgroup — of this or of a concurrent plan's launch — off the CU, so concurrent
// plans spread over the chip instead of piling onto the CUs the dispatcher fills first (tools/gpu_concurrent.sh).
template <typename K>
void launch_rollout_kernel(K kernel, int device, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                           const RolloutParams& P) {
  static bool raised[16] = {false};  // per instantiation and device: allow > 64 KB of dynamic LDS
  if (lds > 0 && device >= 0 && device < 16 && !raised[device]) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    raised[device] = true;
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, P);
}

int launch_rollout(mbd_env* env, const float* d_state0, const float* d_us, int B, int H, float* d_rewss,
                   float* d_rews, float* d_xpos, float* d_state_final, hipStream_t stream) {
  if (B <= 0 || H <= 0) return fail(MBD_ERR_INVALID, "rollout: B=%d H=%d", B, H);
  if (env->kind == ENV_CAR2D) {
    Car2dParams P{d_state0, d_us, d_rewss, d_rews, d_xpos, d_state_final, B, H};
    hipLaunchKernelGGL(car2d_rollout_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, P);
    HIP_TRY(hipGetLastError());
    return MBD_OK;
  }
  RolloutParams P{env->d_model, d_state0, d_us, d_rewss, d_rews, d_xpos, d_state_final, B, H,
                  env->slide_limits ? 1 : 0, env->max_children, env->max_rot, env->d_lane_tab, g_dbg_clock};
  const bool iso = env->model.iso_inertia != 0;
  const int spw = 64 / env->lps;
  const int waves = (B + spw - 1) / spw;
  // Wavefronts per workgroup (they are independent).  Four: the four waves of a workgroup land on the four SIMDs
  // of one CU, which single-wave workgroups do not achieve once a launch (or several concurrent plans) puts four
  // waves on a CU — N=4096: 1.29 ms with one wave per workgroup, 0.75 ms with four; N=1024: 0.720 -> 0.710 ms
  // (tools/gpu_wpb.sh).  MBD_WPB overrides for experiments.
  static const int wpb_env = [] { const char* w = std::getenv("MBD_WPB"); return w ? std::atoi(w) : 0; }();
  static const long lds_env = [] { const char* r = std::getenv("MBD_LDS_RESERVE"); return r ? std::atol(r) : -1L; }();
  const int wpb = wpb_env == 1 || wpb_env == 2 ? wpb_env : 4;
  dim3 grid((waves + wpb - 1) / wpb), block(64 * wpb);
  // up to one workgroup per CU: keep the CU to that workgroup (see launch_rollout_kernel); above, CUs are shared
  const size_t lds = lds_env >= 0 ? (size_t)lds_env : ((wpb == 4 && grid.x <= 256) ? 96 * 1024 : 0);
// rollout_kernel<LPS, ISO, SLIDES, MAXCH, MAXCOL, D0, D1, D2, D3, DIAG, MULTI>: L(...) fixes everything but MULTI,
// which follows the model (a joint with more than one hinge dof).  The humanoid-shaped instantiations are MULTI.
#define MBD_LAUNCH(...) \
  launch_rollout_kernel(rollout_kernel<__VA_ARGS__>, env->device, grid, block, lds, stream, P)
#define L(...)                                              \
  do {                                                      \
    if (multi) MBD_LAUNCH(__VA_ARGS__, true);               \
    else MBD_LAUNCH(__VA_ARGS__, false);                    \
  } while (0)
  const bool multi = env->max_rot > 1, diag = env->diag_inertia;
  const bool humanoid_shape = env->lps == 16 && iso && !env->slides && env->max_children <= 3 && multi;
  const bool dpp_h = env->dpp_family == 0;
  if (humanoid_shape && env->max_col <= 1 && dpp_h) {
    MBD_LAUNCH(16, true, false, 3, 1, kDppD0, kDppD1, kDppD2, 0, false, true);
  } else if (humanoid_shape && env->max_col <= 5 && dpp_h) {
    MBD_LAUNCH(16, true, false, 3, 5, kDppD0, kDppD1, kDppD2, 0, false, true);
  } else if (humanoid_shape && env->max_col <= 1) {
    MBD_LAUNCH(16, true, false, 3, 1, 0, 0, 0, 0, false, true);  // humanoid-like trees that do not fit the DPP shifts
  } else if (humanoid_shape && env->max_col <= 5) {
    MBD_LAUNCH(16, true, false, 3, 5, 0, 0, 0, 0, false, true);  // humanoidstandup: up to 5 colliders on one link
  } else if (env->lps == 16 && iso && !env->slides && env->max_col <= 2 && env->dpp_family == 3) {
    L(16, true, false, 4, 2, 1, -2, -4, -6, false);  // ant
  } else if (env->lps == 16 && iso && !env->slides && env->max_col <= 2) {
    L(16, true, false, 4, 2, 0, 0, 0, 0, false);  // ant-like: free root, no slide / weld joints
  } else if (env->lps == 16) {
    if (iso) L(16, true, true, 4, 2, 0, 0, 0, 0, false); else L(16, false, true, 4, 2, 0, 0, 0, 0, false);
  } else if (env->lps == 8 && env->dpp_family == 1) {  // walker2d, halfcheetah
    if (iso) L(8, true, true, 4, 2, 1, -3, 0, 0, false);
    else if (diag) L(8, false, true, 4, 2, 1, -3, 0, 0, true);  // walker2d
    else L(8, false, true, 4, 2, 1, -3, 0, 0, false);
  } else if (env->lps == 8 && env->dpp_family == 2) {
    if (iso) L(8, true, true, 4, 2, 1, 0, 0, 0, false); else L(8, false, true, 4, 2, 1, 0, 0, 0, false);
  } else if (env->lps == 8) {
    if (iso) L(8, true, true, 4, 2, 0, 0, 0, 0, false); else L(8, false, true, 4, 2, 0, 0, 0, 0, false);
  } else if (env->dpp_family == 2) {  // hopper, cartpole
    if (iso) L(4, true, true, 4, 2, 1, 0, 0, 0, false);
    else if (diag) L(4, false, true, 4, 2, 1, 0, 0, 0, true);  // hopper
    else L(4, false, true, 4, 2, 1, 0, 0, 0, false);
  } else {
    if (iso) L(4, true, true, 4, 2, 0, 0, 0, 0, false); else L(4, false, true, 4, 2, 0, 0, 0, 0, false);
  }
#undef L
#undef MBD_LAUNCH
  HIP_TRY(hipGetLastError());
  return MBD_OK;
}

// host forward kinematics: kinematics.forward + com.from_world (pipeline_init)
void host_forward(const mbd_model_t& m, const float* q, const float* qd, float* state) {
  const int L = m.n_links;
  v3 Xp_[MBD_MAX_LINKS], V[MBD_MAX_LINKS], W[MBD_MAX_LINKS];
  q4 Xr_[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) {
    const int p = m.parent[l];
    v3 Pp = mk3(0, 0, 0), Pv = mk3(0, 0, 0), Pw = mk3(0, 0, 0);
    q4 Pr = q4{1, 0, 0, 0};
    if (p >= 0) { Pp = Xp_[p]; Pr = Xr_[p]; Pv = V[p]; Pw = W[p]; }
    const float* ql = q + m.q_idx[l];
    const float* qdl = qd + m.qd_idx[l];
    if (m.n_rot[l] < 0) {
      Xp_[l] = mk3(ql[0], ql[1], ql[2]);
      Xr_[l] = qnormalize(q4{ql[3], ql[4], ql[5], ql[6]});
      V[l] = mk3(qdl[0], qdl[1], qdl[2]);
      W[l] = mk3(qdl[3], qdl[4], qdl[5]);
    } else {
      v3 jpos = mk3(0, 0, 0), sv = mk3(0, 0, 0), wrel = mk3(0, 0, 0);
      q4 jrot = q4{1, 0, 0, 0};
      const int ns = m.n_slide[l], nr = m.n_rot[l];
      for (int k = 0; k < ns; ++k) {
        v3 a = mk3(m.slide_axis_body[l][k][0], m.slide_axis_body[l][k][1], m.slide_axis_body[l][k][2]);
        jpos = axpy(ql[k], a, jpos);
        sv = axpy(qdl[k], a, sv);
      }
      for (int k = 0; k < nr; ++k) {
        v3 a = mk3(m.rot_axis[l][k][0], m.rot_axis[l][k][1], m.rot_axis[l][k][2]);
        v3 ac = rot(a, jrot);
        wrel = axpy(qdl[ns + k], ac, wrel);
        float s, c;
        sincos_(0.5f * ql[ns + k], &s, &c);
        jrot = qmul(jrot, q4{c, s * a.x, s * a.y, s * a.z});
      }
      v3 jp = mk3(m.joint_pos[l][0], m.joint_pos[l][1], m.joint_pos[l][2]);
      v3 t = rot(jp, jrot);
      jpos = mk3(jpos.x + (jp.x - t.x), jpos.y + (jp.y - t.y), jpos.z + (jp.z - t.z));
      v3 lp = mk3(m.link_pos[l][0], m.link_pos[l][1], m.link_pos[l][2]);
      q4 lr = q4{m.link_rot[l][0], m.link_rot[l][1], m.link_rot[l][2], m.link_rot[l][3]};
      v3 lpos = add(lp, rot(jpos, lr));
      q4 lrot = qmul(lr, jrot);
      Xp_[l] = add(Pp, rot(lpos, Pr));
      Xr_[l] = qmul(Pr, lrot);
      W[l] = add(Pw, rot(rot(wrel, lr), Pr));
      v3 rA = rot(jp, Xr_[l]);
      v3 A = add(Xp_[l], rA);
      v3 vA = add(Pv, cross(Pw, sub(A, Pp)));
      vA = add(vA, rot(rot(sv, lr), Pr));
      V[l] = sub(vA, cross(W[l], rA));
    }
    v3 c = mk3(m.com[l][0], m.com[l][1], m.com[l][2]);
    v3 rc = rot(c, Xr_[l]);
    v3 pi = add(Xp_[l], rc);
    v3 vi = add(V[l], cross(W[l], rc));
    float* o = state + l * MBD_LINK_STATE;
    o[0] = pi.x; o[1] = pi.y; o[2] = pi.z;
    o[3] = Xr_[l].w; o[4] = Xr_[l].x; o[5] = Xr_[l].y; o[6] = Xr_[l].z;
    o[7] = vi.x; o[8] = vi.y; o[9] = vi.z;
    o[10] = W[l].x; o[11] = W[l].y; o[12] = W[l].z;
  }
}

int check_model(const mbd_model_t& m) {
  if (m.n_links < 1 || m.n_links > MBD_MAX_LINKS) return fail(MBD_ERR_INVALID, "n_links=%d", m.n_links);
  if (m.n_act < 0 || m.n_act > MBD_MAX_ACT) return fail(MBD_ERR_INVALID, "n_act=%d", m.n_act);
  if (m.n_col < 0 || m.n_col > MBD_MAX_COL) return fail(MBD_ERR_INVALID, "n_col=%d", m.n_col);
  if (m.n_frames < 1) return fail(MBD_ERR_INVALID, "n_frames=%d", m.n_frames);
  for (int l = 0; l < m.n_links; ++l) {
    if (m.parent[l] >= l) return fail(MBD_ERR_INVALID, "link %d: parent %d must precede it", l, m.parent[l]);
    if (m.n_rot[l] < 0 && m.parent[l] >= 0) return fail(MBD_ERR_INVALID, "free joint below the root");
  }
  // at most one actuator per dof (the kernel keeps one (index, gear) pair per dof slot)
  for (int a = 0; a < m.n_act; ++a)
    for (int b = a + 1; b < m.n_act; ++b)
      if (m.act_link[a] == m.act_link[b] && m.act_slot[a] == m.act_slot[b])
        return fail(MBD_ERR_UNSUPPORTED, "two actuators on one dof");
  return MBD_OK;
}

}  // namespace

// ==================================================================================================
// library
// ==================================================================================================
extern "C" const char* mbd_last_error(void) { return g_err.c_str(); }
extern "C" int mbd_version(void) { return 1; }
// undocumented probe hook (not in include/mbd_hip.h): per-workgroup start/end ticks of the rollout kernel
extern "C" int mbd_debug_set_clock_buffer(void* d_buf) { g_dbg_clock = (unsigned long long*)d_buf; return MBD_OK; }
// undocumented test hook (host logic only, no device needed): the DPP lane layout a model would get.
// Returns the family index (kDppFamilies) or -1; tab[0..15] lane -> link, tab[16..31] link -> lane.
extern "C" int mbd_debug_dpp_layout(const mbd_model_t* model, signed char tab[32], int shifts_out[4]) {
  if (!model || !tab || !shifts_out) return -1;
  const int lps = model->n_links <= 4 ? 4 : (model->n_links <= 8 ? 8 : 16);
  const int order16[2] = {0, 3}, order_small[2] = {2, 1};
  for (int t = 0; t < 2; ++t) {
    const int fam = lps == 16 ? order16[t] : order_small[t];
    if (find_dpp_layout(*model, lps, kDppFamilies[fam], tab)) {
      for (int k = 0; k < 4; ++k) shifts_out[k] = kDppFamilies[fam][k];
      return fam;
    }
  }
  return -1;
}
extern "C" int mbd_device_count(int* count) {
  if (!count) return fail(MBD_ERR_INVALID, "count is NULL");
  *count = device_count_quiet();
  return MBD_OK;
}

extern "C" int mbd_prng_key(uint64_t seed, uint32_t key_out[2]) {
  if (!key_out) return fail(MBD_ERR_INVALID, "key_out is NULL");
  key_out[0] = (uint32_t)(seed >> 32);
  key_out[1] = (uint32_t)(seed & 0xffffffffu);
  return MBD_OK;
}
extern "C" int mbd_prng_split(const uint32_t key[2], int num, int impl, uint32_t* keys_out) {
  if (!key || !keys_out || num < 1) return fail(MBD_ERR_INVALID, "prng_split arguments");
  if (impl != MBD_PRNG_LEGACY && impl != MBD_PRNG_PARTITIONABLE) return fail(MBD_ERR_INVALID, "impl=%d", impl);
  host_split(key, num, impl, keys_out);
  return MBD_OK;
}

// ==================================================================================================
// environments
// ==================================================================================================
static int env_common_init(mbd_env* e) {
  const int S = e->state_size(), A = e->action_size();
  HIP_TRY(hipMalloc(&e->d_s_in, sizeof(float) * S));
  HIP_TRY(hipMalloc(&e->d_s_out, sizeof(float) * S));
  HIP_TRY(hipMalloc(&e->d_act, sizeof(float) * (A > 0 ? A : 1)));
  HIP_TRY(hipMalloc(&e->d_rew, sizeof(float)));
  return MBD_OK;
}

extern "C" int mbd_env_create_car2d(int device, const float* xref, mbd_env** out) {
  if (!out) return fail(MBD_ERR_INVALID, "out is NULL");
  const int n = device_count_quiet();
  if (n == 0) return fail(MBD_ERR_NO_DEVICE, "no gfx950 device visible (libmbd_hip has no CPU fallback)");
  if (device < 0 || device >= n) return fail(MBD_ERR_INVALID, "device %d of %d", device, n);
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<mbd_env> guard(new mbd_env());
  mbd_env* e = guard.get();
  e->kind = ENV_CAR2D;
  e->device = device;
  e->name = "car2d";
  memset(&e->model, 0, sizeof(e->model));
  if (xref) {
    HIP_TRY(hipMalloc(&e->d_xref, sizeof(float) * 50 * 2));
    HIP_TRY(hipMemcpy(e->d_xref, xref, sizeof(float) * 50 * 2, hipMemcpyHostToDevice));
    e->has_xref = true;
    // rew_xref = mean over the 50 demo points of get_reward (car2d.py:71); jnp mean = sum / 50
    float s = 0.0f;
    for (int i = 0; i < 50; ++i) {
      float dx = xref[2 * i] - 0.5f, dy = xref[2 * i + 1] - 0.0f;
      float d = fsqrt(dx * dx + dy * dy);
      d = fclip(d, 0.0f, 0.2f);
      float t = d / 0.2f;
      s = s + (1.0f - t * t);
    }
    e->rew_xref = s / 50.0f;
  }
  int rc = env_common_init(e);
  if (rc != MBD_OK) return rc;
  *out = guard.release();
  return MBD_OK;
}

extern "C" int mbd_env_create_model(const char* env_name, int device, const mbd_model_t* model,
                                    const float* xref, float rew_xref, mbd_env** out) {
  if (!out || !model || !env_name) return fail(MBD_ERR_INVALID, "NULL argument");
  int rc = check_model(*model);
  if (rc != MBD_OK) return rc;
  const int n = device_count_quiet();
  if (n == 0) return fail(MBD_ERR_NO_DEVICE, "no gfx950 device visible (libmbd_hip has no CPU fallback)");
  if (device < 0 || device >= n) return fail(MBD_ERR_INVALID, "device %d of %d", device, n);
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<mbd_env> guard(new mbd_env());
  mbd_env* e = guard.get();
  e->kind = ENV_MODEL;
  e->device = device;
  e->name = env_name;
  e->model = *model;
  const mbd_model_t& m = e->model;
  e->lps = m.n_links <= 4 ? 4 : (m.n_links <= 8 ? 8 : 16);
  int nch[MBD_MAX_LINKS] = {0}, ncl[MBD_MAX_LINKS] = {0};
  for (int l = 0; l < m.n_links; ++l) {
    if (m.parent[l] >= 0) nch[m.parent[l]]++;
    if (m.n_slide[l] > 0 || m.n_rot[l] == 0) e->slides = true;  // slides and welds both need the generic kernels
    if (m.n_rot[l] > e->max_rot) e->max_rot = m.n_rot[l];
    if (m.inv_inertia[l][3] != 0.0f || m.inv_inertia[l][4] != 0.0f || m.inv_inertia[l][5] != 0.0f) e->diag_inertia = false;
    for (int k = 0; k < m.n_slide[l]; ++k)
      if (m.slide_lo[l][k] > -1e8f || m.slide_hi[l][k] < 1e8f) e->slide_limits = true;
  }
  for (int k = 0; k < m.n_col; ++k) ncl[m.col_link[k]]++;
  for (int l = 0; l < m.n_links; ++l) {
    if (nch[l] > e->max_children) e->max_children = nch[l];
    if (ncl[l] > e->max_col) e->max_col = ncl[l];
  }
  if (e->max_children > kMaxChildren) { return fail(MBD_ERR_UNSUPPORTED, "a link has %d children > %d", e->max_children, kMaxChildren); }
  {
    const bool standup_shape = e->lps == 16 && m.iso_inertia && !e->slides && e->max_children <= 3 && e->max_rot > 1;
    if (e->max_col > (standup_shape ? 5 : 2)) {
      const int mc = e->max_col;
      return fail(MBD_ERR_UNSUPPORTED, "a link has %d sphere colliders: more than this kernel family is built for", mc);
    }
  }
  // smallest family first: a chain also fits the wider layouts, but their kernels spend VALU slots on empty slots
  {
    // MBD_NO_DPP=1 keeps the shuffle (ds_bpermute) exchange: the fallback for trees that fit no DPP family,
    // exercised by the test-suite this way
    const char* no_dpp = std::getenv("MBD_NO_DPP");
    const int order16[2] = {0, 3}, order_small[2] = {2, 1};
    for (int t = 0; t < 2 && e->dpp_family < 0 && !(no_dpp && no_dpp[0] == '1'); ++t) {
      const int fam = e->lps == 16 ? order16[t] : order_small[t];
      if (find_dpp_layout(m, e->lps, kDppFamilies[fam], e->lane_tab)) e->dpp_family = fam;
    }
  }
  if (e->dpp_family < 0)
    for (int i = 0; i < 32; ++i) e->lane_tab[i] = (signed char)(i & 15);  // identity (unused by the other kernels)
  HIP_TRY(hipMalloc(&e->d_lane_tab, sizeof(e->lane_tab)));
  HIP_TRY(hipMemcpy(e->d_lane_tab, e->lane_tab, sizeof(e->lane_tab), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&e->d_model, sizeof(mbd_model_t)));
  HIP_TRY(hipMemcpy(e->d_model, &e->model, sizeof(mbd_model_t), hipMemcpyHostToDevice));
  if (xref) {
    if (m.n_track < 1) return fail(MBD_ERR_INVALID, "xref given but n_track = 0");
    const size_t nb = sizeof(float) * (size_t)m.n_track * 50 * 3;
    HIP_TRY(hipMalloc(&e->d_xref, nb));
    HIP_TRY(hipMemcpy(e->d_xref, xref, nb, hipMemcpyHostToDevice));
    e->has_xref = true;
  }
  e->rew_xref = rew_xref;
  rc = env_common_init(e);
  if (rc != MBD_OK) return rc;
  *out = guard.release();
  return MBD_OK;
}

extern "C" int mbd_env_destroy(mbd_env* e) {
  delete e;  // (nullptr is fine)
  return MBD_OK;
}

extern "C" int mbd_env_info(const mbd_env* e, int* action_size, int* observation_size, int* state_size,
                            int* n_links, int* n_frames, float* dt) {
  if (!e) return fail(MBD_ERR_INVALID, "env is NULL");
  if (action_size) *action_size = e->action_size();
  if (observation_size) *observation_size = e->observation_size();
  if (state_size) *state_size = e->state_size();
  if (n_links) *n_links = e->kind == ENV_CAR2D ? 1 : e->model.n_links;
  if (n_frames) *n_frames = e->kind == ENV_CAR2D ? 1 : e->model.n_frames;
  if (dt) *dt = e->kind == ENV_CAR2D ? 0.1f : e->model.dt * (float)e->model.n_frames;
  return MBD_OK;
}

extern "C" int mbd_env_reset(const mbd_env* e, const uint32_t key[2], int impl, float* state_out) {
  if (!e || !key || !state_out) return fail(MBD_ERR_INVALID, "NULL argument");
  if (e->kind == ENV_CAR2D) {  // car2d.py:64,73-75: rng ignored
    state_out[0] = -0.5f;
    state_out[1] = 0.0f;
    state_out[2] = (float)(3.141592653589793 * 3.0 / 2.0);
    return MBD_OK;
  }
  const mbd_model_t& m = e->model;
  float q[MBD_MAX_Q], qd[MBD_MAX_Q];
  for (int i = 0; i < m.n_q; ++i) q[i] = m.init_q[i];
  for (int i = 0; i < m.n_qd; ++i) qd[i] = 0.0f;
  if (m.reset_noise > 0.0f) {
    // rng, rng1, rng2 = split(rng, 3); qpos = init_q + U(rng1, -s, s); qvel = U(rng2, -s, s)
    // (humanoidrun.py:21-27, hopper.py:22-28); halfcheetah: qvel = s * normal(rng2) (brax half_cheetah)
    uint32_t keys[6];
    host_split(key, 3, impl, keys);
    const float s = m.reset_noise;
    for (int i = 0; i < m.n_q; ++i)
      q[i] = m.init_q[i] + bits_to_uniform(random_bits32(keys[2], keys[3], impl, i, m.n_q), -s, s);
    for (int i = 0; i < m.n_qd; ++i) {
      uint32_t bits = random_bits32(keys[4], keys[5], impl, i, m.n_qd);
      qd[i] = (m.reward_kind == MBD_REW_HALFCHEETAH || m.reward_kind == MBD_REW_ANT) ? s * bits_to_normal(bits)
                                                                                  : bits_to_uniform(bits, -s, s);
    }
  }
  host_forward(m, q, qd, state_out);
  return MBD_OK;
}

extern "C" int mbd_env_step(mbd_env* e, const float* state_in, const float* action, float* state_out,
                            float* reward_out, float* obs_out) {
  if (!e || !state_in || !action || !state_out) return fail(MBD_ERR_INVALID, "NULL argument");
  if (obs_out) return fail(MBD_ERR_UNSUPPORTED, "observations are not produced by the hot path (the planner "
                                                "never reads obs, mbd_planner.py:71)");
  HIP_TRY(hipSetDevice(e->device));
  const int S = e->state_size(), A = e->action_size();
  HIP_TRY(hipMemcpy(e->d_s_in, state_in, sizeof(float) * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_act, action, sizeof(float) * A, hipMemcpyHostToDevice));
  int rc = launch_rollout(e, e->d_s_in, e->d_act, 1, 1, e->d_rew, nullptr, nullptr, e->d_s_out, nullptr);
  if (rc != MBD_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(state_out, e->d_s_out, sizeof(float) * S, hipMemcpyDeviceToHost));
  if (reward_out) HIP_TRY(hipMemcpy(reward_out, e->d_rew, sizeof(float), hipMemcpyDeviceToHost));
  return MBD_OK;
}

extern "C" int mbd_env_rew_xref(const mbd_env* e, float* out) {
  if (!e || !out) return fail(MBD_ERR_INVALID, "NULL argument");
  *out = e->rew_xref;
  return MBD_OK;
}

extern "C" int mbd_env_rollout(mbd_env* e, const float* d_state0, const float* d_us, int B, int H,
                               float* d_rewss, float* d_xpos, float* d_state_final, void* stream) {
  if (!e || !d_state0 || !d_us) return fail(MBD_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  return launch_rollout(e, d_state0, d_us, B, H, d_rewss, nullptr, d_xpos, d_state_final, (hipStream_t)stream);
}

// ==================================================================================================
// planner
// ==================================================================================================
static void host_schedule(float beta0, float betaT, int Nd, std::vector<float>& alphas,
                          std::vector<float>& alphas_bar, std::vector<float>& sigmas) {
  // mbd_planner.py:84-87; jnp.linspace = start*(1-t) + stop*t with the endpoint appended
  alphas.resize(Nd); alphas_bar.resize(Nd); sigmas.resize(Nd);
  float cp = 1.0f;
  for (int i = 0; i < Nd; ++i) {
    float t = Nd > 1 ? (float)i / (float)(Nd - 1) : 0.0f;
    float beta = (i == Nd - 1 && Nd > 1) ? betaT : beta0 * (1.0f - t) + betaT * t;
    float a = 1.0f - beta;
    cp = cp * a;
    alphas[i] = a; alphas_bar[i] = cp; sigmas[i] = fsqrt(1.0f - cp);
  }
}

extern "C" int mbd_plan_create(mbd_env* env, const mbd_plan_config* cfg, mbd_plan** out) {
  if (!env || !cfg || !out) return fail(MBD_ERR_INVALID, "NULL argument");
  if (cfg->Nsample < 1 || cfg->Hsample < 1 || cfg->Ndiffuse < 2) return fail(MBD_ERR_INVALID, "Nsample/Hsample/Ndiffuse");
  if (cfg->shard_begin < 0 || cfg->shard_count < 1 || cfg->shard_begin + cfg->shard_count > cfg->Nsample)
    return fail(MBD_ERR_INVALID, "shard [%d,+%d) outside N=%d", cfg->shard_begin, cfg->shard_count, cfg->Nsample);
  if (cfg->update_method < 0 || cfg->update_method > 3) return fail(MBD_ERR_INVALID, "update_method=%d", cfg->update_method);
  if (cfg->update_method > 0 && cfg->enable_demo) return fail(MBD_ERR_INVALID, "path-integral plans do not use demos");
  if (cfg->enable_demo) {
    if (!env->has_xref) return fail(MBD_ERR_INVALID, "enable_demo needs an env created with xref");
    if (cfg->Hsample != 50) return fail(MBD_ERR_INVALID, "demos require Hsample == 50 (xref has 50 rows)");
  }
  if ((size_t)cfg->Nsample * sizeof(float) > 160 * 1024 - 1024) return fail(MBD_ERR_UNSUPPORTED, "Nsample too large for the LDS-resident score kernel");
  if ((size_t)cfg->Nsample * sizeof(float) > 48 * 1024) {  // logp0 [N] in dynamic LDS: beyond the default window
    HIP_TRY(hipSetDevice(env->device));
    HIP_TRY(hipFuncSetAttribute((const void*)score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)cem_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
  }
  HIP_TRY(hipSetDevice(env->device));
  std::unique_ptr<mbd_plan> guard(new mbd_plan());
  mbd_plan* p = guard.get();
  p->env = env;
  p->cfg = *cfg;
  const int N = cfg->Nsample, H = cfg->Hsample, Nu = env->action_size(), Nd = cfg->Ndiffuse, sh = cfg->shard_count;
  p->HNu = H * Nu;
  host_schedule(cfg->beta0, cfg->betaT, Nd, p->alphas, p->alphas_bar, p->sigmas);
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  const int K = env->kind == ENV_CAR2D ? 1 : (env->model.n_track > 0 ? env->model.n_track : 1);
  HIP_TRY(hipMalloc(&p->d_state0, sizeof(float) * env->state_size()));
  HIP_TRY(hipMalloc(&p->d_Y0s, sizeof(float) * (size_t)N * p->HNu));
  HIP_TRY(hipMalloc(&p->d_rewss, sizeof(float) * (size_t)sh * H));
  HIP_TRY(hipMalloc(&p->d_rews, sizeof(float) * (size_t)N));
  HIP_TRY(hipMalloc(&p->d_lp, sizeof(float) * (size_t)N));
  if (cfg->enable_demo) HIP_TRY(hipMalloc(&p->d_xpos, sizeof(float) * (size_t)sh * H * K * 3));
  HIP_TRY(hipMalloc(&p->d_weights, sizeof(float) * (size_t)N));
  HIP_TRY(hipMalloc(&p->d_Ybar, sizeof(float) * (size_t)p->HNu * 2));
  HIP_TRY(hipMalloc(&p->d_mu, sizeof(float) * (size_t)(Nd - 1) * p->HNu));
  HIP_TRY(hipMalloc(&p->d_rewmeans, sizeof(float) * (size_t)Nd));
  HIP_TRY(hipMalloc(&p->d_scratch, sizeof(float) * (size_t)(H + 8)));
  if (cfg->update_method > 0) {
    HIP_TRY(hipMalloc(&p->d_sigma, sizeof(float)));
    HIP_TRY(hipMalloc(&p->d_spread, sizeof(float) * (size_t)p->HNu));
    HIP_TRY(hipMalloc(&p->d_idx, sizeof(int) * 16));
    const float one = 1.0f;  // path_integral.py:131
    HIP_TRY(hipMemcpy(p->d_sigma, &one, sizeof(float), hipMemcpyHostToDevice));
  }
  *out = guard.release();
  return MBD_OK;
}

extern "C" int mbd_plan_destroy(mbd_plan* p) {
  delete p;  // (nullptr is fine)
  return MBD_OK;
}

extern "C" int mbd_plan_schedule(const mbd_plan* p, float* alphas, float* alphas_bar, float* sigmas) {
  if (!p) return fail(MBD_ERR_INVALID, "plan is NULL");
  const size_t nb = sizeof(float) * p->alphas.size();
  if (alphas) memcpy(alphas, p->alphas.data(), nb);
  if (alphas_bar) memcpy(alphas_bar, p->alphas_bar.data(), nb);
  if (sigmas) memcpy(sigmas, p->sigmas.data(), nb);
  return MBD_OK;
}

extern "C" int mbd_plan_set_state0(mbd_plan* p, const float* state0) {
  if (!p || !state0) return fail(MBD_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->env->device));
  HIP_TRY(hipMemcpy(p->d_state0, state0, sizeof(float) * p->env->state_size(), hipMemcpyHostToDevice));
  return MBD_OK;
}

extern "C" int mbd_plan_sample_rollout(mbd_plan* p, int i, const uint32_t key_sample[2], const float* d_Ybar_i,
                                       float* d_rews_local, float* d_logpd_local, void* stream_) {
  if (!p || !key_sample || !d_Ybar_i || !d_rews_local) return fail(MBD_ERR_INVALID, "NULL argument");
  const mbd_plan_config& c = p->cfg;
  if (i < 1 || i >= c.Ndiffuse) return fail(MBD_ERR_INVALID, "diffusion index %d outside [1,%d)", i, c.Ndiffuse);
  if (c.enable_demo && !d_logpd_local) return fail(MBD_ERR_INVALID, "enable_demo needs d_logpd_local");
  mbd_env* e = p->env;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = (hipStream_t)stream_;
  const int N = c.Nsample, H = c.Hsample, HNu = p->HNu;
  // A1: every rank generates ALL N candidate sequences (counter-based noise), so that phase 2 needs no second
  // collective and is bit-identical for every shard layout.  A sharded plan samples its own rows first and the
  // others' on a second stream, behind the rollout (which leaves three quarters of the CUs idle at 1024
  // candidates); mbd_plan_score_update joins that stream before it reads them.
  {
    auto sample = [&](hipStream_t st, uint64_t e0, uint64_t cnt) {
      if (cnt == 0) return;
      const uint64_t size = (uint64_t)N * HNu;
      const bool pair_blocks = c.prng_impl != MBD_PRNG_PARTITIONABLE && e0 == 0 && cnt == size;
      const uint64_t threads = pair_blocks ? (size + 1) / 2 : cnt;
      hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, key_sample[0],
                         key_sample[1], c.prng_impl, N, HNu, (unsigned long long)e0, (unsigned long long)cnt,
                         p->sigmas[i], c.update_method > 0 ? (const float*)p->d_sigma : (const float*)nullptr,
                         d_Ybar_i, p->d_Y0s);
    };
    const uint64_t own0 = (uint64_t)c.shard_begin * HNu, own1 = own0 + (uint64_t)c.shard_count * HNu;
    static const bool no_aux = [] { const char* v = std::getenv("MBD_NO_AUX"); return v && v[0] == '1'; }();
    // worth the two events only when the other ranks' rows dominate (tools/gpu_rank_emu.sh: 8 shards 0.774 ->
    // 0.762 ms per step, 2 shards 0.736 -> 0.742); MBD_NO_AUX=1 keeps everything on the caller's stream (A/B)
    if (c.shard_count == N || no_aux || (long long)N < 5LL * c.shard_count) {
      sample(s, 0, (uint64_t)N * HNu);
    } else {
      if (!p->aux) {
        HIP_TRY(hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p->ev_aux, hipEventDisableTiming));
      }
      HIP_TRY(hipEventRecord(p->ev_in, s));  // Ybar_i is final and the previous step is done with Y0s
      HIP_TRY(hipStreamWaitEvent(p->aux, p->ev_in, 0));
      sample(s, own0, own1 - own0);
      sample(p->aux, 0, own0);
      sample(p->aux, own1, (uint64_t)N * HNu - own1);
      HIP_TRY(hipEventRecord(p->ev_aux, p->aux));
      p->aux_pending = true;
    }
    HIP_TRY(hipGetLastError());
  }
  // A2/A3: rollout of the local shard
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (p->timing) {
    if (p->events_used == p->events.size()) {
      hipEvent_t a, b;
      HIP_TRY(hipEventCreate(&a));
      HIP_TRY(hipEventCreate(&b));
      p->events.emplace_back(a, b);
    }
    ev0 = p->events[p->events_used].first;
    ev1 = p->events[p->events_used].second;
    p->events_used++;
    HIP_TRY(hipEventRecord(ev0, s));
  }
  int rc = launch_rollout(e, p->d_state0, p->d_Y0s + (size_t)c.shard_begin * HNu, c.shard_count, H, p->d_rewss,
                          d_rews_local, c.enable_demo ? p->d_xpos : nullptr, nullptr, s);
  if (rc != MBD_OK) return rc;
  if (p->timing) HIP_TRY(hipEventRecord(ev1, s));
  // A5: demo log-densities of the local shard
  if (c.enable_demo) {
    const int B = c.shard_count;
    if (e->kind == ENV_CAR2D)
      hipLaunchKernelGGL(logpd_car2d_kernel, dim3((B + 63) / 64), dim3(64), 0, s, p->d_xpos, e->d_xref, B, H, d_logpd_local);
    else
      hipLaunchKernelGGL(logpd_track_kernel, dim3((B + 63) / 64), dim3(64), 0, s, p->d_xpos, e->d_xref, B, H,
                         e->model.n_track, d_logpd_local);
    HIP_TRY(hipGetLastError());
  }
  return MBD_OK;
}

extern "C" int mbd_plan_score_update(mbd_plan* p, int i, const uint32_t key_sample[2], const float* d_Ybar_i,
                                     const float* d_rews_all, const float* d_logpd_all, float* d_Ybar_im1,
                                     float* d_rew_mean, void* stream_) {
  (void)key_sample;  // Y0s of all N candidates is already resident from phase 1 of this step
  if (!p || !d_Ybar_i || !d_rews_all || !d_Ybar_im1 || !d_rew_mean) return fail(MBD_ERR_INVALID, "NULL argument");
  const mbd_plan_config& c = p->cfg;
  if (i < 1 || i >= c.Ndiffuse) return fail(MBD_ERR_INVALID, "diffusion index %d outside [1,%d)", i, c.Ndiffuse);
  if (c.enable_demo && !d_logpd_all) return fail(MBD_ERR_INVALID, "enable_demo needs d_logpd_all");
  HIP_TRY(hipSetDevice(p->env->device));
  hipStream_t s = (hipStream_t)stream_;
  const int N = c.Nsample, HNu = p->HNu;
  if (p->aux_pending) {  // the other ranks' rows of Y0s (sampled behind the rollout)
    HIP_TRY(hipStreamWaitEvent(s, p->ev_aux, 0));
    p->aux_pending = false;
  }
  hipLaunchKernelGGL(score_kernel, dim3(1), dim3(kScoreThreads), sizeof(float) * (size_t)N, s, d_rews_all,
                     c.enable_demo ? d_logpd_all : nullptr, N, p->env->rew_xref, c.temp_sample,
                     c.update_method == 0 ? 1 : 0, p->d_weights, d_rew_mean);
  HIP_TRY(hipGetLastError());
  const dim3 ge((HNu + 63) / 64), b64(64);
  if (c.update_method == 3) {  // cem_update (path_integral.py:48-52)
    const int K = N < 10 ? N : 10;
    hipLaunchKernelGGL(cem_select_kernel, dim3(1), b64, sizeof(float) * (size_t)N, s, p->d_weights, N, K, p->d_idx);
    hipLaunchKernelGGL(cem_mean_kernel, ge, b64, 0, s, p->d_idx, K, p->d_Y0s, HNu, d_Ybar_im1);
  } else {  // MBD (:128-133), mppi (:33-36), cma-es (:39-45)
    hipLaunchKernelGGL(wmean_kernel, dim3((HNu + kWmE - 1) / kWmE), dim3(kWmE * kWmG), 0, s, p->d_weights,
                       p->d_Y0s, N, HNu, d_Ybar_i, p->alphas[i],
                       p->alphas_bar[i], p->alphas_bar[i - 1], c.update_method == 0 ? c.literal_score : 0, d_Ybar_im1);
    if (c.update_method == 2) {
      hipLaunchKernelGGL(cma_spread_kernel, ge, b64, 0, s, p->d_weights, p->d_Y0s, N, HNu, d_Ybar_i, p->d_spread);
      hipLaunchKernelGGL(cma_sigma_kernel, dim3(1), b64, 0, s, p->d_spread, HNu, p->d_sigma);
    }
  }
  HIP_TRY(hipGetLastError());
  return MBD_OK;
}

extern "C" int mbd_plan_set_sigma(mbd_plan* p, float sigma) {
  if (!p) return fail(MBD_ERR_INVALID, "plan is NULL");
  if (!p->d_sigma) return fail(MBD_ERR_STATE, "not a path-integral plan (update_method == 0)");
  HIP_TRY(hipSetDevice(p->env->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(p->d_sigma, &sigma, sizeof(float), hipMemcpyHostToDevice));
  return MBD_OK;
}
extern "C" int mbd_plan_get_sigma(mbd_plan* p, float* sigma_out) {
  if (!p || !sigma_out) return fail(MBD_ERR_INVALID, "NULL argument");
  if (!p->d_sigma) return fail(MBD_ERR_STATE, "not a path-integral plan (update_method == 0)");
  HIP_TRY(hipSetDevice(p->env->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(sigma_out, p->d_sigma, sizeof(float), hipMemcpyDeviceToHost));
  return MBD_OK;
}

static int reverse_once_impl(mbd_plan* p, int i, uint32_t key_inout[2], const float* d_Ybar_in, float* d_Ybar_out,
                             float* d_rew_mean, hipStream_t s) {
  if (p->cfg.shard_count != p->cfg.Nsample)
    return fail(MBD_ERR_STATE, "reverse_once on a sharded plan: use sample_rollout + all-gather + score_update");
  uint32_t keys[4];
  host_split(key_inout, 2, p->cfg.prng_impl, keys);  // rng, Y0s_rng = split(rng)  (mbd_planner.py:103)
  const uint32_t ks[2] = {keys[2], keys[3]};
  int rc = mbd_plan_sample_rollout(p, i, ks, d_Ybar_in, p->d_rews, p->cfg.enable_demo ? p->d_lp : nullptr, s);
  if (rc != MBD_OK) return rc;
  rc = mbd_plan_score_update(p, i, ks, d_Ybar_in, p->d_rews, p->d_lp, d_Ybar_out, d_rew_mean, s);
  if (rc != MBD_OK) return rc;
  key_inout[0] = keys[0];
  key_inout[1] = keys[1];
  return MBD_OK;
}

extern "C" int mbd_plan_reverse_once(mbd_plan* p, int i, uint32_t key_inout[2], float* d_Ybar, float* d_rew_mean,
                                     void* stream_) {
  if (!p || !key_inout || !d_Ybar || !d_rew_mean) return fail(MBD_ERR_INVALID, "NULL argument");
  hipStream_t s = (hipStream_t)stream_;
  // the update is not in place on the device (wmean reads Ybar_i while writing Ybar_{i-1})
  int rc = reverse_once_impl(p, i, key_inout, d_Ybar, p->d_Ybar, d_rew_mean, s);
  if (rc != MBD_OK) return rc;
  HIP_TRY(hipMemcpyAsync(d_Ybar, p->d_Ybar, sizeof(float) * p->HNu, hipMemcpyDeviceToDevice, s));
  return MBD_OK;
}

extern "C" int mbd_plan_run(mbd_plan* p, const uint32_t key[2], float* mu_0ts_out, float* rew_means_out,
                            float* rew_final_out, double* loop_seconds_out) {
  if (!p || !key) return fail(MBD_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->env->device));
  const int Nd = p->cfg.Ndiffuse, HNu = p->HNu;
  hipStream_t s = p->stream;
  uint32_t rng[2] = {key[0], key[1]};
  float* cur = p->d_Ybar;  // YN = zeros (mbd_planner.py:95; mu_0T path_integral.py:107)
  if (p->d_sigma) {
    const float one = 1.0f;  // sigma = 1.0 (path_integral.py:131)
    HIP_TRY(hipMemcpy(p->d_sigma, &one, sizeof(float), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemsetAsync(cur, 0, sizeof(float) * HNu, s));
  HIP_TRY(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = Nd - 1; i >= 1; --i) {  // reverse() (mbd_planner.py:138-148)
    float* nxt = p->d_mu + (size_t)(Nd - 1 - i) * HNu;  // Ybars.append(Yi)
    int rc = reverse_once_impl(p, i, rng, cur, nxt, p->d_rewmeans + (Nd - 1 - i), s);
    if (rc != MBD_OK) return rc;
    cur = nxt;
  }
  HIP_TRY(hipStreamSynchronize(s));
  auto t1 = std::chrono::steady_clock::now();
  if (loop_seconds_out) *loop_seconds_out = std::chrono::duration<double>(t1 - t0).count();
  if (mu_0ts_out) HIP_TRY(hipMemcpy(mu_0ts_out, p->d_mu, sizeof(float) * (size_t)(Nd - 1) * HNu, hipMemcpyDeviceToHost));
  if (rew_means_out) HIP_TRY(hipMemcpy(rew_means_out, p->d_rewmeans, sizeof(float) * (size_t)(Nd - 1), hipMemcpyDeviceToHost));
  if (rew_final_out) {  // rollout_us(state_init, Yi[-1]).mean()  (mbd_planner.py:179-180)
    int rc = launch_rollout(p->env, p->d_state0, cur, 1, p->cfg.Hsample, nullptr, p->d_scratch, nullptr, nullptr, s);
    if (rc != MBD_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(rew_final_out, p->d_scratch, sizeof(float), hipMemcpyDeviceToHost));
  }
  return MBD_OK;
}

extern "C" int mbd_plan_eval(mbd_plan* p, const float* Y, float* rew_final_out) {
  if (!p || !Y || !rew_final_out) return fail(MBD_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->env->device));
  HIP_TRY(hipMemcpy(p->d_Ybar + p->HNu, Y, sizeof(float) * p->HNu, hipMemcpyHostToDevice));
  int rc = launch_rollout(p->env, p->d_state0, p->d_Ybar + p->HNu, 1, p->cfg.Hsample, nullptr, p->d_scratch, nullptr,
                          nullptr, p->stream);
  if (rc != MBD_OK) return rc;
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipMemcpy(rew_final_out, p->d_scratch, sizeof(float), hipMemcpyDeviceToHost));
  return MBD_OK;
}

extern "C" int mbd_plan_peek(mbd_plan* p, float* Y0s_out, float* rewss_out, float* weights_out) {
  if (!p) return fail(MBD_ERR_INVALID, "plan is NULL");
  HIP_TRY(hipSetDevice(p->env->device));
  HIP_TRY(hipDeviceSynchronize());
  const mbd_plan_config& c = p->cfg;
  if (Y0s_out) HIP_TRY(hipMemcpy(Y0s_out, p->d_Y0s, sizeof(float) * (size_t)c.Nsample * p->HNu, hipMemcpyDeviceToHost));
  if (rewss_out) HIP_TRY(hipMemcpy(rewss_out, p->d_rewss, sizeof(float) * (size_t)c.shard_count * c.Hsample, hipMemcpyDeviceToHost));
  if (weights_out) HIP_TRY(hipMemcpy(weights_out, p->d_weights, sizeof(float) * (size_t)c.Nsample, hipMemcpyDeviceToHost));
  return MBD_OK;
}

extern "C" int mbd_plan_enable_timing(mbd_plan* p, int enable) {
  if (!p) return fail(MBD_ERR_INVALID, "plan is NULL");
  p->timing = enable != 0;
  return MBD_OK;
}

extern "C" int mbd_plan_kernel_time(mbd_plan* p, float* avg_ms_out, int* count_out, int reset) {
  if (!p) return fail(MBD_ERR_INVALID, "plan is NULL");
  HIP_TRY(hipSetDevice(p->env->device));
  HIP_TRY(hipDeviceSynchronize());
  double tot = 0.0;
  for (size_t k = 0; k < p->events_used; ++k) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, p->events[k].first, p->events[k].second));
    tot += ms;
  }
  if (avg_ms_out) *avg_ms_out = p->events_used ? (float)(tot / (double)p->events_used) : 0.0f;
  if (count_out) *count_out = (int)p->events_used;
  if (reset) p->events_used = 0;
  return MBD_OK;
}
