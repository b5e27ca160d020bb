// mbd_plan.hip — the planner fast path (include/mbd_hip.h): one reverse-diffusion step split at its exchange point
// (mbd_plan_sample_rollout / mbd_plan_score_update), the noise ring of lazy plans, and the loops over steps
// (mbd_plan_reverse_once, mbd_plan_run, mbd_plan_eval, mbd_plan_peek).  mbd_planner.py:84-148,179-180.
#include "mbd_internal.h"

// ==================================================================================================
// planner
// ==================================================================================================
void host_schedule(float beta0, float betaT, int Nd, std::vector<float>& alphas,
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
  // logp0 [N] of the score kernel (and the cem selection's copy of the weights) live in LDS up to kLdsN candidates:
  // beyond the default 48 KB window the kernels' dynamic-LDS limit is raised; beyond kLdsN they use a plan-owned
  // global scratch instead — the candidate count is bounded by HBM, not by LDS
  if ((size_t)cfg->Nsample * sizeof(float) > 48 * 1024 && cfg->Nsample <= kLdsN) {
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
  HIP_TRY(hipMalloc(&p->d_wm_partial, sizeof(float) * (size_t)kWmG * p->HNu));
  if (N > kLdsN) HIP_TRY(hipMalloc(&p->d_lg, sizeof(float) * (size_t)N));
  // lazy candidates: the MBD update on a rigid-body env (the path-integral updates and car2d keep the materialised
  // Y0s: their kernels read it, and car2d's sampler is a few microseconds).  MBD_NO_LAZY=1: the materialised path (A/B)
  const bool no_lazy = env_flag("MBD_NO_LAZY");
  p->lazy = cfg->update_method == 0 && env->kind == ENV_MODEL && !no_lazy;
  if (p->lazy) {
    HIP_TRY(hipMalloc(&p->d_eps[0], sizeof(float) * (size_t)N * p->HNu));
    HIP_TRY(hipMalloc(&p->d_eps[1], sizeof(float) * (size_t)N * p->HNu));
    HIP_TRY(hipMalloc(&p->d_eps[2], sizeof(float) * (size_t)N * p->HNu));
    HIP_TRY(hipHostMalloc((void**)&p->h_progress, sizeof(int), hipHostMallocDefault));
    *p->h_progress = 0;
    HIP_TRY(hipMalloc(&p->d_ybar_keep, sizeof(float) * (size_t)p->HNu));
  }
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

static int ensure_aux(mbd_plan* p) {
  if (p->aux) return MBD_OK;
  HIP_TRY(hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_aux, hipEventDisableTiming));
  for (int b = 0; b < 3; ++b) HIP_TRY(hipEventCreateWithFlags(&p->ev_noise[b], hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_wm, hipEventDisableTiming));
  return MBD_OK;
}

static void launch_noise(mbd_plan* p, hipStream_t st, const uint32_t key[2], float* out) {
  const mbd_plan_config& c = p->cfg;
  const uint64_t total = (uint64_t)c.Nsample * p->HNu;
  const uint64_t items = c.prng_impl == MBD_PRNG_PARTITIONABLE ? total : (total + 1) / 2;
  uint64_t blocks = (items + 255) / 256;
  if (blocks > 65536) blocks = 65536;  // (grid-stride)
  hipLaunchKernelGGL(noise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, key[0], key[1], c.prng_impl, c.Nsample,
                     p->HNu, out);
}

// The normals of a diffusion step depend on its key only, not on the previous step's result.  This call DECLARES the
// key of the step AFTER the next mbd_plan_sample_rollout: that launch then also generates the declared step's normals —
// in spare workgroups of the rollout launch itself when the rollout leaves CUs idle (up to three quarters of the CUs:
// ~3000 humanoid candidates; no extra launch, no event), on the plan's second stream otherwise — so that the declared
// step starts without a sampler on its critical path.  A hint: a step whose normals were not prepared (no declaration,
// another key, a non-lazy plan) generates them on the spot; results are bit-identical either way.
extern "C" int mbd_plan_prefetch_noise(mbd_plan* p, const uint32_t key_next[2], void* stream_) {
  (void)stream_;
  if (!p || !key_next) return fail(MBD_ERR_INVALID, "NULL argument");
  const bool off = env_flag("MBD_NO_PREFETCH");
  if (off || !p->lazy) return MBD_OK;
  p->hint_key[0] = key_next[0];
  p->hint_key[1] = key_next[1];
  p->hint_valid = true;
  return MBD_OK;
}

// A plan's phases depend on each other through its buffers (the normals one step's launch prepares are read by the
// next; phase 2 reads what phase 1 wrote): stream order covers that while the caller stays on one stream; when a call
// arrives on another stream it is ordered behind the previous call with an event.
static int plan_enter(mbd_plan* p, hipStream_t s) {
  if (p->last_stream_set && p->last_stream != s) {
    if (!p->ev_xs) HIP_TRY(hipEventCreateWithFlags(&p->ev_xs, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->ev_xs, p->last_stream));
    HIP_TRY(hipStreamWaitEvent(s, p->ev_xs, 0));
  }
  p->last_stream = s;
  p->last_stream_set = true;
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
  {
    int rc = plan_enter(p, s);
    if (rc != MBD_OK) return rc;
  }
  const int N = c.Nsample, H = c.Hsample, HNu = p->HNu;
  LazyArgs lz;
  bool noise_on_aux = false;
  int nxt = 0;
  if (p->lazy) {
    // A1, lazy: every rank holds the normals of ALL N candidates (counter-based noise), so that phase 2 needs no
    // second collective and is bit-identical for every shard layout; the candidates themselves are formed at the
    // rollout's action fetch and inside the weighted mean.
    // the reader of a buffer the aux stream filled: no wait on the step's stream when the job has already finished
    auto join_aux = [&](int b) -> int {
      if (!p->eps_on_aux[b]) return MBD_OK;
      p->eps_on_aux[b] = false;
      if (hipEventQuery(p->ev_noise[b]) == hipSuccess) return MBD_OK;
      (void)hipGetLastError();  // (hipErrorNotReady is not an error here)
      HIP_TRY(hipStreamWaitEvent(s, p->ev_noise[b], 0));
      return MBD_OK;
    };
    int cur = -1;
    for (int b = 0; b < 3; ++b)
      if (p->eps_valid[b] && p->eps_key[b][0] == key_sample[0] && p->eps_key[b][1] == key_sample[1]) cur = b;
    if (cur >= 0) {  // prepared behind the previous rollout
      int rc = join_aux(cur);
      if (rc != MBD_OK) return rc;
    } else {  // not prepared: generate now, into the buffer behind the previous step's (stream order protects it)
      cur = (p->eps_cur + 1) % 3;
      int rc = join_aux(cur);  // (a stale prefetch may still be writing it)
      if (rc != MBD_OK) return rc;
      launch_noise(p, s, key_sample, p->d_eps[cur]);
      HIP_TRY(hipGetLastError());
      p->eps_key[cur][0] = key_sample[0];
      p->eps_key[cur][1] = key_sample[1];
      p->eps_valid[cur] = true;
    }
    p->eps_cur = cur;
    p->peek_ybar = d_Ybar_i;  // (the caller keeps it unchanged until phase 2 has run)
    p->sigma_last = p->sigmas[i];
    lz.ybar = d_Ybar_i;
    lz.sigma = p->sigmas[i];
    nxt = (cur + 1) % 3;
    // Preparing the next step's normals ahead only pays for a plan that has the device to itself (the caller says so:
    // mbd_plan_config.shares_device): beside other plans
    // (seed / temperature sweeps as concurrent plans, scripts/run_mbd.py) the noise workgroups would hold — through
    // the launch's LDS reservation — the CUs the other plans' rollouts need, and a second stream per plan runs the
    // process out of hardware queues (four N=1024 plans: 3100 plan-steps/s either way against 6200 with the normals
    // generated in front of each rollout, where the other plans' rollouts hide them anyway).
    const bool alone = c.shares_device == 0;
    const bool want = alone && p->hint_valid &&
                      !(p->hint_key[0] == key_sample[0] && p->hint_key[1] == key_sample[1]);
    p->hint_valid = false;
    if (want) {
      // d_eps[nxt] was last read two steps ago
      {
        int rc = join_aux(nxt);  // (a stale prefetch of another key: let it finish before it is overwritten)
        if (rc != MBD_OK) return rc;
      }
      p->eps_valid[nxt] = false;
      lz.nz_out = p->d_eps[nxt];
      lz.nz_key[0] = p->hint_key[0]; lz.nz_key[1] = p->hint_key[1];
      lz.nz_impl = c.prng_impl; lz.nz_N = N; lz.nz_HNu = HNu;
      // (launches that take the job into spare workgroups need no second stream and none of its events: a record
      // behind every weighted mean idles the queue ~5.5 us, 1 % of a step — profiles/r02_timeline.txt)
      noise_on_aux = !rollout_fuses_noise(e, c.shard_count);
    }
  } else {
    // A1, materialised (car2d, path-integral updates): every rank samples ALL N candidate sequences.  A sharded plan
    // samples its own rows first and the others' on a second stream, behind the rollout; mbd_plan_score_update joins
    // that stream before it reads them.
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
    const bool no_aux = env_flag("MBD_NO_AUX");
    const uint64_t total = (uint64_t)N * HNu;
    if (c.shard_count == N || no_aux || (long long)N < 5LL * c.shard_count) {
      // worth the two events only when the other ranks' rows dominate (tools/gpu_rank_emu.sh: 8 shards 0.774 ->
      // 0.762 ms per step, 2 shards 0.736 -> 0.742); MBD_NO_AUX=1 keeps everything on the caller's stream (A/B)
      sample(s, 0, total);
    } else {
      int rc = ensure_aux(p);
      if (rc != MBD_OK) return rc;
      HIP_TRY(hipEventRecord(p->ev_in, s));  // Ybar_i is final and the previous step is done with Y0s
      HIP_TRY(hipStreamWaitEvent(p->aux, p->ev_in, 0));
      sample(s, own0, own1 - own0);
      sample(p->aux, 0, own0);
      sample(p->aux, own1, total - own1);
      HIP_TRY(hipEventRecord(p->ev_aux, p->aux));
      p->aux_pending = true;
    }
    HIP_TRY(hipGetLastError());
  }
  bool aux_needs_mark = false;
  if (noise_on_aux) {
    // aux-stream generation must start after the last reader of d_eps[nxt] — the weighted mean behind rollout launch
    // number eps_read_seq[nxt] — and should not wait for THIS rollout.  That reader has finished once the NEXT rollout
    // launch of the plan has started (the progress word); a host that has not seen that yet puts a mark onto s in front
    // of this launch for the aux stream to wait on
    int rc = ensure_aux(p);
    if (rc != MBD_OK) return rc;
    const int r = p->eps_read_seq[nxt];
    p->in_step = (r == 0 || progress_read(p->h_progress) >= r + 1) && !p->kept_in_step;
    p->kept_in_step = false;
    if (r != 0 && progress_read(p->h_progress) < r + 1) {
      // the caller runs ahead of the device (an asynchronous loop): it is held here until the rollout before this one has
      // started — the queue still holds that rollout and its score, so the device does not wait for the host — rather
      // than paying a record on s and a wait on the aux stream per step (~20 us at N = 8192).  Bounded: a stream that is
      // itself waiting for something the caller has yet to do gets the mark after 5 ms.
      const auto w0 = std::chrono::steady_clock::now();
      while (progress_read(p->h_progress) < r + 1 && std::chrono::steady_clock::now() - w0 < std::chrono::milliseconds(5))
        std::this_thread::sleep_for(std::chrono::microseconds(10));
      aux_needs_mark = progress_read(p->h_progress) < r + 1;
    }
    if (aux_needs_mark) HIP_TRY(hipEventRecord(p->ev_wm, s));
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
  const float* d_cand = p->lazy ? p->d_eps[p->eps_cur] : p->d_Y0s;
  if (p->lazy) {
    lz.progress = p->h_progress;
    lz.progress_val = ++p->seq;
    p->eps_read_seq[p->eps_cur] = p->seq;
  }
  // A5: the demo log-densities of the local shard come out of the rollout itself where its instantiation accumulates them
  // (round 6: no [shard][H][K][3] round trip, no second launch); otherwise from the tracked positions, below
  const bool fused_lp = c.enable_demo && rollout_fuses_logpd(e, c.shard_count, H);
  int rc = launch_rollout(e, p->d_state0, d_cand + (size_t)c.shard_begin * HNu, c.shard_count, H, p->d_rewss,
                          d_rews_local, (c.enable_demo && !fused_lp) ? p->d_xpos : nullptr, nullptr, s, p->lazy ? &lz : nullptr,
                          nullptr, fused_lp ? d_logpd_local : nullptr);
  if (rc != MBD_OK) return rc;
  if (p->timing) HIP_TRY(hipEventRecord(ev1, s));
  if (lz.nz_out) {
    if (!lz.nz_fused) {  // the rollout fills the chip: the next step's normals on the second stream, beside it
      rc = ensure_aux(p);
      if (rc != MBD_OK) return rc;
      // (a caller in step with the device — it reads every step's mean reward before it dispatches the next — launches
      // this while the previous step's weighted mean is still running: the job waits for the mark behind that kernel,
      // which cost nothing there (the queue was about to drain), instead of competing with it for the memory system)
      if (aux_needs_mark || (p->in_step && p->wm_mark_valid)) HIP_TRY(hipStreamWaitEvent(p->aux, p->ev_wm, 0));
      launch_noise(p, p->aux, lz.nz_key, lz.nz_out);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(p->ev_noise[nxt], p->aux));
      p->eps_on_aux[nxt] = true;
    }
    p->eps_key[nxt][0] = lz.nz_key[0];
    p->eps_key[nxt][1] = lz.nz_key[1];
    p->eps_valid[nxt] = true;
  }
  if (c.enable_demo && !fused_lp) {
    rc = launch_logpd(e, p->d_xpos, c.shard_count, H, d_logpd_local, s);
    if (rc != MBD_OK) return rc;
  }
  return MBD_OK;
}

extern "C" int mbd_plan_score_update(mbd_plan* p, int i, const uint32_t key_sample[2], const float* d_Ybar_i,
                                     const float* d_rews_all, const float* d_logpd_all, float* d_Ybar_im1,
                                     float* d_rew_mean, void* stream_) {
  (void)key_sample;  // the candidates (or their normals) of all N are already resident from phase 1 of this step
  if (!p || !d_Ybar_i || !d_rews_all || !d_Ybar_im1 || !d_rew_mean) return fail(MBD_ERR_INVALID, "NULL argument");
  const mbd_plan_config& c = p->cfg;
  if (i < 1 || i >= c.Ndiffuse) return fail(MBD_ERR_INVALID, "diffusion index %d outside [1,%d)", i, c.Ndiffuse);
  if (c.enable_demo && !d_logpd_all) return fail(MBD_ERR_INVALID, "enable_demo needs d_logpd_all");
  HIP_TRY(hipSetDevice(p->env->device));
  hipStream_t s = (hipStream_t)stream_;
  {
    int rc = plan_enter(p, s);
    if (rc != MBD_OK) return rc;
  }
  const int N = c.Nsample, HNu = p->HNu;
  if (p->aux_pending) {  // the other ranks' rows of Y0s (sampled behind the rollout)
    HIP_TRY(hipStreamWaitEvent(s, p->ev_aux, 0));
    p->aux_pending = false;
  }
  const size_t lds_n = N > kLdsN ? 0 : sizeof(float) * (size_t)N;
  // While the N weights fit the default 48 KB LDS window (12 288 candidates) score and weighted mean are ONE launch
  // (score_wmean_kernel: every workgroup re-derives the weights — same bits); beyond, score_kernel and the row-major
  // two-kernel weighted mean (same bits again: see wmean_partial_kernel).  One launch beats three even where the
  // row-major reads are faster (N = 4096: +0.8 % of a step, N = 8192: +0.3 %).  MBD_WMEAN_SPLIT=0/1,
  // MBD_NO_FUSED_SCORE=1 force the variants (A/B, tests).
  const int split_env = lever("MBD_WMEAN_SPLIT");
  const bool split = split_env >= 0 ? split_env != 0 : (size_t)N * sizeof(float) > 48 * 1024;
  const bool no_fused_score = env_flag("MBD_NO_FUSED_SCORE");
  const bool fused_score = !split && c.update_method != 3 && (size_t)N * sizeof(float) <= 48 * 1024 && !no_fused_score;
  if (!fused_score) {
    hipLaunchKernelGGL(score_kernel, dim3(1), dim3(kScoreThreads), lds_n, s, d_rews_all,
                       c.enable_demo ? d_logpd_all : nullptr, N, p->env->rew_xref, c.temp_sample,
                       c.update_method == 0 ? 1 : 0, p->d_weights, d_rew_mean, p->d_lg, PiBatch{});
    HIP_TRY(hipGetLastError());
  }
  const dim3 ge((HNu + 63) / 64), b64(64);
  const float* d_cand = p->lazy ? p->d_eps[p->eps_cur] : p->d_Y0s;
  const int lazy = p->lazy ? 1 : 0;
  const float sigma_i = p->sigmas[i];
  if (c.update_method == 3) {  // cem_update (path_integral.py:48-52)
    const int K = N < 10 ? N : 10;
    hipLaunchKernelGGL(cem_select_kernel, dim3(1), b64, lds_n, s, p->d_weights, N, K, p->d_idx, p->d_lg, PiBatch{});
    hipLaunchKernelGGL(cem_mean_kernel, ge, b64, 0, s, p->d_idx, K, p->d_Y0s, HNu, d_Ybar_im1, PiBatch{});
  } else {  // MBD (:128-133), mppi (:33-36), cma-es (:39-45)
    const int lit = c.update_method == 0 ? c.literal_score : 0;
    if (fused_score) {
      // XCD pinning (mbd_step_kernels.h pinned_tile): the T tiles on the fewest XCDs X in {1, 2, 4, 8} that give every tile
      // a CU of its own (32 per XCD) and keep an XCD's share of the candidates' rows within its 4 MB L2; the launch is
      // 8 ceil(T / X) workgroups long, those of the other XCDs leave at once.  MBD_WMEAN_XCDS = 1 / 2 / 4 / 8 forces X.
      // outputs per thread: one (two measured slower at every size and no lighter: mbd_step_kernels.h).  MBD_WMEAN_V1 = 2 forces two.
      const int V = lever("MBD_WMEAN_V1") == 2 ? 2 : 1;
      const int T = (HNu + kWmE * V - 1) / (kWmE * V);
      int X = 1;
      while (X < 8 && ((T + X - 1) / X > 32 || (size_t)N * HNu * sizeof(float) / X > (size_t)4 << 20)) X *= 2;
      if (!device_has_eight_xcds(p->env)) X = 8;  // (a partition or another part: the plain launch, no empty workgroups)
      const int x_env = lever("MBD_WMEAN_XCDS");
      if (x_env == 1 || x_env == 2 || x_env == 4 || x_env == 8) X = x_env;
      auto kern = V == 2 ? score_wmean_kernel<2> : score_wmean_kernel<1>;
      hipLaunchKernelGGL(kern, dim3(8 * ((T + X - 1) / X)), dim3(kWmE * kWmG), sizeof(float) * (size_t)N,
                         s, d_rews_all, c.enable_demo ? d_logpd_all : nullptr, N, p->env->rew_xref, c.temp_sample,
                         c.update_method == 0 ? 1 : 0, p->d_weights, d_rew_mean, d_cand, HNu, d_Ybar_i, p->alphas[i],
                         p->alphas_bar[i], p->alphas_bar[i - 1], lit, d_Ybar_im1, lazy, sigma_i, p->d_ybar_keep, T, X);
    } else if (split) {
      hipLaunchKernelGGL(wmean_partial_kernel, dim3((HNu + kWmT - 1) / kWmT, kWmG), dim3(kWmT), 0, s, p->d_weights,
                         d_cand, N, HNu, p->d_wm_partial, lazy, sigma_i, d_Ybar_i);
      hipLaunchKernelGGL(wmean_finish_kernel, dim3((HNu + 63) / 64), dim3(64), 0, s, p->d_wm_partial, HNu, d_Ybar_i,
                         p->alphas[i], p->alphas_bar[i], p->alphas_bar[i - 1], lit, d_Ybar_im1, p->d_ybar_keep);
    } else {
      hipLaunchKernelGGL(wmean_kernel, dim3((HNu + kWmE - 1) / kWmE), dim3(kWmE * kWmG), sizeof(float) * (size_t)N, s,
                         p->d_weights, d_cand, N, HNu, d_Ybar_i, p->alphas[i], p->alphas_bar[i],
                         p->alphas_bar[i - 1], lit, d_Ybar_im1, lazy, sigma_i, p->d_ybar_keep);
    }
    if (c.update_method == 2) {
      hipLaunchKernelGGL(cma_spread_kernel, ge, b64, 0, s, p->d_weights, p->d_Y0s, N, HNu, d_Ybar_i, p->d_spread, PiBatch{});
      hipLaunchKernelGGL(cma_sigma_kernel, dim3(1), b64, 0, s, p->d_spread, HNu, p->d_sigma, PiBatch{});
    }
  }
  HIP_TRY(hipGetLastError());
  if (p->lazy) {
    p->peek_ybar = p->d_ybar_keep;
    p->wm_mark_valid = false;
    if (p->aux && p->in_step && p->ev_wm) {  // (see sample_rollout: only where the record is free)
      HIP_TRY(hipEventRecord(p->ev_wm, s));
      p->wm_mark_valid = true;
    }
  }
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

// Loops that enqueue step after step (mbd_plan_run): a plan whose next step's normals are generated on the second stream
// stays ONE step behind the device — it enqueues step q once the rollout of step q-1 has started (the queue still holds
// that rollout and its score: the device never waits for the host) — so that sample_rollout finds the progress word where
// it needs it and the step's stream carries no event (mbd_plan: the ring of three buffers).
static int plan_keep_in_step(mbd_plan* p) {
  if (!p->lazy || !p->h_progress || p->seq == 0 || p->cfg.shares_device != 0) return MBD_OK;
  if (rollout_fuses_noise(p->env, p->cfg.shard_count)) return MBD_OK;
  p->kept_in_step = true;
  const auto w0 = std::chrono::steady_clock::now();
  while (progress_read(p->h_progress) < p->seq) {
    // a stream that is legitimately slow (a shared or time-sliced GPU, a profiler, a system pause) is not an error: the
    // loop stops keeping step and mbd_plan_sample_rollout orders the two streams with an event instead (its own bounded
    // wait, then a mark on the step's stream for the aux stream)
    if (std::chrono::steady_clock::now() - w0 > std::chrono::milliseconds(kInStepWaitMs)) {
      p->kept_in_step = false;
      break;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  return MBD_OK;
}

static int reverse_once_impl(mbd_plan* p, int i, uint32_t key_inout[2], const float* d_Ybar_in, float* d_Ybar_out,
                             float* d_rew_mean, hipStream_t s) {
  if (p->cfg.shard_count != p->cfg.Nsample)
    return fail(MBD_ERR_STATE, "reverse_once on a sharded plan: use sample_rollout + all-gather + score_update");
  uint32_t keys[4];
  host_split(key_inout, 2, p->cfg.prng_impl, keys);  // rng, Y0s_rng = split(rng)  (mbd_planner.py:103)
  const uint32_t ks[2] = {keys[2], keys[3]};
  int rc;
  if (i > 1) {  // the next step's normals beside this rollout: its key is the next split of the advanced rng
    uint32_t nk[4];
    const uint32_t adv[2] = {keys[0], keys[1]};
    host_split(adv, 2, p->cfg.prng_impl, nk);
    const uint32_t next_ks[2] = {nk[2], nk[3]};
    rc = mbd_plan_prefetch_noise(p, next_ks, s);
    if (rc != MBD_OK) return rc;
  }
  rc = mbd_plan_sample_rollout(p, i, ks, d_Ybar_in, p->d_rews, p->cfg.enable_demo ? p->d_lp : nullptr, s);
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
    int rc = plan_keep_in_step(p);
    if (rc != MBD_OK) return rc;
    rc = reverse_once_impl(p, i, rng, cur, nxt, p->d_rewmeans + (Nd - 1 - i), s);
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
  if (Y0s_out && p->lazy) {  // lazy plans never formed Y0s: do it now from the last step's normals, Ybar_i and sigma_i
    if (!p->peek_ybar) return fail(MBD_ERR_STATE, "peek: no diffusion step to show yet");
    const uint64_t total = (uint64_t)c.Nsample * p->HNu;
    hipLaunchKernelGGL(shift_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream,
                       p->d_eps[p->eps_cur], p->HNu, 0ull, (unsigned long long)total, p->sigma_last,
                       (const float*)nullptr, p->peek_ybar, p->d_Y0s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
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
