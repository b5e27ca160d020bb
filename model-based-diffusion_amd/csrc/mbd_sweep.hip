// mbd_sweep.hip — sweeps (include/mbd_hip.h): several plans of one env advanced in lockstep, one rollout launch over all
// their candidates per step (mbd/scripts/run_mbd.py:17-64); MBD plans and the path-integral baselines.
#include "mbd_internal.h"

namespace {
// outputs per thread of the sweeps' score + weighted mean launch (score_wmean_batch_kernel<V>): 2 — a workgroup owns 32 outputs,
// 128 bytes of every candidate row, P x 27 workgroups for the humanoid.  Measured on sweep8 (profiles/r05_score_ab.txt; the
// launch's algorithmic bytes are 28.0 MB): V = 1 15.5 us / 48.2 MB, V = 2 11.3 us / 28.2 MB, V = 4 22.0 us / 28.2 MB.
// MBD_WMEAN_V = 1 / 2 / 4 forces it (A/B, tests; same bits whatever it is)
int wmean_batch_v() {
  const int v = lever("MBD_WMEAN_V");
  return (v == 1 || v == 2 || v == 4) ? v : 2;
}
template <typename... Args>
void launch_score_wmean_batch(int V, int HNu, int P, size_t lds, hipStream_t s, Args... args) {
  using namespace mbd;
  const dim3 block(kWmE * kWmG);
  auto tiles = [&](int v) { return dim3((unsigned)((HNu + kWmE * v - 1) / (kWmE * v)), (unsigned)P); };
  if (V == 1) hipLaunchKernelGGL(score_wmean_batch_kernel<1>, tiles(1), block, lds, s, args...);
  else if (V == 2) hipLaunchKernelGGL(score_wmean_batch_kernel<2>, tiles(2), block, lds, s, args...);
  else hipLaunchKernelGGL(score_wmean_batch_kernel<4>, tiles(4), block, lds, s, args...);
}
}  // namespace

// ---- sweeps: P plans of one env in lockstep (mbd/scripts/run_mbd.py:17-64) ---------------------------------------
struct mbd_sweep {
  mbd_env* env = nullptr;
  mbd_plan_config cfg;
  int P = 0, HNu = 0;
  std::vector<float> temps, alphas, alphas_bar, sigmas;
  hipStream_t stream = nullptr, aux = nullptr;
  // The normals of a step live in a ring of THREE buffers, so that the stream of the steps carries no event at all: the
  // buffer step k+1's normals go into was last read by step k-2's weighted mean, which has finished once the rollout of
  // step k-1 has STARTED — the host learns that from the progress word the rollout launches store into (pinned memory),
  // and the loop stays that close behind the device.  (Two buffers + events: a record behind every weighted mean and a
  // wait in front of every rollout idled the queue 17 us per step, profiles/r03_ring_ab.txt.)
  hipEvent_t ev_ready[3] = {nullptr, nullptr, nullptr};  // eps[b] holds the normals of its step (recorded on aux)
  hipEvent_t ev_order = nullptr;  // the event-ordered fallback of a loop whose stream is slower than kInStepWaitMs
  int event_fallbacks = 0;
  int* h_progress = nullptr;
  float *d_state0 = nullptr, *d_eps[3] = {nullptr, nullptr, nullptr}, *d_rews = nullptr, *d_rewss = nullptr, *d_lp = nullptr;
  float *d_xpos = nullptr, *d_weights = nullptr, *d_zero = nullptr, *d_mu = nullptr, *d_rewmeans = nullptr;
  float *d_temps = nullptr, *d_final = nullptr, *d_final_rew = nullptr;
  // path-integral sweeps (update_method != 0; path_integral.py:111-127): materialised candidates, the carried sigma of
  // every plan, cma-es' spread, cem's selection
  float *d_Y0s = nullptr, *d_sigma = nullptr, *d_spread = nullptr;
  int* d_idx = nullptr;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  mbd_sweep() = default;
  mbd_sweep(const mbd_sweep&) = delete;
  mbd_sweep& operator=(const mbd_sweep&) = delete;
  ~mbd_sweep() {
    if (env) (void)hipSetDevice(env->device);
    for (auto& ev : events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipFree(d_state0); (void)hipFree(d_eps[0]); (void)hipFree(d_eps[1]); (void)hipFree(d_eps[2]); (void)hipFree(d_rews);
    (void)hipFree(d_rewss);
    if (h_progress) (void)hipHostFree(h_progress);
    (void)hipFree(d_lp); (void)hipFree(d_xpos); (void)hipFree(d_weights); (void)hipFree(d_zero); (void)hipFree(d_mu);
    (void)hipFree(d_rewmeans); (void)hipFree(d_temps); (void)hipFree(d_final); (void)hipFree(d_final_rew);
    (void)hipFree(d_Y0s); (void)hipFree(d_sigma); (void)hipFree(d_spread); (void)hipFree(d_idx);
    for (int b = 0; b < 3; ++b)
      if (ev_ready[b]) (void)hipEventDestroy(ev_ready[b]);
    if (ev_order) (void)hipEventDestroy(ev_order);
    if (stream) (void)hipStreamDestroy(stream);
    if (aux) (void)hipStreamDestroy(aux);
  }
};

extern "C" int mbd_sweep_create(mbd_env* env, const mbd_plan_config* cfg, int n_plans, const float* temps, mbd_sweep** out) {
  if (!env || !cfg || !out) return fail(MBD_ERR_INVALID, "NULL argument");
  if (device_count_quiet() < 1) return fail(MBD_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback");
  if (n_plans < 1 || n_plans > MBD_SWEEP_MAX_PLANS) return fail(MBD_ERR_INVALID, "n_plans=%d outside [1,%d]", n_plans, MBD_SWEEP_MAX_PLANS);
  if (cfg->Nsample < 1 || cfg->Hsample < 1 || cfg->Ndiffuse < 2) return fail(MBD_ERR_INVALID, "bad plan sizes");
  if (cfg->update_method < 0 || cfg->update_method > 3) return fail(MBD_ERR_INVALID, "update_method=%d", cfg->update_method);
  if (env->kind != ENV_MODEL) return fail(MBD_ERR_UNSUPPORTED, "sweeps batch plans on rigid-body envs; run car2d plans as plans");
  if (cfg->update_method > 0 && cfg->enable_demo) return fail(MBD_ERR_INVALID, "path-integral plans do not use demos");
  if (cfg->shard_begin != 0 || cfg->shard_count != cfg->Nsample) return fail(MBD_ERR_INVALID, "sweeps are not sharded");
  if ((size_t)cfg->Nsample * sizeof(float) > 48 * 1024)
    return fail(MBD_ERR_UNSUPPORTED, "plans of more than 12288 candidates fill the chip on their own: run them as plans");
  if (cfg->enable_demo && (!env->has_xref || cfg->Hsample != 50)) return fail(MBD_ERR_INVALID, "enable_demo: the env has no demo / H != 50");
  HIP_TRY(hipSetDevice(env->device));
  std::unique_ptr<mbd_sweep> guard(new mbd_sweep());
  mbd_sweep* w = guard.get();
  w->env = env; w->cfg = *cfg; w->P = n_plans;
  const int N = cfg->Nsample, H = cfg->Hsample, Nu = env->action_size(), Nd = cfg->Ndiffuse, P = n_plans;
  w->HNu = H * Nu;
  w->temps.assign(P, cfg->temp_sample);
  if (temps) for (int k = 0; k < P; ++k) w->temps[k] = temps[k];
  host_schedule(cfg->beta0, cfg->betaT, Nd, w->alphas, w->alphas_bar, w->sigmas);
  HIP_TRY(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&w->aux, hipStreamNonBlocking));
  for (int b = 0; b < 3; ++b) HIP_TRY(hipEventCreateWithFlags(&w->ev_ready[b], hipEventDisableTiming));
  HIP_TRY(hipHostMalloc((void**)&w->h_progress, sizeof(int), hipHostMallocDefault));
  *w->h_progress = 0;
  const int K = env->model.n_track > 0 ? env->model.n_track : 1;
  const size_t PN = (size_t)P * N;
  HIP_TRY(hipMalloc(&w->d_state0, sizeof(float) * (size_t)P * env->state_size()));
  if (cfg->update_method == 0) {  // MBD plans: lazy candidates, the normals in a ring of three buffers
    HIP_TRY(hipMalloc(&w->d_eps[0], sizeof(float) * PN * w->HNu));
    HIP_TRY(hipMalloc(&w->d_eps[1], sizeof(float) * PN * w->HNu));
    HIP_TRY(hipMalloc(&w->d_eps[2], sizeof(float) * PN * w->HNu));
  } else {  // path-integral plans: the candidates themselves (their kernels read them), sigma per plan on the device
    HIP_TRY(hipMalloc(&w->d_Y0s, sizeof(float) * PN * w->HNu));
    HIP_TRY(hipMalloc(&w->d_sigma, sizeof(float) * P));
    HIP_TRY(hipMalloc(&w->d_spread, sizeof(float) * (size_t)P * w->HNu));
    HIP_TRY(hipMalloc(&w->d_idx, sizeof(int) * (size_t)P * 16));
  }
  HIP_TRY(hipMalloc(&w->d_rews, sizeof(float) * PN));
  HIP_TRY(hipMalloc(&w->d_rewss, sizeof(float) * PN * H));
  HIP_TRY(hipMalloc(&w->d_lp, sizeof(float) * PN));
  if (cfg->enable_demo) HIP_TRY(hipMalloc(&w->d_xpos, sizeof(float) * PN * H * K * 3));
  HIP_TRY(hipMalloc(&w->d_weights, sizeof(float) * PN));
  HIP_TRY(hipMalloc(&w->d_zero, sizeof(float) * (size_t)P * w->HNu));
  HIP_TRY(hipMemset(w->d_zero, 0, sizeof(float) * (size_t)P * w->HNu));
  HIP_TRY(hipMalloc(&w->d_mu, sizeof(float) * (size_t)P * (Nd - 1) * w->HNu));
  HIP_TRY(hipMalloc(&w->d_rewmeans, sizeof(float) * (size_t)P * (Nd - 1)));
  HIP_TRY(hipMalloc(&w->d_temps, sizeof(float) * P));
  HIP_TRY(hipMemcpy(w->d_temps, w->temps.data(), sizeof(float) * P, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&w->d_final, sizeof(float) * (size_t)P * w->HNu));
  HIP_TRY(hipMalloc(&w->d_final_rew, sizeof(float) * (size_t)P));
  *out = guard.release();
  return MBD_OK;
}

extern "C" int mbd_sweep_destroy(mbd_sweep* w) {
  delete w;
  return MBD_OK;
}

extern "C" int mbd_sweep_set_state0(mbd_sweep* w, int k, const float* state0) {
  if (!w || !state0) return fail(MBD_ERR_INVALID, "NULL argument");
  if (k < 0 || k >= w->P) return fail(MBD_ERR_INVALID, "plan %d outside [0,%d)", k, w->P);
  HIP_TRY(hipSetDevice(w->env->device));
  const size_t S = w->env->state_size();
  HIP_TRY(hipMemcpy(w->d_state0 + (size_t)k * S, state0, sizeof(float) * S, hipMemcpyHostToDevice));
  return MBD_OK;
}

extern "C" int mbd_sweep_kernel_time(mbd_sweep* w, int enable, float* avg_ms_out, int* count_out) {
  if (!w) return fail(MBD_ERR_INVALID, "sweep is NULL");
  HIP_TRY(hipSetDevice(w->env->device));
  HIP_TRY(hipDeviceSynchronize());
  double tot = 0.0;
  for (size_t k = 0; k < w->events_used; ++k) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, w->events[k].first, w->events[k].second));
    tot += ms;
  }
  if (avg_ms_out) *avg_ms_out = w->events_used ? (float)(tot / (double)w->events_used) : 0.0f;
  if (count_out) *count_out = (int)w->events_used;
  w->events_used = 0;
  w->timing = enable != 0;
  return MBD_OK;
}

// A sweep of path-integral plans (mbd/scripts/run_mbd.py:22-26,46-50 over path_integral.py:111-127): per refinement step
// ONE sampling launch (every plan's key, carried sigma and mean), ONE rollout launch over the P * N materialised
// candidates, and the update rule's kernels with blockIdx.y = plan — mppi: the fused score + weighted mean; cma-es: plus
// spread and sigma; cem: score, selection, mean of the K best.  Same kernels, same order, same bits as mbd_plan_run.
static int sweep_run_path_integral(mbd_sweep* w, const uint32_t* keys, float* mu_0ts_out, float* rew_means_out,
                                   float* rew_final_out, double* loop_seconds_out) {
  mbd_env* e = w->env;
  const mbd_plan_config& c = w->cfg;
  const int P = w->P, N = c.Nsample, H = c.Hsample, Nd = c.Ndiffuse, HNu = w->HNu, S = e->state_size();
  hipStream_t s = w->stream;
  std::vector<uint32_t> rng(keys, keys + 2 * (size_t)P);
  const uint64_t per_plan = (uint64_t)N * HNu;
  const uint64_t items = c.prng_impl == MBD_PRNG_PARTITIONABLE ? per_plan : (per_plan + 1) / 2;
  uint64_t nblocks = (items + 255) / 256;
  if (nblocks > 4096) nblocks = 4096;  // (grid-stride)
  {
    std::vector<float> ones((size_t)P, 1.0f);  // sigma = 1.0 (path_integral.py:131)
    HIP_TRY(hipMemcpy(w->d_sigma, ones.data(), sizeof(float) * P, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  const dim3 b64(64);
  PiBatch pb;
  pb.rews = N; pb.weights = N; pb.mean = Nd - 1; pb.cand = (long long)per_plan; pb.spread = HNu; pb.sigma = 1; pb.idx = 16;
  pb.out = (long long)(Nd - 1) * HNu; pb.temps = w->d_temps;
  for (int i = Nd - 1, step = 0; i >= 1; --i, ++step) {
    SweepKeys sk;
    for (int k = 0; k < P; ++k) {  // rng, Y0s_rng = split(rng) (path_integral.py:114)
      uint32_t ks[4];
      host_split(&rng[2 * k], 2, c.prng_impl, ks);
      rng[2 * k] = ks[0]; rng[2 * k + 1] = ks[1];
      sk.k[k][0] = ks[2]; sk.k[k][1] = ks[3];
    }
    const float* mu_in = step == 0 ? w->d_zero : w->d_mu + (size_t)(step - 1) * HNu;
    const long long mu_stride = step == 0 ? HNu : (long long)(Nd - 1) * HNu;
    float* mu_out = w->d_mu + (size_t)step * HNu;
    hipLaunchKernelGGL(sample_batch_kernel, dim3((unsigned)nblocks, (unsigned)P), dim3(256), 0, s, sk, c.prng_impl, N, HNu,
                       (const float*)w->d_sigma, mu_in, mu_stride, w->d_Y0s);
    HIP_TRY(hipGetLastError());
    hipEvent_t ev1 = nullptr;
    if (w->timing) {
      if (w->events_used == w->events.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        w->events.emplace_back(a, b);
      }
      ev1 = w->events[w->events_used].second;
      HIP_TRY(hipEventRecord(w->events[w->events_used].first, s));
      w->events_used++;
    }
    const int sw[3] = {N, S, 0};
    int rc = launch_rollout(e, w->d_state0, w->d_Y0s, P * N, H, w->d_rewss, w->d_rews, nullptr, nullptr, s, nullptr, sw);
    if (rc != MBD_OK) return rc;
    if (ev1) HIP_TRY(hipEventRecord(ev1, s));
    pb.mu = mu_stride;
    if (c.update_method == 3) {  // cem_update (path_integral.py:48-52)
      const int K = N < 10 ? N : 10;
      hipLaunchKernelGGL(score_kernel, dim3(1, (unsigned)P), dim3(kScoreThreads), sizeof(float) * (size_t)N, s, (const float*)w->d_rews,
                         (const float*)nullptr, N, e->rew_xref, c.temp_sample, 0, w->d_weights, w->d_rewmeans + step,
                         (float*)nullptr, pb);
      hipLaunchKernelGGL(cem_select_kernel, dim3(1, (unsigned)P), b64, sizeof(float) * (size_t)N, s, (const float*)w->d_weights, N, K,
                         w->d_idx, (float*)nullptr, pb);
      hipLaunchKernelGGL(cem_mean_kernel, dim3((HNu + 63) / 64, (unsigned)P), b64, 0, s, (const int*)w->d_idx, K,
                         (const float*)w->d_Y0s, HNu, mu_out, pb);
    } else {  // mppi (:33-36), cma-es (:39-45): softmax weights and the weighted mean in one launch
      ScoreBatch sb;
      sb.rews = N; sb.lp = N; sb.weights = N; sb.mean = Nd - 1; sb.cand = (long long)per_plan;
      sb.ybar_in = mu_stride; sb.ybar_out = (long long)(Nd - 1) * HNu; sb.keep = 0; sb.temps = w->d_temps;
      launch_score_wmean_batch(wmean_batch_v(), HNu, P, sizeof(float) * (size_t)N, s, w->d_rews, (const float*)nullptr, N, e->rew_xref,
                               c.temp_sample, 0, w->d_weights, w->d_rewmeans + step, (const float*)w->d_Y0s, HNu, mu_in, 1.0f, 1.0f,
                               1.0f, 0, mu_out, 0, 0.0f, (float*)nullptr, sb);
      if (c.update_method == 2) {
        hipLaunchKernelGGL(cma_spread_kernel, dim3((HNu + 63) / 64, (unsigned)P), b64, 0, s, (const float*)w->d_weights,
                           (const float*)w->d_Y0s, N, HNu, mu_in, w->d_spread, pb);
        hipLaunchKernelGGL(cma_sigma_kernel, dim3(1, (unsigned)P), b64, 0, s, (const float*)w->d_spread, HNu, w->d_sigma, pb);
      }
    }
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(s));
  auto t1 = std::chrono::steady_clock::now();
  if (loop_seconds_out) *loop_seconds_out = std::chrono::duration<double>(t1 - t0).count();
  const size_t mu_n = (size_t)P * (Nd - 1) * HNu;
  if (mu_0ts_out) HIP_TRY(hipMemcpy(mu_0ts_out, w->d_mu, sizeof(float) * mu_n, hipMemcpyDeviceToHost));
  if (rew_means_out) HIP_TRY(hipMemcpy(rew_means_out, w->d_rewmeans, sizeof(float) * (size_t)P * (Nd - 1), hipMemcpyDeviceToHost));
  if (rew_final_out) {  // eval_us(state_init, mu_0).mean() of every plan (path_integral.py:146): one launch of P candidates
    HIP_TRY(hipMemcpy2DAsync(w->d_final, sizeof(float) * HNu, w->d_mu + (size_t)(Nd - 2) * HNu,
                             sizeof(float) * (size_t)(Nd - 1) * HNu, sizeof(float) * HNu, P, hipMemcpyDeviceToDevice, s));
    const int sw[3] = {1, S, 0};
    int rc = launch_rollout(e, w->d_state0, w->d_final, P, H, nullptr, w->d_final_rew, nullptr, nullptr, s, nullptr, sw);
    if (rc != MBD_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(rew_final_out, w->d_final_rew, sizeof(float) * P, hipMemcpyDeviceToHost));
  }
  return MBD_OK;
}

extern "C" int mbd_sweep_get_sigmas(mbd_sweep* w, float* sigmas_out) {
  if (!w || !sigmas_out) return fail(MBD_ERR_INVALID, "NULL argument");
  if (!w->d_sigma) return fail(MBD_ERR_STATE, "not a path-integral sweep (update_method == 0)");
  HIP_TRY(hipSetDevice(w->env->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(sigmas_out, w->d_sigma, sizeof(float) * w->P, hipMemcpyDeviceToHost));
  return MBD_OK;
}

extern "C" int mbd_sweep_run(mbd_sweep* w, const uint32_t* keys, float* mu_0ts_out, float* rew_means_out,
                             float* rew_final_out, double* loop_seconds_out) {
  if (!w || !keys) return fail(MBD_ERR_INVALID, "NULL argument");
  mbd_env* e = w->env;
  HIP_TRY(hipSetDevice(e->device));
  const mbd_plan_config& c = w->cfg;
  const int P = w->P, N = c.Nsample, H = c.Hsample, Nd = c.Ndiffuse, HNu = w->HNu, S = e->state_size();
  hipStream_t s = w->stream;
  // per plan: rng, Y0s_rng = split(rng) per step (mbd_planner.py:103) — the whole key chain is host arithmetic
  std::vector<uint32_t> rng(keys, keys + 2 * (size_t)P);
  auto step_keys = [&](SweepKeys& out) {
    for (int k = 0; k < P; ++k) {
      uint32_t ks[4];
      host_split(&rng[2 * k], 2, c.prng_impl, ks);
      rng[2 * k] = ks[0]; rng[2 * k + 1] = ks[1];
      out.k[k][0] = ks[2]; out.k[k][1] = ks[3];
    }
  };
  // the normals of a step depend on its keys only: they are generated on the second stream while the previous step's
  // rollout runs (a ring of three buffers, see mbd_sweep), like a single large plan's
  const uint64_t per_plan = (uint64_t)N * HNu;
  const uint64_t items = c.prng_impl == MBD_PRNG_PARTITIONABLE ? per_plan : (per_plan + 1) / 2;
  uint64_t nblocks = (items + 255) / 256;
  if (nblocks > 4096) nblocks = 4096;  // (grid-stride)
  auto launch_noise_step = [&](int buf, hipStream_t st) {
    SweepKeys sk;
    step_keys(sk);
    hipLaunchKernelGGL(noise_batch_kernel, dim3((unsigned)nblocks, (unsigned)P), dim3(256), 0, st, sk, c.prng_impl, N, HNu,
                       w->d_eps[buf]);
  };
  if (c.update_method != 0) return sweep_run_path_integral(w, keys, mu_0ts_out, rew_means_out, rew_final_out, loop_seconds_out);
  const int sweep_args[3] = {N, S, HNu};
  ScoreBatch sb;
  sb.rews = N; sb.lp = N; sb.weights = N; sb.mean = Nd - 1; sb.cand = (long long)per_plan;
  sb.ybar_out = (long long)(Nd - 1) * HNu; sb.keep = 0; sb.temps = w->d_temps;
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipStreamSynchronize(w->aux));
  __atomic_store_n(w->h_progress, 0, __ATOMIC_RELEASE);
  auto t0 = std::chrono::steady_clock::now();
  launch_noise_step(0, s);  // step Nd-1
  HIP_TRY(hipGetLastError());
  for (int i = Nd - 1, step = 0; i >= 1; --i, ++step) {
    const int cur = step % 3, nxt = (step + 1) % 3;
    if (step > 0 && hipEventQuery(w->ev_ready[cur]) != hipSuccess) {  // (generated a whole step ago: ready in practice)
      (void)hipGetLastError();
      HIP_TRY(hipStreamWaitEvent(s, w->ev_ready[cur], 0));
    }
    if (i > 1) {  // the next step's normals beside this rollout
      // eps[nxt] was last read by the weighted mean of step - 2, finished once the rollout of step - 1 (which stores
      // `step` into the progress word) has started: the loop waits for that — one step behind the device, whose queue
      // still holds that rollout and its score — instead of ordering the two streams with events
      if (step >= 2) {
        const auto w0 = std::chrono::steady_clock::now();
        while (progress_read(w->h_progress) < step) {
          if (std::chrono::steady_clock::now() - w0 > std::chrono::milliseconds(kInStepWaitMs)) {
            // a legitimately slow stream (shared / time-sliced GPU, profiler, system pause): order the streams with an
            // event instead — everything enqueued on s so far, the reader of eps[nxt] included, precedes the generation
            if (!w->ev_order) HIP_TRY(hipEventCreateWithFlags(&w->ev_order, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(w->ev_order, s));
            HIP_TRY(hipStreamWaitEvent(w->aux, w->ev_order, 0));
            w->event_fallbacks++;
            break;
          }
          std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
      }
      launch_noise_step(nxt, w->aux);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(w->ev_ready[nxt], w->aux));
    }
    const float* ybar_in = step == 0 ? w->d_zero : w->d_mu + (size_t)(step - 1) * HNu;
    const long long ybar_in_stride = step == 0 ? HNu : (long long)(Nd - 1) * HNu;
    LazyArgs lz;
    lz.ybar = ybar_in;
    lz.sigma = w->sigmas[i];
    lz.progress = w->h_progress;
    lz.progress_val = step + 1;
    const int sw[3] = {sweep_args[0], sweep_args[1], (int)ybar_in_stride};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (w->timing) {
      if (w->events_used == w->events.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        w->events.emplace_back(a, b);
      }
      ev0 = w->events[w->events_used].first; ev1 = w->events[w->events_used].second;
      w->events_used++;
      HIP_TRY(hipEventRecord(ev0, s));
    }
    const bool fused_lp = c.enable_demo && rollout_fuses_logpd(e, P * N, H, sw);  // (mbd_plan.hip: the log-densities out of the rollout)
    int rc = launch_rollout(e, w->d_state0, w->d_eps[cur], P * N, H, w->d_rewss, w->d_rews,
                            (c.enable_demo && !fused_lp) ? w->d_xpos : nullptr, nullptr, s, &lz, sw, fused_lp ? w->d_lp : nullptr);
    if (rc != MBD_OK) return rc;
    if (w->timing) HIP_TRY(hipEventRecord(ev1, s));
    if (c.enable_demo && !fused_lp) {
      rc = launch_logpd(e, w->d_xpos, P * N, H, w->d_lp, s);
      if (rc != MBD_OK) return rc;
    }
    sb.ybar_in = ybar_in_stride;
    launch_score_wmean_batch(wmean_batch_v(), HNu, P, sizeof(float) * (size_t)N, s, w->d_rews, c.enable_demo ? w->d_lp : nullptr, N,
                             e->rew_xref, c.temp_sample, 1, w->d_weights, w->d_rewmeans + step, w->d_eps[cur], HNu, ybar_in,
                             w->alphas[i], w->alphas_bar[i], w->alphas_bar[i - 1], c.literal_score,
                             w->d_mu + (size_t)step * HNu, 1, w->sigmas[i], (float*)nullptr, sb);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(s));
  auto t1 = std::chrono::steady_clock::now();
  if (loop_seconds_out) *loop_seconds_out = std::chrono::duration<double>(t1 - t0).count();
  const size_t mu_n = (size_t)P * (Nd - 1) * HNu;
  if (mu_0ts_out) HIP_TRY(hipMemcpy(mu_0ts_out, w->d_mu, sizeof(float) * mu_n, hipMemcpyDeviceToHost));
  if (rew_means_out) HIP_TRY(hipMemcpy(rew_means_out, w->d_rewmeans, sizeof(float) * (size_t)P * (Nd - 1), hipMemcpyDeviceToHost));
  if (rew_final_out) {  // rollout_us(state_init, Yi[-1]).mean() of every plan (mbd_planner.py:179-180): one launch of P candidates
    HIP_TRY(hipMemcpy2DAsync(w->d_final, sizeof(float) * HNu, w->d_mu + (size_t)(Nd - 2) * HNu,
                             sizeof(float) * (size_t)(Nd - 1) * HNu, sizeof(float) * HNu, P, hipMemcpyDeviceToDevice, s));
    const int sw[3] = {1, S, 0};
    int rc = launch_rollout(e, w->d_state0, w->d_final, P, H, nullptr, w->d_final_rew, nullptr, nullptr, s, nullptr, sw);
    if (rc != MBD_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(rew_final_out, w->d_final_rew, sizeof(float) * P, hipMemcpyDeviceToHost));
  }
  return MBD_OK;
}
