// mbd_step_kernels.h — the kernels of a diffusion step around the rollout (mbd_planner.py:103-135): sampling / noise,
// demo log-densities, score (standardise, demo blend, softmax), weighted mean + score update, their batched forms for sweeps,
// the path-integral update rules, and the car2d rollout.  Non-template and `static`: each host unit (mbd_env / mbd_plan /
// mbd_sweep .hip) compiles the ones it launches.  The rollout kernels and what they share live in mbd_kernels.h.
#pragma once

#include "mbd_kernels.h"

namespace mbd {

// ---- car2d (mbd/envs/car2d.py): one candidate per lane -------------------------------------------------
struct Car2dParams {
  const float* q0;  // [3]
  const float* us;  // [B][H][2]
  float* rewss;     // [B][H] or nullptr
  float* rews;      // [B] or nullptr
  float* qs;        // [B][H][3] or nullptr
  float* q_final;   // [B][3] or nullptr
  int B, H;
};

__device__ __forceinline__ void car_dyn(const float x[3], float u0, float u1, float dx[3]) {
  float sn, cs;
  sincos_(x[2], &sn, &cs);
  dx[0] = u1 * sn * 3.0f;
  dx[1] = u1 * cs * 3.0f;
  dx[2] = u0 * 3.14159274101257324f / 3.0f * 2.0f;
}
__device__ __forceinline__ float car_reward(const float q[3]) {
  float dx = q[0] - 0.5f, dy = q[1] - 0.0f;
  // (sqrt_floor is the correctly rounded square root from 1e-30 up — exhaustive, probe_short — and its floor of 1e-15
  // below that leaves the reward at exactly 1, like the true root)
  float d = sqrt_floor(dx * dx + dy * dy);
  d = fclip(d, 0.0f, 0.2f);
  float t = d / 0.2f;
  return 1.0f - t * t;
}

static __global__ __launch_bounds__(64) void car2d_rollout_kernel(Car2dParams P) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  // obstacle centres (car2d.py:48-63): python float64 products cast to f32
  const float cx[11] = {(float)(0.3 * -3), (float)(0.3 * -2), (float)(0.3 * -1), 0.0f, 0.0f, 0.0f, 0.0f,
                        (float)(0.3 * -3), (float)(0.3 * -2), (float)(0.3 * -1), 0.0f};
  const float cy[11] = {(float)(0.3 * 2), (float)(0.3 * 2), (float)(0.3 * 2), (float)(0.3 * 2),
                        (float)(0.3 * 1), 0.0f, (float)(0.3 * -1), (float)(0.3 * -2), (float)(0.3 * -2),
                        (float)(0.3 * -2), (float)(0.3 * -2)};
  const float dt = (float)0.1, dt2 = (float)(0.1 / 2), dt6 = (float)(0.1 / 6);
  float q[3] = {P.q0[0], P.q0[1], P.q0[2]};
  float sum = 0.0f;
  for (int t = 0; t < P.H; ++t) {
    const float* u = P.us + ((size_t)b * P.H + t) * 2;
    float a0 = fclip(u[0], -1.0f, 1.0f), a1 = fclip(u[1], -1.0f, 1.0f);
    float k1[3], k2[3], k3[3], k4[3], x[3], qn[3];
    car_dyn(q, a0, a1, k1);
    for (int i = 0; i < 3; ++i) x[i] = q[i] + dt2 * k1[i];
    car_dyn(x, a0, a1, k2);
    for (int i = 0; i < 3; ++i) x[i] = q[i] + dt2 * k2[i];
    car_dyn(x, a0, a1, k3);
    for (int i = 0; i < 3; ++i) x[i] = q[i] + dt * k3[i];
    car_dyn(x, a0, a1, k4);
    for (int i = 0; i < 3; ++i) qn[i] = q[i] + dt6 * (k1[i] + 2.0f * k2[i] + 2.0f * k3[i] + k4[i]);
    bool collide = false;
    for (int i = 0; i < 11; ++i) {
      float dx = qn[0] - cx[i], dy = qn[1] - cy[i];
      // sqrtf(x) < 0.3f  <=>  x < 0x1.70a3d8p-4 (0.09000000357627869): the correctly rounded square root is monotonic and
      // that is the smallest float32 whose root rounds to >= 0.3f (tests/test_spec_math.py checks the window around it)
      collide = collide || (dx * dx + dy * dy < 0.09000000357627869f);
    }
    for (int i = 0; i < 3; ++i) q[i] = collide ? q[i] : qn[i];
    float rew = car_reward(q);
    sum = sum + rew;
    if (P.rewss) P.rewss[(size_t)b * P.H + t] = rew;
    if (P.qs) { float* o = P.qs + ((size_t)b * P.H + t) * 3; o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; }
  }
  if (P.rews) P.rews[b] = sum / (float)P.H;
  if (P.q_final) { P.q_final[b * 3] = q[0]; P.q_final[b * 3 + 1] = q[1]; P.q_final[b * 3 + 2] = q[2]; }
}

// ---- A1: sampling (mbd_planner.py:103-106) -------------------------------------------------------------
// Y0s[e] for the flat elements [e_begin, e_begin + e_count) of the global [N][HNu] tensor (a rank's own rows
// first, the other ranks' rows on a second stream while the rollout runs).  Whole tensor, legacy layout: one
// thread per threefry block, which pairs element j with j+half (both outputs used).  Otherwise one thread per
// element (partitionable layout: its own block; legacy layout on a sub-range: the block it belongs to).
static __global__ __launch_bounds__(256) void sample_kernel(uint32_t k0, uint32_t k1, int impl, int N, int HNu,
                                                      unsigned long long e_begin, unsigned long long e_count,
                                                      float sigma_host, const float* __restrict__ sigma_dev,
                                                      const float* __restrict__ Ybar, float* __restrict__ Y0s) {
  const float sigma = sigma_dev ? *sigma_dev : sigma_host;  // path-integral plans carry sigma on the device
  const uint64_t size = (uint64_t)N * (uint64_t)HNu;
  const uint64_t half = (size + 1) / 2;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (impl == 1) {
    if (tid >= e_count) return;
    const uint64_t e = e_begin + tid;
    uint32_t o0, o1;
    threefry2x32(k0, k1, (uint32_t)(e >> 32), (uint32_t)e, o0, o1);
    float eps = bits_to_normal(o0 ^ o1);
    float y = eps * sigma + Ybar[e % (uint64_t)HNu];
    Y0s[e] = fclip(y, -1.0f, 1.0f);
    return;
  }
  if (e_begin == 0 && e_count == size) {
    if (tid >= half) return;
    const uint64_t j1 = tid + half;
    uint32_t o0, o1;
    threefry2x32(k0, k1, (uint32_t)tid, j1 < size ? (uint32_t)j1 : 0u, o0, o1);
    {
      float y = bits_to_normal(o0) * sigma + Ybar[tid % (uint64_t)HNu];
      Y0s[tid] = fclip(y, -1.0f, 1.0f);
    }
    if (j1 < size) {
      float y = bits_to_normal(o1) * sigma + Ybar[j1 % (uint64_t)HNu];
      Y0s[j1] = fclip(y, -1.0f, 1.0f);
    }
    return;
  }
  if (tid >= e_count) return;
  const uint64_t e = e_begin + tid;
  const uint64_t j0 = e < half ? e : e - half, j1 = j0 + half;
  uint32_t o0, o1;
  threefry2x32(k0, k1, (uint32_t)j0, j1 < size ? (uint32_t)j1 : 0u, o0, o1);
  float y = bits_to_normal(e < half ? o0 : o1) * sigma + Ybar[e % (uint64_t)HNu];
  Y0s[e] = fclip(y, -1.0f, 1.0f);
}

// The sampler in two halves, for LAZY plans (RolloutParams): eps does not depend on the previous step's result, only
// the shift by Ybar does.
//   noise_kernel  eps = normal(key, (N, HNu)) — the threefry counters, layouts and the ErfInv polynomial of
//                 sample_kernel (grid-stride: any grid).  The same noise_fill runs in the noise workgroups of a rollout
//                 launch, which generate the NEXT step's normals on the CUs the rollout leaves idle.
//   shift_kernel  Y0s[e] = clip(eps[e] * sigma + Ybar[e mod HNu], -1, 1) — the same two roundings as sample_kernel;
//                 lazy plans form these values at the rollout's action fetch and inside the weighted mean instead, and
//                 run this kernel only when somebody asks for Y0s (mbd_plan_peek)
static __global__ __launch_bounds__(256) void noise_kernel(uint32_t k0, uint32_t k1, int impl, int N, int HNu,
                                                     float* __restrict__ eps) {
  noise_fill(k0, k1, impl, (uint64_t)N * (uint64_t)HNu, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x,
             (uint64_t)gridDim.x * blockDim.x, eps);
}
static __global__ __launch_bounds__(256) void shift_kernel(const float* __restrict__ eps, int HNu, unsigned long long e_begin,
                                                     unsigned long long e_count, float sigma_host,
                                                     const float* __restrict__ sigma_dev,
                                                     const float* __restrict__ Ybar, float* __restrict__ Y0s) {
  const float sigma = sigma_dev ? *sigma_dev : sigma_host;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= e_count) return;
  const uint64_t e = e_begin + tid;
  float y = eps[e] * sigma + Ybar[e % (uint64_t)HNu];
  Y0s[e] = fclip(y, -1.0f, 1.0f);
}

// ---- A5: demo log-densities ------------------------------------------------------------------------------
// HumanoidTrack.eval_xref_logpd (humanoidtrack.py:98-106): xpos [B][H][K][3], xref [K][H][3].  One workgroup per
// candidate: its K*H terms ((clip(|x - xref|, 0, .5) / .5)^2, the candidate's 3 K H floats are contiguous) are formed in
// parallel and parked in LDS; thread k then adds link k's H terms in t order (S_k), thread 0 the K sums in link order —
// the contract's order since round 6 (rounds 1-5: one chain over all K H terms), chosen so that the rollout kernels can
// accumulate S_k on the tracked link's lane as the control steps go by (RolloutParams::lp) and produce the same bits
// without this launch; the standalone entry (mbd_env_xref_logpd) and the instantiations that do not accumulate use this kernel.
constexpr int kLogpdThreads = 256, kLogpdMaxTerms = MBD_MAX_TRACK * 64;
static __global__ __launch_bounds__(kLogpdThreads) void logpd_track_kernel(const float* __restrict__ xpos,
                                                                    const float* __restrict__ xref, int B, int H, int K,
                                                                    float* __restrict__ lp) {
  __shared__ float term[kLogpdMaxTerms];
  const int b = blockIdx.x;
  const int n = K * H;
  for (int i = threadIdx.x; i < n; i += kLogpdThreads) {
    const int t = i / K, k = i - t * K;  // (xpos order: t outer, k inner)
    const float* a = xpos + ((size_t)b * n + i) * 3;
    const float* c = xref + ((size_t)k * H + t) * 3;
    float ex = a[0] - c[0], ey = a[1] - c[1], ez = a[2] - c[2];
    float d = fsqrt(ex * ex + ey * ey + ez * ez);
    d = fclip(d, 0.0f, 0.5f);
    float s = d / 0.5f;
    term[k * H + t] = s * s;
  }
  __shared__ float part[MBD_MAX_TRACK];
  __syncthreads();
  if ((int)threadIdx.x < K) {
    float a = 0.0f;
    for (int t = 0; t < H; ++t) a = a + term[threadIdx.x * H + t];
    part[threadIdx.x] = a;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float acc = part[0];
  for (int k = 1; k < K; ++k) acc = acc + part[k];
  lp[b] = 0.0f - acc / (float)n;
}
// Car2d.eval_xref_logpd (car2d.py:95-102): qs [B][H][3], xref [H][2]
static __global__ __launch_bounds__(64) void logpd_car2d_kernel(const float* __restrict__ qs,
                                                         const float* __restrict__ xref, int B, int H,
                                                         float* __restrict__ lp) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float acc = 0.0f;
  for (int t = 0; t < H; ++t) {
    const float* a = qs + ((size_t)b * H + t) * 3;
    float ex = a[0] - xref[2 * t], ey = a[1] - xref[2 * t + 1];
    float d = fsqrt(ex * ex + ey * ey);
    d = fclip(d, 0.0f, 0.5f);
    float s = d / 0.5f;
    acc = acc + s * s;
  }
  lp[b] = 0.0f - acc / (float)H;
}

// ---- A4-A6: standardise, demo blend, softmax -> weights[N] (mbd_planner.py:110-127) -------------------
// The canonical one-wavefront reduction of the numerical contract: lane j accumulates the elements
// i = j, j+64, ... in increasing i, then a xor-butterfly over the 64 lanes.
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = x + __shfl_xor(x, off, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = fmax_(x, __shfl_xor(x, off, 64));
  return x;
}
// ONE 1024-thread workgroup.  Reduction order of the contract ("sumB"): thread t accumulates
// i = t, t+1024, ... in increasing i; each wavefront runs the xor-butterfly; the 16 wavefront sums are added
// sequentially in wavefront order (every thread does that same sum from LDS).
constexpr int kScoreThreads = 1024;
__device__ __forceinline__ float block_sum(float x, float* red) {
  x = wave_sum(x);
  __syncthreads();  // red[] may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int w = 1; w < kScoreThreads / 64; ++w) s = s + red[w];
  return s;
}
__device__ __forceinline__ float block_max(float x, float* red) {
  x = wave_max(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int w = 1; w < kScoreThreads / 64; ++w) s = fmax_(s, red[w]);
  return s;
}

// sum and max of two values in one pass over the barriers (red: 2 x 16 entries)
__device__ __forceinline__ void block_sum_max(float& xs, float& xm, float* red) {
  xs = wave_sum(xs);
  xm = wave_max(xm);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = xs;
    red[kScoreThreads / 64 + (threadIdx.x >> 6)] = xm;
  }
  __syncthreads();
  float s = red[0], m = red[kScoreThreads / 64];
#pragma unroll
  for (int w = 1; w < kScoreThreads / 64; ++w) {
    s = s + red[w];
    m = fmax_(m, red[kScoreThreads / 64 + w]);
  }
  xs = s;
  xm = m;
}

// The body of the score step for ONE 1024-thread workgroup: leaves exp(logp0 - max) of every candidate in lg[] (LDS
// or a global scratch; each thread only ever touches its own entries i = tid, tid + 1024, ...) and returns the softmax
// denominator; the caller divides.  Shared by score_kernel and by the fused score + weighted-mean kernel, whose
// every workgroup re-derives the weights — same code, same order, same bits.
template <bool REGS = true>  // REGS = false: the rewards are re-read per pass (sweeps: registers for two workgroups per CU)
__device__ __forceinline__ float score_block(const float* __restrict__ rews, const float* __restrict__ lp_demo, int N,
                                             float rew_xref, float temp, int std_guard, float* __restrict__ lg,
                                             float* red, float& rew_mean_out, float* early_mean = nullptr) {
  const int tid = threadIdx.x;
  // the rewards cross from memory ONCE (up to 8 per thread: plans up to 8192 candidates; the passes below are then
  // register arithmetic — as three reads of rews[] each pass paid an L2 round trip), and their maximum rides on the first
  // reduction: logp0 is a monotonic function of the reward (the same three correctly rounded operations for every
  // candidate), so max(logp0) = logp0(max reward) exactly — one reduction fewer for plans without the demonstration blend
  constexpr int RK = 8;
  const bool in_regs = REGS && N <= RK * kScoreThreads;  // (uniform)
  float r[RK];
#pragma unroll
  for (int k = 0; k < RK; ++k) r[k] = (in_regs && tid + k * kScoreThreads < N) ? rews[tid + k * kScoreThreads] : 0.0f;
  float part = 0.0f, rmax = -__builtin_inff();
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      const bool ok = tid + k * kScoreThreads < N;
      part = ok ? part + r[k] : part;
      rmax = ok ? fmax_(rmax, r[k]) : rmax;
    }
  } else {
    for (int i = tid; i < N; i += kScoreThreads) { part = part + rews[i]; rmax = fmax_(rmax, rews[i]); }
  }
  block_sum_max(part, rmax, red);
  const float rew_mean = part / (float)N;
  // (the caller's early copy of the step's mean reward: a host that polls for it turns around ~5 us sooner)
  if (early_mean && tid == 0) __hip_atomic_store(early_mean, rew_mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  part = 0.0f;
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      const float d = r[k] - rew_mean;
      part = tid + k * kScoreThreads < N ? ffma(d, d, part) : part;
    }
  } else {
    for (int i = tid; i < N; i += kScoreThreads) {
      float d = rews[i] - rew_mean;
      part = ffma(d, d, part);
    }
  }
  float rew_std = fsqrt(block_sum(part, red) / (float)N);
  rew_std = (std_guard && rew_std < 1e-4f) ? 1.0f : rew_std;  // mbd_planner.py:112; path_integral.py:123 has none
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < RK; ++k)
      if (tid + k * kScoreThreads < N) lg[tid + k * kScoreThreads] = ((r[k] - rew_mean) / rew_std) / temp;
  } else {
    for (int i = tid; i < N; i += kScoreThreads) lg[i] = ((rews[i] - rew_mean) / rew_std) / temp;
  }
  if (lp_demo) {  // (each thread only ever touches its own lg[i]: no barrier needed around them)
    float mx = -__builtin_inff();
    for (int i = tid; i < N; i += kScoreThreads) mx = fmax_(mx, lp_demo[i]);
    mx = block_max(mx, red);
    part = 0.0f;
    for (int i = tid; i < N; i += kScoreThreads) {
      float lpd = ((((lp_demo[i] - mx) + rew_xref) - rew_mean) / rew_std) / temp;
      float v = lpd > lg[i] ? lpd : lg[i];
      lg[i] = v;
      part = part + v;
    }
    const float m = block_sum(part, red) / (float)N;
    part = 0.0f;
    for (int i = tid; i < N; i += kScoreThreads) {
      float d = lg[i] - m;
      part = ffma(d, d, part);
    }
    const float sd = fsqrt(block_sum(part, red) / (float)N);
    for (int i = tid; i < N; i += kScoreThreads) lg[i] = ((lg[i] - m) / sd) / temp;
  }
  float mx;
  if (lp_demo || !std_guard) {  // (without the guard a zero deviation makes logp0 NaN: keep the reduction's own answer)
    mx = -__builtin_inff();
    for (int i = tid; i < N; i += kScoreThreads) mx = fmax_(mx, lg[i]);
    mx = block_max(mx, red);
  } else {
    mx = ((rmax - rew_mean) / rew_std) / temp;
  }
  part = 0.0f;
  for (int i = tid; i < N; i += kScoreThreads) {
    float e = exp_(lg[i] - mx);
    lg[i] = e;
    part = part + e;
  }
  rew_mean_out = rew_mean;
  return block_sum(part, red);
}

// Sweeps of path-integral plans (mbd_sweep_*, update_method != 0): the small kernels of the update rules take blockIdx.y
// = plan and these strides (in elements) from plan 0's data; a single plan launches them with one row and zero strides.
struct PiBatch {
  long long rews = 0, weights = 0, mean = 0, cand = 0, mu = 0, out = 0, spread = 0, sigma = 0, idx = 0;
  const float* temps = nullptr;  // [P], or nullptr: every plan at the launch's `temp`
};
static __global__ __launch_bounds__(kScoreThreads) void score_kernel(const float* __restrict__ rews,
                                                              const float* __restrict__ lp_demo, int N,
                                                              float rew_xref, float temp, int std_guard,
                                                              float* __restrict__ weights,
                                                              float* __restrict__ rew_mean_out,
                                                              float* __restrict__ lg_global, PiBatch pb) {
  const long long plan = blockIdx.y;
  rews += plan * pb.rews; weights += plan * pb.weights; rew_mean_out += plan * pb.mean;
  if (lp_demo) lp_demo += plan * pb.rews;
  if (pb.temps) temp = pb.temps[plan];
  // logp0 [N]: in LDS while it fits (every plan of the reference's sizes), in a plan-owned global scratch beyond
  // (each thread only ever touches its own entries, so the scratch needs no synchronisation either)
  extern __shared__ __attribute__((aligned(16))) float lg_lds[];
  float* __restrict__ lg = lg_global ? lg_global : lg_lds;
  __shared__ float red[2 * (kScoreThreads / 64)];
  float rew_mean;
  const float den = score_block<true>(rews, lp_demo, N, rew_xref, temp, std_guard, lg, red, rew_mean);
  for (int i = threadIdx.x; i < N; i += kScoreThreads) weights[i] = lg[i] / den;
  if (threadIdx.x == 0) *rew_mean_out = rew_mean;
}

// ---- A7-A8: weighted mean + score update (mbd_planner.py:128-133) ---------------------------------------
// A workgroup owns kWmE = 16 consecutive outputs e of [H][Nu] and splits the candidates into 64 groups:
// thread (g, j) runs a sequential fma over n = g, g+64, ... for output j (a wavefront reads four 64-byte row
// segments per load instruction); the 64 partials of an output are then added sequentially in g
// ("wsum64" of the contract).  ceil(HNu/16) workgroups of 1024 threads: every load of a thread is in flight at once.
constexpr int kWmE = 16, kWmG = 64;
// XCD-aware tile order.  Workgroups are dispatched round-robin over the chip's 8 XCDs — each with its own L2 — in the order
// of their linear index; a tile of 16 outputs reads 64-byte row segments, HALF a 128-byte line, so the tile next door wants
// the same lines again.  The workgroups that land on one XCD therefore take CONSECUTIVE tiles and the shared lines meet
// in one L2 instead of being fetched by two (PMC, round 4: 3.0x the algorithmic bytes per launch with tile = blockIdx.x).
// x: index within the row of n workgroups, first: linear index of the row's first workgroup (sweeps: blockIdx.y * n).
// A bijection on [0, n); which workgroup computes which outputs does not change their values.
__device__ __forceinline__ int xcd_tile(int x, int n, int first) {
  const int c = (first + x) & 7;
  int start = 0;
  for (int d = 0; d < 8; ++d) {
    if (d == c) break;
    const int x0 = (d - first) & 7;  // the first workgroup of the row on XCD d
    start += x0 < n ? (n - 1 - x0) / 8 + 1 : 0;
  }
  return start + (x - ((c - first) & 7)) / 8;
}
static __global__ __launch_bounds__(kWmE * kWmG) void wmean_kernel(const float* __restrict__ weights,
                                                            const float* __restrict__ Y0s, int N, int HNu,
                                                            const float* __restrict__ Ybar_i, float alpha_i,
                                                            float alpha_bar_i, float alpha_bar_im1, int literal,
                                                            float* __restrict__ Ybar_im1, int lazy, float sigma,
                                                            float* __restrict__ ybar_keep) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // the N weights: read once, shared by the 16 outputs
  __shared__ float red[kWmG][kWmE + 1];
  const int j = threadIdx.x & (kWmE - 1), g = threadIdx.x / kWmE;
  const int e_raw = xcd_tile(blockIdx.x, gridDim.x, 0) * kWmE + j;
  const int e = e_raw < HNu ? e_raw : HNu - 1;
  const float* __restrict__ col = Y0s + e;
  // lazy plans: Y0s holds the step's normals; the candidate value is formed here exactly as at the rollout's fetch
  const float yb = lazy ? Ybar_i[e] : 0.0f;
  auto val = [&](float x) { return lazy ? fclip(x * sigma + yb, -1.0f, 1.0f) : x; };
  for (int i = threadIdx.x; i < N; i += kWmE * kWmG) wl[i] = weights[i];
  __syncthreads();  // (plans of >= 4096 candidates take the row-major kernels below: N always fits here)
  float acc = 0.0f;
  int n = g;
  // the chain over n is sequential by contract, its loads are not: a thread keeps 32 rows of Y0s in flight while
  // there are that many, then 16, then the tail (the kernel is bound by memory round trips per batch, not by
  // bandwidth — multi-GPU plans average over all N_total candidates on every rank).  The scheduling barrier keeps
  // the compiler from interleaving loads and the dependent fma chain at a shallower depth.
  for (; n + 31 * kWmG < N; n += 32 * kWmG) {
    float y[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) y[k] = col[(size_t)(n + k * kWmG) * HNu];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = ffma(wl[n + k * kWmG], val(y[k]), acc);
  }
  for (; n + 15 * kWmG < N; n += 16 * kWmG) {
    float y[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] = col[(size_t)(n + k * kWmG) * HNu];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = ffma(wl[n + k * kWmG], val(y[k]), acc);
  }
  for (; n < N; n += kWmG) acc = ffma(wl[n], val(col[(size_t)n * HNu]), acc);
  red[g][j] = acc;
  __syncthreads();
  if (g != 0 || e_raw >= HNu) return;
  if (ybar_keep) ybar_keep[e] = Ybar_i[e];  // what mbd_plan_peek needs to materialise Y0s after the caller moved on
  float tot = red[0][j];
#pragma unroll 8
  for (int k = 1; k < kWmG; ++k) tot = tot + red[k][j];
  float out = tot;
  if (literal) {
    const float sab = fsqrt(alpha_bar_i);
    float Yi = Ybar_i[e] * sab;
    float t1 = 1.0f / (1.0f - alpha_bar_i);
    float t2 = sab * tot;
    float score = t1 * (-Yi + t2);
    float t3 = (1.0f - alpha_bar_i) * score;
    float Yim1 = (1.0f / fsqrt(alpha_i)) * (Yi + t3);
    out = Yim1 / fsqrt(alpha_bar_im1);
  }
  Ybar_im1[e] = out;
}

// Score + weighted mean in ONE launch (plans below 4096 candidates): every workgroup of the tile kernel above first
// re-derives the softmax weights from the N rewards — score_block, the code and order of score_kernel — straight into
// the LDS array the weighted mean reads them from; workgroup 0 also stores them and the step's mean reward.  A kernel
// of a few microseconds is mostly launch latency (an empty kernel takes 4 us on the timeline): one launch instead of
// two takes ~3.5 us off every step, and the rows of the candidates are already in flight while the weights are derived.
// V (round 5): outputs per thread — a workgroup owns 16 V CONSECUTIVE outputs (64 V bytes of every candidate row), thread
// (g, j) runs the V chains of outputs j V .. j V + V - 1 over its candidates n = g, g + 64, ...: the chains, their order and
// the order of the 64 partials are the contract's whatever V is (same bits).  What V changes is how many 128-byte lines a
// row segment shares with the tiles next door: a 64-byte segment touches 1.5 lines on average and shares each with a
// neighbour — when the neighbours drift apart in time and the line has left the L2 in between it is fetched twice (the sweep's
// batch form: 2.0x its algorithmic bytes, PMC round 4); a 256-byte segment touches 3 lines for 2 lines' worth of data:
// 1.5x at worst, whatever the L2 does.
template <int ROUND, bool REGS, int V = 1>  // rows in flight per round of the chain, score_block's REGS (48, true: one plan; sweeps: 32, false)
__device__ __forceinline__ void score_wmean_body(
    const float* __restrict__ rews, const float* __restrict__ lp_demo, int N, float rew_xref, float temp, int std_guard,
    float* __restrict__ weights_out, float* __restrict__ rew_mean_out, const float* __restrict__ Y0s, int HNu,
    const float* __restrict__ Ybar_i, float alpha_i, float alpha_bar_i, float alpha_bar_im1, int literal,
    float* __restrict__ Ybar_im1, int lazy, float sigma, float* __restrict__ ybar_keep, int tile, bool writer) {
  // tile: which 16 outputs this workgroup owns (XCD-aware, xcd_tile / the batch kernel); writer: the one workgroup of the
  // plan that stores the weights and the mean reward (every workgroup derives the same values)
  static_assert(kWmE * kWmG == kScoreThreads, "score_block runs on the weighted mean's workgroup");
  extern __shared__ __attribute__((aligned(16))) float wl[];  // logp0, then the N weights
  __shared__ float red_s[2 * (kScoreThreads / 64)];
  __shared__ float red[kWmG][kWmE * V + 1];
  const int j = threadIdx.x & (kWmE - 1), g = threadIdx.x / kWmE;
  const int e0 = tile * (kWmE * V) + j * V;  // this thread's first output
  int e[V];
  float yb[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    e[v] = e0 + v < HNu ? e0 + v : HNu - 1;
    yb[v] = lazy ? Ybar_i[e[v]] : 0.0f;
  }
  auto val = [&](float x, int v) { return lazy ? fclip(x * sigma + yb[v], -1.0f, 1.0f) : x; };
  // the first rows of this thread's chains leave now and land while the weights are derived
  constexpr int PRE = 16 / V;
  const bool pre = g + (PRE - 1) * kWmG < N;
  float y0[PRE][V];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
#pragma unroll
    for (int v = 0; v < V; ++v) y0[k][v] = pre ? Y0s[(size_t)(g + k * kWmG) * HNu + e[v]] : 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
  float rew_mean;
  // the step's mean reward may go to a pinned host slot the host is polling (per-step progress, mbd_planner.py:147):
  // a system-scope store leaves as soon as the first reduction has it — a plain one would sit in the cache until the
  // kernel ends — so the host turns around while the rest of the score and the weighted mean below run
  const float den = score_block<REGS>(rews, lp_demo, N, rew_xref, temp, std_guard, wl, red_s, rew_mean,
                                      writer ? rew_mean_out : nullptr);
  for (int i = threadIdx.x; i < N; i += kScoreThreads) {
    const float w = wl[i] / den;
    wl[i] = w;
    if (writer) weights_out[i] = w;
  }
  __syncthreads();
  float acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = 0.0f;
  int n = g;
  if (pre) {  // (the chain over n is sequential by contract: the prefetched rows are its first terms)
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const float w = wl[n + k * kWmG];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = ffma(w, val(y0[k][v], v), acc[v]);
    }
    n += PRE * kWmG;
  }
  // (a round = the rows in flight at once, then their dependent fma chain: large plans are bound by the number of rounds —
  // 48 rows per round: N = 4096 one round behind the prefetch instead of two, N = 8192 three instead of four)
  constexpr int RND = ROUND / V;
  for (; n + (RND - 1) * kWmG < N; n += RND * kWmG) {
    float y[RND][V];
#pragma unroll
    for (int k = 0; k < RND; ++k) {
#pragma unroll
      for (int v = 0; v < V; ++v) y[k][v] = Y0s[(size_t)(n + k * kWmG) * HNu + e[v]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < RND; ++k) {
      const float w = wl[n + k * kWmG];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = ffma(w, val(y[k][v], v), acc[v]);
    }
  }
  constexpr int TAIL = 16 / V;
  for (; n + (TAIL - 1) * kWmG < N; n += TAIL * kWmG) {
    float y[TAIL][V];
#pragma unroll
    for (int k = 0; k < TAIL; ++k) {
#pragma unroll
      for (int v = 0; v < V; ++v) y[k][v] = Y0s[(size_t)(n + k * kWmG) * HNu + e[v]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < TAIL; ++k) {
      const float w = wl[n + k * kWmG];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = ffma(w, val(y[k][v], v), acc[v]);
    }
  }
  for (; n < N; n += kWmG) {
    const float w = wl[n];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = ffma(w, val(Y0s[(size_t)n * HNu + e[v]], v), acc[v]);
  }
#pragma unroll
  for (int v = 0; v < V; ++v) red[g][j * V + v] = acc[v];
  __syncthreads();
  // the 16 V outputs of the tile, one per thread of the first 16 V threads: the 64 partials of an output in group order
  if ((int)threadIdx.x >= kWmE * V) return;
  const int jo = threadIdx.x, eo = tile * (kWmE * V) + jo;
  if (eo >= HNu) return;
  if (ybar_keep) ybar_keep[eo] = Ybar_i[eo];
  float tot = red[0][jo];
#pragma unroll 8
  for (int k = 1; k < kWmG; ++k) tot = tot + red[k][jo];
  float out = tot;
  if (literal) {
    const float sab = fsqrt(alpha_bar_i);
    float Yi = Ybar_i[eo] * sab;
    float t1 = 1.0f / (1.0f - alpha_bar_i);
    float t2 = sab * tot;
    float score = t1 * (-Yi + t2);
    float t3 = (1.0f - alpha_bar_i) * score;
    float Yim1 = (1.0f / fsqrt(alpha_i)) * (Yi + t3);
    out = Yim1 / fsqrt(alpha_bar_im1);
  }
  Ybar_im1[eo] = out;
}
// XCD PINNING of the single-plan launches (round 5).  Workgroup i of a launch lands on XCD i mod 8.  A launch of T tiles is
// therefore made 8 ceil(T / X) workgroups long, and only those with (i mod 8) < X work — on tile (i mod 8) ceil(T / X) + i / 8
// — the others leave at once: the tiles of the launch then sit on X of the 8 XCDs, CONSECUTIVE tiles behind one L2, and a
// line of a candidate row that two tiles share is fetched from HBM once per XCD instead of once per tile.  X (host side,
// wmean_xcds): the fewest XCDs that give every tile a CU of its own (32 per XCD) and keep an XCD's share of the normals
// within its 4 MB L2.  hopper512's ten tiles sat on eight XCDs and read 3.3x their algorithmic bytes, the metric's 54 tiles
// 1.35x (seven tiles per XCD: 448 bytes of a row, plus a boundary line at each end).  -1: no such workgroup.
__device__ __forceinline__ int pinned_tile(int i, int T, int X) {
  const int c = i & 7, per = (T + X - 1) / X, t = c * per + (i >> 3);
  return (c < X && (i >> 3) < per && t < T) ? t : -1;
}
// V (round 6): outputs per thread of the single-plan launch, like the sweeps'.  Built to test the round-5 verdict's reading of
// the launch's extra traffic at large N (1.30x its algorithmic bytes at N = 8192, 1.15x at 4096) as half-line sharing between
// neighbouring tiles; MEASURED (profiles/r06_score_v1_ab.txt): V = 2 moves the bytes by 0 ... 3 % (36.3 -> 35.2 MB) and makes
// the kernel 45 ... 55 % SLOWER (27 workgroups in flight instead of 54: 15.5 -> 22.4 us at N = 4096, 25.3 -> 38.8 at 8192).
// The extra bytes are the XCD partition's boundary lines instead: a candidate's row is 3400 bytes — not a multiple of the
// 128-byte line — so the 448 bytes of a row that one XCD's seven tiles own (N = 8192: X = 8) start anywhere in a line and
// touch 4.5 lines on average for 3.5 lines of data, 1.29x; fourteen tiles (N = 4096: X = 4) 8 for 7, 1.14x — the measured
// ratios.  Only a row stride padded to whole lines would remove them (the normals' flat index would then no longer be their
// address: sampler, rollout fetch and the peek entries all change) — for a launch that is 3 % of a step.  V = 1 stays the
// library's choice at every size; MBD_WMEAN_V1 = 2 runs this form (same chains, same order, same bits).
template <int V>
static __global__ __launch_bounds__(kWmE * kWmG) void score_wmean_kernel(
    const float* __restrict__ rews, const float* __restrict__ lp_demo, int N, float rew_xref, float temp, int std_guard,
    float* __restrict__ weights_out, float* __restrict__ rew_mean_out, const float* __restrict__ Y0s, int HNu,
    const float* __restrict__ Ybar_i, float alpha_i, float alpha_bar_i, float alpha_bar_im1, int literal,
    float* __restrict__ Ybar_im1, int lazy, float sigma, float* __restrict__ ybar_keep, int T, int X) {
  const int tile = pinned_tile(blockIdx.x, T, X);
  if (tile < 0) return;  // (wave-uniform: the whole workgroup)
  score_wmean_body<48, true, V>(rews, lp_demo, N, rew_xref, temp, std_guard, weights_out, rew_mean_out, Y0s, HNu, Ybar_i, alpha_i,
                   alpha_bar_i, alpha_bar_im1, literal, Ybar_im1, lazy, sigma, ybar_keep, tile, tile == 0);
}
// SWEEPS (mbd_sweep_*): the same for P plans of one env in ONE launch — blockIdx.y is the plan, whose buffers sit at
// fixed strides (in floats) from plan 0's; temperatures may differ per plan (run_mbd.py:42-64).  The body is the
// single-plan kernel's: same code, same order, same bits.
struct ScoreBatch {
  long long rews, lp, weights, mean, cand, ybar_in, ybar_out, keep;
  const float* temps;  // [P], or nullptr: every plan at `temp`
};
// V = 1: eight wavefronts per SIMD — two workgroups per CU, 64 registers: the P x 54 workgroups of a sweep's step in one
// round (rounds 3-4).  V = 2 (round 5, the default for sweeps: mbd_sweep.hip wmean_batch_v): P x 27 workgroups of 32 outputs
// each, ONE per CU of plan k's XCD — they start together and stay together, a row segment is 128 bytes: the 2.0x of V = 1
// (plan k's 54 tiles, two to a CU, drifting apart behind XCD k's 4 MB L2 while 3.5 MB of normals stream through it) is gone:
// sweep8 28.2 MB per launch for 28.0 MB algorithmic, 11.3 us instead of 15.5 (profiles/r05_score_ab.txt).
template <int V>
static __global__ __launch_bounds__(kWmE * kWmG, V == 1 ? 8 : 4) void score_wmean_batch_kernel(
    const float* __restrict__ rews, const float* __restrict__ lp_demo, int N, float rew_xref, float temp, int std_guard,
    float* __restrict__ weights_out, float* __restrict__ rew_mean_out, const float* __restrict__ Y0s, int HNu,
    const float* __restrict__ Ybar_i, float alpha_i, float alpha_bar_i, float alpha_bar_im1, int literal,
    float* __restrict__ Ybar_im1, int lazy, float sigma, float* __restrict__ ybar_keep, ScoreBatch sb) {
  // Which plan and which tile: with a multiple of 8 plans, plan k's tiles all go to XCD k mod 8 (workgroups land on the
  // XCDs round-robin in linear order: the m-th workgroup of XCD c is linear index c + 8 m), so every line of a plan's
  // normals is fetched by ONE L2; otherwise each plan's row of workgroups gets the XCD-aware tile order of the
  // single-plan kernel.  A bijection on (plan, tile) either way; values do not depend on it.
  const int T = gridDim.x, P = gridDim.y;
  long long k = blockIdx.y;
  int tile = xcd_tile(blockIdx.x, T, blockIdx.y * T);
  if ((P & 7) == 0) {
    const int lin = blockIdx.y * T + blockIdx.x, c = lin & 7, m = lin >> 3;
    k = c + 8 * (m / T);
    tile = m % T;
  }
  score_wmean_body<32, false, V>(rews + k * sb.rews, lp_demo ? lp_demo + k * sb.lp : nullptr, N, rew_xref, sb.temps ? sb.temps[k] : temp,
                   std_guard, weights_out + k * sb.weights, rew_mean_out + k * sb.mean, Y0s + k * sb.cand, HNu,
                   Ybar_i + k * sb.ybar_in, alpha_i, alpha_bar_i, alpha_bar_im1, literal, Ybar_im1 + k * sb.ybar_out, lazy,
                   sigma, ybar_keep ? ybar_keep + k * sb.keep : nullptr, tile, tile == 0);
}
// the normals of P plans (one key each) in one launch: blockIdx.y is the plan
struct SweepKeys {
  uint32_t k[32][2];
};
static __global__ __launch_bounds__(256) void noise_batch_kernel(SweepKeys keys, int impl, int N, int HNu, float* __restrict__ eps) {
  const uint64_t size = (uint64_t)N * (uint64_t)HNu;
  noise_fill(keys.k[blockIdx.y][0], keys.k[blockIdx.y][1], impl, size, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x,
             (uint64_t)gridDim.x * blockDim.x, eps + (uint64_t)blockIdx.y * size);
}

// the materialised candidates of P path-integral plans in one launch (blockIdx.y = plan; grid-stride over the thread-items
// of sample_kernel's whole-tensor forms — same counters, same two roundings): Y0s[k] = clip(eps_k * sigma[k] + mu[k], -1, 1)
static __global__ __launch_bounds__(256) void sample_batch_kernel(SweepKeys keys, int impl, int N, int HNu,
                                                            const float* __restrict__ sigma_dev /* [P] */,
                                                            const float* __restrict__ mu, long long mu_stride,
                                                            float* __restrict__ Y0s) {
  const long long plan = blockIdx.y;
  const uint32_t k0 = keys.k[plan][0], k1 = keys.k[plan][1];
  const float sigma = sigma_dev[plan];
  const float* __restrict__ Ybar = mu + plan * mu_stride;
  const uint64_t size = (uint64_t)N * (uint64_t)HNu, half = (size + 1) / 2;
  float* __restrict__ out = Y0s + (uint64_t)plan * size;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t items = impl == 1 ? size : half;
  for (uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < items; tid += stride) {
    uint32_t o0, o1;
    if (impl == 1) {
      threefry2x32(k0, k1, (uint32_t)(tid >> 32), (uint32_t)tid, o0, o1);
      const float y = bits_to_normal(o0 ^ o1) * sigma + Ybar[tid % (uint64_t)HNu];
      out[tid] = fclip(y, -1.0f, 1.0f);
    } else {
      const uint64_t j1 = tid + half;
      threefry2x32(k0, k1, (uint32_t)tid, j1 < size ? (uint32_t)j1 : 0u, o0, o1);
      const float y0 = bits_to_normal(o0) * sigma + Ybar[tid % (uint64_t)HNu];
      out[tid] = fclip(y0, -1.0f, 1.0f);
      if (j1 < size) {
        const float y1 = bits_to_normal(o1) * sigma + Ybar[j1 % (uint64_t)HNu];
        out[j1] = fclip(y1, -1.0f, 1.0f);
      }
    }
  }
}

// The same weighted mean for LARGE N (multi-GPU plans average over all N_total candidates on every rank).  The tile
// kernel above reads 64-byte pieces of rows 3.4 KB apart (1.4 TB/s at N = 8192); here a workgroup owns ONE candidate
// group g and 256 consecutive outputs, so a wavefront reads 256 contiguous bytes of a row per load and the weight is a
// scalar.  The (group, output) partials go through a [64][HNu] scratch and wmean_finish_kernel adds the 64 partials
// of an output in group order and applies the update: the same chains and the same final order as wmean_kernel —
// the same bits.
constexpr int kWmT = 256;  // outputs per workgroup of the row-major variant
static __global__ __launch_bounds__(kWmT) void wmean_partial_kernel(const float* __restrict__ weights,
                                                             const float* __restrict__ Y0s, int N, int HNu,
                                                             float* __restrict__ partial, int lazy, float sigma,
                                                             const float* __restrict__ Ybar_i) {
  const int g = blockIdx.y;
  const int e_raw = blockIdx.x * kWmT + threadIdx.x;
  const int e = e_raw < HNu ? e_raw : HNu - 1;
  const float* __restrict__ col = Y0s + e;
  const float yb = lazy ? Ybar_i[e] : 0.0f;  // lazy plans: Y0s holds normals (wmean_kernel)
  auto val = [&](float x) { return lazy ? fclip(x * sigma + yb, -1.0f, 1.0f) : x; };
  float acc = 0.0f;
  int n = g;
  for (; n + 31 * kWmG < N; n += 32 * kWmG) {
    float y[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) y[k] = col[(size_t)(n + k * kWmG) * HNu];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = ffma(weights[n + k * kWmG], val(y[k]), acc);
  }
  for (; n < N; n += kWmG) acc = ffma(weights[n], val(col[(size_t)n * HNu]), acc);
  if (e_raw < HNu) partial[(size_t)g * HNu + e_raw] = acc;
}
static __global__ __launch_bounds__(64) void wmean_finish_kernel(const float* __restrict__ partial, int HNu,
                                                          const float* __restrict__ Ybar_i, float alpha_i,
                                                          float alpha_bar_i, float alpha_bar_im1, int literal,
                                                          float* __restrict__ Ybar_im1, float* __restrict__ ybar_keep) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= HNu) return;
  if (ybar_keep) ybar_keep[e] = Ybar_i[e];
  float tot = partial[e];
#pragma unroll 8
  for (int k = 1; k < kWmG; ++k) tot = tot + partial[(size_t)k * HNu + e];
  float out = tot;
  if (literal) {
    const float sab = fsqrt(alpha_bar_i);
    float Yi = Ybar_i[e] * sab;
    float t1 = 1.0f / (1.0f - alpha_bar_i);
    float t2 = sab * tot;
    float score = t1 * (-Yi + t2);
    float t3 = (1.0f - alpha_bar_i) * score;
    float Yim1 = (1.0f / fsqrt(alpha_i)) * (Yi + t3);
    out = Yim1 / fsqrt(alpha_bar_im1);
  }
  Ybar_im1[e] = out;
}

// ---- path-integral baselines (mbd/planners/path_integral.py:39-52) ---------------------------------------
// cma-es: s[e] = sqrt(sum_n w_n (Y0s[n][e] - mu_t[e])^2), one thread per output, sequential fma over n
static __global__ __launch_bounds__(64) void cma_spread_kernel(const float* __restrict__ weights,
                                                        const float* __restrict__ Y0s, int N, int HNu,
                                                        const float* __restrict__ mu_t, float* __restrict__ s_out, PiBatch pb) {
  const long long plan = blockIdx.y;
  weights += plan * pb.weights; Y0s += plan * pb.cand; mu_t += plan * pb.mu; s_out += plan * pb.spread;
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= HNu) return;
  const float m = mu_t[e];
  float acc = 0.0f;
#pragma unroll 16
  for (int n = 0; n < N; ++n) {
    float d = Y0s[(size_t)n * HNu + e] - m;
    acc = ffma(weights[n], d * d, acc);
  }
  s_out[e] = fsqrt(acc);
}
// sigma <- max(mean_e(s) * sigma, 1e-3): one wavefront, canonical reduction order
static __global__ __launch_bounds__(64) void cma_sigma_kernel(const float* __restrict__ s, int HNu, float* __restrict__ sigma, PiBatch pb) {
  s += (long long)blockIdx.y * pb.spread; sigma += (long long)blockIdx.y * pb.sigma;
  float part = 0.0f;
  for (int i = threadIdx.x; i < HNu; i += 64) part = part + s[i];
  float sig = (wave_sum(part) / (float)HNu) * (*sigma);
  if (threadIdx.x == 0) *sigma = sig > 1e-3f ? sig : 1e-3f;
}
// cem: indices of the K (<= 10) largest weights, ties towards the higher index (argsort()[::-1][:10])
static __global__ __launch_bounds__(64) void cem_select_kernel(const float* __restrict__ weights, int N, int K,
                                                        int* __restrict__ idx_out, float* __restrict__ wl_global, PiBatch pb) {
  weights += (long long)blockIdx.y * pb.weights; idx_out += (long long)blockIdx.y * pb.idx;
  extern __shared__ __attribute__((aligned(16))) float wl_lds[];
  float* __restrict__ wl = wl_global ? wl_global : wl_lds;  // (a lane only ever touches the indices = lane mod 64)
  const int lane = threadIdx.x;
  for (int i = lane; i < N; i += 64) wl[i] = weights[i];
  for (int k = 0; k < K; ++k) {
    float bv = -1.0f;
    int bi = -1;
    for (int i = lane; i < N; i += 64)
      if (wl[i] >= bv) { bv = wl[i]; bi = i; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float ov = __shfl_xor(bv, off, 64);
      int oi = __shfl_xor(bi, off, 64);
      bool take = ov > bv || (ov == bv && oi > bi);
      bv = take ? ov : bv;
      bi = take ? oi : bi;
    }
    if (lane == 0) idx_out[k] = bi;
    if (bi >= 0 && (bi & 63) == lane) wl[bi] = -2.0f;
  }
}
static __global__ __launch_bounds__(64) void cem_mean_kernel(const int* __restrict__ idx, int K, const float* __restrict__ Y0s,
                                                      int HNu, float* __restrict__ mu_out, PiBatch pb) {
  idx += (long long)blockIdx.y * pb.idx; Y0s += (long long)blockIdx.y * pb.cand; mu_out += (long long)blockIdx.y * pb.out;
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= HNu) return;
  float acc = 0.0f;
  for (int k = 0; k < K; ++k) acc = acc + Y0s[(size_t)idx[k] * HNu + e];
  mu_out[e] = acc / (float)K;
}


}  // namespace mbd
