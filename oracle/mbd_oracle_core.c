/*
 * mbd_oracle_core.c — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into or imported by the
 * product (libmbd_hip.so / mbd_hip); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.
 *
 * A plain-C restatement of the algorithm layer of the reference's hot path:
 *   - the noise schedule, sampling, reward standardisation, demo blend, softmax, weighted mean and
 *     score update of mbd/planners/mbd_planner.py:84-135 (functions orc_schedule, orc_sample,
 *     orc_score_update),
 *   - the in-tree car2d environment mbd/envs/car2d.py:10-32,77-102 (orc_car2d_*),
 *   - the pieces of JAX the path calls: jax.random.{PRNGKey,split,uniform,normal} over threefry2x32
 *     and XLA's f32 ErfInv (third-party, NOT in /root/reference: jax/jaxlib are un-pinned in the
 *     reference's setup.py:20 and absent from this container; restated from the published algorithms:
 *     Salmon et al. "Parallel random numbers: as easy as 1, 2, 3" (Random123, Threefry-2x32-20) and
 *     M. Giles "Approximating the erfinv function" (single-precision polynomial)).
 *
 * PINNING STATUS
 *   - threefry2x32: pinned by the three Random123 known-answer vectors (tests/test_oracle_prng.py).
 *   - schedule: pinned by the f32 known answers of SURVEY.md §8(a) row A0.
 *   - car2d: pinned by closed-form cases (tests/test_oracle_car2d.py).
 *   - score update, demo blend, softmax, weighted mean, the loop's key chain, car2d, the path-integral update rules: since
 *     round 4 also pinned by the reference's OWN source executed under a numpy stand-in for jax (tools/make_ref_golden.py,
 *     tests/golden/ref_*.npz, tests/test_ref_golden.py: BASELINE config 1 at all 49 steps) — at float32 round-off, since
 *     numpy is not XLA.
 *   - split / random_bits layouts, uniform bit trick, normal, ErfInv coefficients: restated from the
 *     JAX/XLA sources and pinned by outputs of the real JAX printed in its public documentation (key
 *     splits and normal/uniform draws for PRNGKey(0)/PRNGKey(42) in the legacy layout and key(42) in the
 *     partitionable layout; tests/test_oracle_prng.py lists values and sources) — reproduced bit for bit.
 *     tools/dump_golden.py produces longer vectors under a real jax install.
 *
 * All arithmetic is float32 (the reference keeps jax_enable_x64 off, mbd_planner.py:13-14) except the
 * PRNG (uint32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#undef ORC_REAL
#define ORC_REAL float /* this file is float32-only */
#include "spec_math.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------ */
/* threefry2x32, 20 rounds  (jax/_src/prng.py: threefry2x32_p; Random123 threefry2x32_R(20, ...))    */
/* ------------------------------------------------------------------------------------------------ */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

ORC_API void orc_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* o0,
                              uint32_t* o1) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int s = 0; s < 5; ++s) {
    const int* r = R[s & 1];
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, r[j]);
      x1 ^= x0;
    }
    x0 += ks[(s + 1) % 3];
    x1 += ks[(s + 2) % 3] + (uint32_t)(s + 1);
  }
  *o0 = x0;
  *o1 = x1;
}

/* jax.random.PRNGKey(seed): [seed >> 32, seed & 0xffffffff]  (mbd_planner.py:40) */
ORC_API void orc_prng_key(uint64_t seed, uint32_t key[2]) {
  key[0] = (uint32_t)(seed >> 32);
  key[1] = (uint32_t)(seed & 0xffffffffu);
}

/* 32 random bits for flat element j of an array with `size` elements.
 * impl 0 (legacy, jax_threefry_partitionable=False): counts = iota(size) (+1 zero pad if odd), first
 *   half -> x0, second half -> x1, output = concat(out0, out1)[:size].
 * impl 1 (partitionable): (b1,b2) = threefry(key, hi(j), lo(j)); bits = b1 ^ b2. */
ORC_API uint32_t orc_random_bits32(const uint32_t key[2], int impl, uint64_t j, uint64_t size) {
  uint32_t o0, o1;
  if (impl == 1) {
    orc_threefry2x32(key[0], key[1], (uint32_t)(j >> 32), (uint32_t)j, &o0, &o1);
    return o0 ^ o1;
  }
  uint64_t half = (size + 1) / 2; /* padded size / 2 */
  if (j < half) {
    uint64_t c1 = j + half; /* the padded element (index == size) has count 0 */
    uint32_t cc1 = (c1 < size) ? (uint32_t)c1 : 0u;
    orc_threefry2x32(key[0], key[1], (uint32_t)j, cc1, &o0, &o1);
    return o0;
  }
  orc_threefry2x32(key[0], key[1], (uint32_t)(j - half), (uint32_t)j, &o0, &o1);
  return o1;
}

/* jax.random.split(key, num) -> keys[num][2]  (mbd_planner.py:79,103,150; humanoidrun.py:21) */
ORC_API void orc_prng_split(const uint32_t key[2], int num, int impl, uint32_t* keys) {
  if (impl == 1) {
    for (int j = 0; j < num; ++j)
      orc_threefry2x32(key[0], key[1], 0u, (uint32_t)j, &keys[2 * j], &keys[2 * j + 1]);
    return;
  }
  /* legacy: threefry_2x32(key, iota(2*num)) reshaped (num, 2) */
  for (int e = 0; e < 2 * num; ++e) keys[e] = orc_random_bits32(key, 0, (uint64_t)e, (uint64_t)(2 * num));
}

/* jax.random.uniform f32: bits>>9 | 0x3F800000 -> f32 - 1 -> *(max-min)+min -> max(min, .) */
static inline float bits_to_uniform(uint32_t bits, float minval, float maxval) {
  uint32_t fb = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &fb, 4);
  f = f - 1.0f;
  float v = f * (maxval - minval) + minval;
  return v > minval ? v : minval; /* lax.max(minval, v) */
}

ORC_API void orc_uniform(const uint32_t key[2], int impl, uint64_t size, float minval, float maxval,
                         float* out) {
  for (uint64_t j = 0; j < size; ++j)
    out[j] = bits_to_uniform(orc_random_bits32(key, impl, j, size), minval, maxval);
}

/* XLA f32 ErfInv (Giles' single-precision polynomial), as lowered for lax.erf_inv */
ORC_API float orc_erfinv_f32(float x) {
  float w = -sp_log1p_f32(-(x * x));
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = __builtin_fmaf(p, w, 3.43273939e-07f);
    p = __builtin_fmaf(p, w, -3.5233877e-06f);
    p = __builtin_fmaf(p, w, -4.39150654e-06f);
    p = __builtin_fmaf(p, w, 0.00021858087f);
    p = __builtin_fmaf(p, w, -0.00125372503f);
    p = __builtin_fmaf(p, w, -0.00417768164f);
    p = __builtin_fmaf(p, w, 0.246640727f);
    p = __builtin_fmaf(p, w, 1.50140941f);
  } else {
    w = __builtin_sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = __builtin_fmaf(p, w, 0.000100950558f);
    p = __builtin_fmaf(p, w, 0.00134934322f);
    p = __builtin_fmaf(p, w, -0.00367342844f);
    p = __builtin_fmaf(p, w, 0.00573950773f);
    p = __builtin_fmaf(p, w, -0.0076224613f);
    p = __builtin_fmaf(p, w, 0.00943887047f);
    p = __builtin_fmaf(p, w, 1.00167406f);
    p = __builtin_fmaf(p, w, 2.83297682f);
  }
  if (fabsf(x) == 1.0f) return x * INFINITY;
  return p * x;
}

/* threads of the OpenMP build (bench.py's cpu_baseline: the host's usable CPUs — cgroup quota, affinity — not its
 * hardware thread count); returns the count in force (1 for the serial builds) */
#ifdef _OPENMP
#include <omp.h>
#endif
ORC_API int orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* jax.random.normal f32: sqrt(2) * erf_inv(uniform(nextafter(-1,0), 1)) */
ORC_API void orc_normal(const uint32_t key[2], int impl, uint64_t begin, uint64_t count,
                        uint64_t size, float* out) {
  const float lo = nextafterf(-1.0f, 0.0f);
  const float sqrt2 = (float)1.4142135623730951;
  for (uint64_t j = 0; j < count; ++j) {
    float u = bits_to_uniform(orc_random_bits32(key, impl, begin + j, size), lo, 1.0f);
    out[j] = sqrt2 * orc_erfinv_f32(u);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* noise schedule  (mbd_planner.py:84-87)                                                            */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_schedule(float beta0, float betaT, int Nd, float* alphas, float* alphas_bar,
                          float* sigmas) {
  /* jnp.linspace(beta0, betaT, Nd) in f32 (jax/_src/numpy/lax_numpy.py): t = iota/div,
   * out = start*(1-t) + stop*t for the first Nd-1 points, endpoint appended exactly. */
  float cp = 1.0f;
  for (int i = 0; i < Nd; ++i) {
    float t = (Nd > 1) ? (float)i / (float)(Nd - 1) : 0.0f;
    float beta = (i == Nd - 1 && Nd > 1) ? betaT : beta0 * (1.0f - t) + betaT * t;
    float a = 1.0f - beta;
    cp = cp * a; /* jnp.cumprod, sequential in f32 */
    alphas[i] = a;
    alphas_bar[i] = cp;
    sigmas[i] = sqrtf(1.0f - cp);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* sampling  (mbd_planner.py:103-106): Y0s = clip(eps * sigma_i + Ybar_i, -1, 1)                     */
/* rows [begin, begin+count) of the global [N][H*Nu] noise tensor                                    */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_sample(const uint32_t key_sample[2], int impl, int N, int HNu, int begin, int count,
                        float sigma, const float* Ybar, float* Y0s /* [count][HNu] */,
                        float* eps_opt /* [count][HNu] or NULL */) {
  const uint64_t size = (uint64_t)N * (uint64_t)HNu;
  /* counter-based noise: every element depends on (key, flat index) only, so the rows are independent — the OpenMP
   * build (the all-core cpu_baseline of bench.py) spreads them over the cores; same values in any order */
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    float* row = (float*)malloc(sizeof(float) * (size_t)HNu);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int n = 0; n < count; ++n) {
      orc_normal(key_sample, impl, (uint64_t)(begin + n) * (uint64_t)HNu, (uint64_t)HNu, size, row);
      for (int e = 0; e < HNu; ++e) {
        float y = row[e] * sigma + Ybar[e];
        y = y < -1.0f ? -1.0f : (y > 1.0f ? 1.0f : y);
        Y0s[(size_t)n * HNu + e] = y;
        if (eps_opt) eps_opt[(size_t)n * HNu + e] = row[e];
      }
    }
    free(row);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* score estimate  (mbd_planner.py:110-135)                                                          */
/*   rews[N] (= mean_H rewss), optional demo log-densities lp[N], Y0s[N][HNu], Ybar_i[HNu]           */
/*   -> weights[N], Ybar_im1[HNu], returns rews.mean()                                                */
/* reductions follow the fixed orders below (XLA uses its own tree reductions: last-bit differences  */
/* expected).                                                                          */
/* ------------------------------------------------------------------------------------------------ */
/* canonical reductions (spec_math.h).
 *   sum_f32:  one wavefront — 64 strided partials + xor-butterfly (cma-es sigma, orc_sp_sum)
 *   sumB / sumsq_devB:  one 1024-thread workgroup, the order of the score kernel — partial t accumulates
 *     i = t, t+1024, ... in increasing i; each wavefront (64 consecutive t) runs the xor-butterfly; the 16
 *     wavefront sums are then added sequentially in wavefront order
 *   wsum64: the weighted mean over candidates — 64 partials, partial g is a sequential fma over
 *     n = g, g+64, ... in increasing n; the 64 partials are then added sequentially in g               */
static float sum_f32(const float* x, int n) {
  float part[64];
  for (int j = 0; j < 64; ++j) part[j] = 0.0f;
  for (int i = 0; i < n; ++i) part[i & 63] = part[i & 63] + x[i];
  return sp_reduce_sum64(part);
}
static float reduceB(const float* part /* [1024] */) {
  float s = sp_reduce_sum64(part);
  for (int w = 1; w < 16; ++w) s = s + sp_reduce_sum64(part + 64 * w);
  return s;
}
static float sumB(const float* x, int n) {
  float part[1024];
  for (int j = 0; j < 1024; ++j) part[j] = 0.0f;
  for (int i = 0; i < n; ++i) part[i & 1023] = part[i & 1023] + x[i];
  return reduceB(part);
}
static float sumsq_devB(const float* x, int n, float mean) {
  float part[1024];
  for (int j = 0; j < 1024; ++j) part[j] = 0.0f;
  for (int i = 0; i < n; ++i) {
    float d = x[i] - mean;
    part[i & 1023] = __builtin_fmaf(d, d, part[i & 1023]);
  }
  return reduceB(part);
}
static float wsum64(const float* weights, const float* col /* stride HNu */, int N, int HNu) {
  float part[64];
  for (int g = 0; g < 64; ++g) part[g] = 0.0f;
  for (int n = 0; n < N; ++n) part[n & 63] = __builtin_fmaf(weights[n], col[(size_t)n * HNu], part[n & 63]);
  float s = part[0];
  for (int g = 1; g < 64; ++g) s = s + part[g];
  return s;
}
static float max_f32(const float* x, int n) {
  float m = x[0];
  for (int i = 1; i < n; ++i) m = x[i] > m ? x[i] : m;
  return m;
}

ORC_API float orc_score_update(int N, int HNu, const float* rews, const float* lp_demo /* or NULL */,
                               float rew_xref, float temp, const float* Y0s, const float* Ybar_i,
                               float alpha_i, float alpha_bar_i, float alpha_bar_im1, int literal,
                               float* weights /* [N] */, float* Ybar_im1 /* [HNu] */) {
  float* logp0 = (float*)malloc(sizeof(float) * (size_t)N);
  float rew_mean = sumB(rews, N) / (float)N;                                   /* :113 */
  float rew_std = __builtin_sqrtf(sumsq_devB(rews, N, rew_mean) / (float)N);   /* :111 (ddof 0) */
  if (rew_std < 1e-4f) rew_std = 1.0f;                                            /* :112 */
  for (int n = 0; n < N; ++n) logp0[n] = ((rews[n] - rew_mean) / rew_std) / temp; /* :114 */
  if (lp_demo) {                                                                  /* :117-125 */
    float mx = max_f32(lp_demo, N);
    for (int n = 0; n < N; ++n) {
      float lpd = ((((lp_demo[n] - mx) + rew_xref) - rew_mean) / rew_std) / temp;
      if (lpd > logp0[n]) logp0[n] = lpd;
    }
    float m = sumB(logp0, N) / (float)N;
    float sd = __builtin_sqrtf(sumsq_devB(logp0, N, m) / (float)N);
    for (int n = 0; n < N; ++n) logp0[n] = ((logp0[n] - m) / sd) / temp; /* no zero-std guard (:125) */
  }
  /* jax.nn.softmax (:127) */
  float mx = max_f32(logp0, N);
  for (int n = 0; n < N; ++n) weights[n] = sp_exp_f32(logp0[n] - mx);
  float den = sumB(weights, N);
  for (int n = 0; n < N; ++n) weights[n] = weights[n] / den;
  /* Ybar = einsum("n,nij->ij") (:128): wsum64 order */
  const float sab = __builtin_sqrtf(alpha_bar_i);
  /* (columns are independent; each keeps its canonical wsum64 order — the OpenMP build spreads them over the cores) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int e = 0; e < HNu; ++e) {
    float Ybar = wsum64(weights, Y0s + e, N, HNu);
    if (literal) { /* :100,130-133 */
      float Yi = Ybar_i[e] * sab;
      float t1 = 1.0f / (1.0f - alpha_bar_i);
      float t2 = sab * Ybar;
      float score = t1 * (-Yi + t2);
      float t3 = (1.0f - alpha_bar_i) * score;
      float Yim1 = (1.0f / __builtin_sqrtf(alpha_i)) * (Yi + t3);
      Ybar_im1[e] = Yim1 / __builtin_sqrtf(alpha_bar_im1);
    } else {
      Ybar_im1[e] = Ybar; /* algebraic identity, SURVEY G7 */
    }
  }
  free(logp0);
  return rew_mean; /* rews.mean() (:135) */
}

/* rews = rewss.mean(axis=-1) (:110): sequential sum over t, one division */
ORC_API void orc_mean_h(const float* rewss, int B, int H, float* rews) {
  for (int b = 0; b < B; ++b) {
    float s = 0.0f;
    for (int t = 0; t < H; ++t) s = s + rewss[(size_t)b * H + t];
    rews[b] = s / (float)H;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* car2d  (mbd/envs/car2d.py)                                                                        */
/* ------------------------------------------------------------------------------------------------ */
static const float CAR_PI = 3.14159274101257324f; /* float32(jnp.pi) */

static void car_dyn(const float x[3], const float u[2], float dx[3]) { /* car2d.py:10-19 */
  float sn, cs;
  sp_sincos(x[2], &sn, &cs);
  dx[0] = u[1] * sn * 3.0f;
  dx[1] = u[1] * cs * 3.0f;
  dx[2] = u[0] * CAR_PI / 3.0f * 2.0f;
}

static void car_rk4(const float x[3], const float u[2], float out[3]) { /* car2d.py:22-27, dt = 0.1 */
  /* dt is a python float: dt/2 and dt/6 are evaluated in float64, then enter the f32 graph */
  const float dt = (float)0.1, dt2 = (float)(0.1 / 2), dt6 = (float)(0.1 / 6);
  float k1[3], k2[3], k3[3], k4[3], t[3];
  car_dyn(x, u, k1);
  for (int i = 0; i < 3; ++i) t[i] = x[i] + dt2 * k1[i];
  car_dyn(t, u, k2);
  for (int i = 0; i < 3; ++i) t[i] = x[i] + dt2 * k2[i];
  car_dyn(t, u, k3);
  for (int i = 0; i < 3; ++i) t[i] = x[i] + dt * k3[i];
  car_dyn(t, u, k4);
  for (int i = 0; i < 3; ++i) out[i] = x[i] + dt6 * (k1[i] + 2.0f * k2[i] + 2.0f * k3[i] + k4[i]);
}

/* obstacle centres, car2d.py:48-63 with r_obs = 0.3 (products evaluated in f32 like jnp.array does) */
static void car_obstacles(float c[11][2]) {
  const float r = 0.3f;
  const float m[11][2] = {{-3, 2}, {-2, 2}, {-1, 2}, {0, 2}, {0, 1}, {0, 0},
                          {0, -1}, {-3, -2}, {-2, -2}, {-1, -2}, {0, -2}};
  for (int i = 0; i < 11; ++i) {
    /* python evaluates -r_obs*3 etc. in float64, then casts the list to f32 */
    c[i][0] = (float)((double)0.3 * (double)m[i][0]);
    c[i][1] = (float)((double)0.3 * (double)m[i][1]);
  }
  (void)r;
}

ORC_API float orc_car2d_reward(const float q[3]) { /* car2d.py:88-93 */
  float dx = q[0] - 0.5f, dy = q[1] - 0.0f;
  float d = sqrtf(dx * dx + dy * dy);
  d = d < 0.0f ? 0.0f : (d > 0.2f ? 0.2f : d);
  float t = d / 0.2f;
  return 1.0f - t * t;
}

ORC_API void orc_car2d_reset(float q[3]) { /* car2d.py:64,73-75 */
  q[0] = -0.5f;
  q[1] = 0.0f;
  q[2] = (float)(3.141592653589793 * 3.0 / 2.0); /* jnp.pi*3/2 evaluated in python float64 */
}

ORC_API float orc_car2d_step(const float q[3], const float action[2], float q_out[3]) { /* :77-86 */
  float c[11][2];
  car_obstacles(c);
  float a[2] = {action[0] < -1.0f ? -1.0f : (action[0] > 1.0f ? 1.0f : action[0]),
                action[1] < -1.0f ? -1.0f : (action[1] > 1.0f ? 1.0f : action[1])};
  float qn[3];
  car_rk4(q, a, qn);
  int collide = 0;
  for (int i = 0; i < 11; ++i) { /* car2d.py:30-32 */
    float dx = qn[0] - c[i][0], dy = qn[1] - c[i][1];
    if (sqrtf(dx * dx + dy * dy) < 0.3f) collide = 1;
  }
  for (int i = 0; i < 3; ++i) q_out[i] = collide ? q[i] : qn[i];
  return orc_car2d_reward(q_out);
}

/* rollout_us for car2d over a batch (utils.py:14-20 under vmap, mbd_planner.py:109) */
ORC_API void orc_car2d_rollout(const float q0[3], const float* us /* [B][H][2] */, int B, int H,
                               float* rewss /* [B][H] */, float* qs /* [B][H][3] or NULL */) {
  for (int b = 0; b < B; ++b) {
    float q[3] = {q0[0], q0[1], q0[2]};
    for (int t = 0; t < H; ++t) {
      float qn[3];
      float r = orc_car2d_step(q, &us[((size_t)b * H + t) * 2], qn);
      memcpy(q, qn, sizeof(q));
      rewss[(size_t)b * H + t] = r;
      if (qs) memcpy(&qs[((size_t)b * H + t) * 3], q, sizeof(q));
    }
  }
}

/* Car2d.eval_xref_logpd (car2d.py:95-102): xs [H][3], xref [H][2] */
ORC_API float orc_car2d_xref_logpd(const float* xs, const float* xref, int H) {
  float acc = 0.0f;
  for (int t = 0; t < H; ++t) {
    float ex = xs[3 * t] - xref[2 * t], ey = xs[3 * t + 1] - xref[2 * t + 1];
    float d = sqrtf(ex * ex + ey * ey);
    d = d < 0.0f ? 0.0f : (d > 0.5f ? 0.5f : d);
    float s = d / 0.5f;
    acc += s * s;
  }
  return 0.0f - acc / (float)H;
}

/* HumanoidTrack.eval_xref_logpd (humanoidtrack.py:98-106): xpos [H][K][3], xref [K][H][3].
 * Order of the mean's sum (the contract; the reference's jnp.mean over the [K, H] array leaves it to XLA): link k's H terms
 * in t order (S_k), then S_0 + S_1 + ... in link order — rows first, the way a row-major [K, H] reduction goes, and the order
 * a rollout can accumulate as its control steps go by (round 6; rounds 1-5 ran one chain over all K H terms: 1e-7 apart). */
ORC_API float orc_track_xref_logpd(const float* xpos, const float* xref, int H, int K) {
  float acc = 0.0f;
  for (int k = 0; k < K; ++k) {
    float sk = 0.0f;
    for (int t = 0; t < H; ++t) {
      const float* a = &xpos[((size_t)t * K + k) * 3];
      const float* b = &xref[((size_t)k * H + t) * 3];
      float ex = a[0] - b[0], ey = a[1] - b[1], ez = a[2] - b[2];
      float d = sqrtf(ex * ex + ey * ey + ez * ez);
      d = d < 0.0f ? 0.0f : (d > 0.5f ? 0.5f : d);
      float s = d / 0.5f;
      sk += s * s;
    }
    acc = k == 0 ? sk : acc + sk;
  }
  return 0.0f - acc / (float)(H * K);
}

/* ---- test hooks for the numerical-contract primitives (tests/test_spec_math.py) ------------------ */
ORC_API float orc_sp_atan2(float y, float x) { return sp_atan2(y, x); }
ORC_API float orc_sp_asin(float v) { return sp_asin(v); }
ORC_API void orc_sp_sincos(float x, float* s, float* c) { sp_sincos(x, s, c); }
ORC_API float orc_sp_exp(float x) { return sp_exp_f32(x); }
ORC_API float orc_sp_log1p(float t) { return sp_log1p_f32(t); }
ORC_API void orc_sp_rot(const float v[3], const float q[4], float o[3]) { sp_rot(v, q, o); }
ORC_API void orc_sp_qmul(const float a[4], const float b[4], float o[4]) { sp_qmul(a, b, o); }
ORC_API float orc_sp_sum(const float* x, int n) { return sum_f32(x, n); }

/* ------------------------------------------------------------------------------------------------ */
/* path-integral baselines  (mbd/planners/path_integral.py:33-52,111-127)                            */
/*   logp0 = (rews - mean)/std/temp WITHOUT the zero-std guard (:123); weights = softmax (:124);     */
/*   method 1 mppi   : mu = sum_n w_n Y0s_n                                       (softmax_update)    */
/*   method 2 cma-es : mu as mppi; sigma = max(mean_e sqrt(sum_n w_n (Y0s-mu_t)^2) * sigma, 1e-3)     */
/*   method 3 cem    : mu = mean of the 10 candidates with the largest weights    (cem_update)        */
/* returns rews.mean(); *sigma_inout is updated for cma-es only.                                     */
/* ------------------------------------------------------------------------------------------------ */
ORC_API float orc_pi_update(int method, int N, int HNu, const float* rews, float temp, const float* Y0s,
                            const float* mu_t, float* sigma_inout, float* weights, float* mu_tm1) {
  float* logp0 = (float*)malloc(sizeof(float) * (size_t)N);
  float rew_mean = sumB(rews, N) / (float)N;
  float rew_std = __builtin_sqrtf(sumsq_devB(rews, N, rew_mean) / (float)N);
  for (int n = 0; n < N; ++n) logp0[n] = ((rews[n] - rew_mean) / rew_std) / temp;
  float mx = max_f32(logp0, N);
  for (int n = 0; n < N; ++n) weights[n] = sp_exp_f32(logp0[n] - mx);
  float den = sumB(weights, N);
  for (int n = 0; n < N; ++n) weights[n] = weights[n] / den;
  if (method == 3) { /* argsort(weights)[::-1][:10]: ties resolved towards the HIGHER index */
    int K = N < 10 ? N : 10;
    char* used = (char*)calloc((size_t)N, 1);
    int idx[10];
    for (int k = 0; k < K; ++k) {
      int best = -1;
      for (int n = 0; n < N; ++n)
        if (!used[n] && (best < 0 || weights[n] >= weights[best])) best = n;
      used[best] = 1;
      idx[k] = best;
    }
    for (int e = 0; e < HNu; ++e) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k) acc = acc + Y0s[(size_t)idx[k] * HNu + e];
      mu_tm1[e] = acc / (float)K;
    }
    free(used);
  } else {
    for (int e = 0; e < HNu; ++e) mu_tm1[e] = wsum64(weights, Y0s + e, N, HNu);
    if (method == 2) {
      float* s = (float*)malloc(sizeof(float) * (size_t)HNu);
      for (int e = 0; e < HNu; ++e) {
        float acc = 0.0f;
        for (int n = 0; n < N; ++n) {
          float d = Y0s[(size_t)n * HNu + e] - mu_t[e];
          acc = __builtin_fmaf(weights[n], d * d, acc);
        }
        s[e] = __builtin_sqrtf(acc);
      }
      float sig = (sum_f32(s, HNu) / (float)HNu) * (*sigma_inout);
      *sigma_inout = sig > 1e-3f ? sig : 1e-3f;
      free(s);
    }
  }
  free(logp0);
  return rew_mean;
}
