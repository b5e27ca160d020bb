/*
 * count_ops.cc — TEST INFRASTRUCTURE ONLY (CPU oracle): the op counter SURVEY.md §8(d) prescribes for F_sub(model).
 *
 * Compiles the UNCHANGED physics restatement (mbd_oracle_physics.c) as C++ with `real` replaced by a float wrapper
 * whose arithmetic operators count what they execute: add/sub, mul, fma (2 flops), div, sqrt, and — separately,
 * not flops — compares / min / max / abs.  Only real links and real contacts are visited (no padding lanes, no
 * masked slots): this is the ALGORITHMIC work of one substep, the numerator of bench.py's `valu.algorithmic_frac`.
 * Values are bit-identical to the plain f32 build (the wrapper only counts).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

struct orc_counts { uint64_t add, mul, fma, div, sqrt_, cmp; };
static thread_local orc_counts g_cnt;

/* Round 5: every value also carries its DEPENDENCY DEPTH — the length of the longest chain of dependent operations that
 * produced it, in issue slots of the kernels' own sequences: add / sub / mul / fma / min / max / clip (v_med3) /
 * copysign (v_bfi) 1; a division 7 (the flush-or-clamp of the numerator, v_rcp, and the five dependent FMA / MUL of
 * mbd_math.h's div_core_); a square root 5 (clamp, v_rsq, g = x r, d = x - g g, g + d h); negation and |x| 0 (operand
 * modifiers).  Inputs start at depth 0.  What it does NOT see: a select's dependence on its CONDITION (C control flow) —
 * the arms' own depths dominate everywhere but the stick / limit tests, where the condition is one or two operations
 * deeper than the value it selects; and the exchange of values between links (DPP row shifts: one slot each).  So the
 * figure is a LOWER bound on the latency floor of a substep, and a close one.  orc_depth_substeps reports it. */
enum { D_OP = 1, D_DIV = 7, D_SQRT = 5 };
struct creal {
  float v;
  uint32_t d;
  creal() = default;
  creal(float x) : v(x), d(0) {}
  creal(double x) : v((float)x), d(0) {}
  creal(int x) : v((float)x), d(0) {}
  creal(float x, uint32_t dd) : v(x), d(dd) {}
  explicit operator float() const { return v; }
  explicit operator double() const { return (double)v; }
  explicit operator int() const { return (int)v; }
};
static inline uint32_t dmax(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline creal operator+(creal a, creal b) { ++g_cnt.add; return creal(a.v + b.v, dmax(a.d, b.d) + D_OP); }
static inline creal operator-(creal a, creal b) { ++g_cnt.add; return creal(a.v - b.v, dmax(a.d, b.d) + D_OP); }
static inline creal operator*(creal a, creal b) { ++g_cnt.mul; return creal(a.v * b.v, dmax(a.d, b.d) + D_OP); }
static inline creal operator/(creal a, creal b) { ++g_cnt.div; return creal(a.v / b.v, dmax(a.d, b.d) + D_DIV); }
static inline creal operator-(creal a) { return creal(-a.v, a.d); }  /* a sign flip is an operand modifier, not an op */
static inline bool operator<(creal a, creal b) { ++g_cnt.cmp; return a.v < b.v; }
static inline bool operator>(creal a, creal b) { ++g_cnt.cmp; return a.v > b.v; }
static inline bool operator<=(creal a, creal b) { ++g_cnt.cmp; return a.v <= b.v; }
static inline bool operator>=(creal a, creal b) { ++g_cnt.cmp; return a.v >= b.v; }
static inline bool operator==(creal a, creal b) { ++g_cnt.cmp; return a.v == b.v; }
static inline bool operator!=(creal a, creal b) { ++g_cnt.cmp; return a.v != b.v; }
static inline creal& operator+=(creal& a, creal b) { a = a + b; return a; }

#define ORC_REAL creal
#define ORC_REAL_IS_F32 1
#define ORC_COUNT_OPS 1
#define ORC_CUSTOM_PRIMS 1
static inline creal sp_fma(creal a, creal b, creal c) {
  ++g_cnt.fma;
  return creal(__builtin_fmaf(a.v, b.v, c.v), dmax(dmax(a.d, b.d), c.d) + D_OP);
}
static inline creal sp_sqrt(creal x) { ++g_cnt.sqrt_; return creal(__builtin_sqrtf(x.v), x.d + D_SQRT); }
static inline creal sp_abs(creal x) { return creal(__builtin_fabsf(x.v), x.d); }
static inline creal sp_copysign(creal mag, creal sgn) { return creal(__builtin_copysignf(mag.v, sgn.v), dmax(mag.d, sgn.d) + D_OP); }
static inline creal sp_min(creal a, creal b) { ++g_cnt.cmp; return creal(a.v < b.v ? a.v : b.v, dmax(a.d, b.d) + D_OP); }
static inline creal sp_max(creal a, creal b) { ++g_cnt.cmp; return creal(a.v > b.v ? a.v : b.v, dmax(a.d, b.d) + D_OP); }
static inline creal sp_clip(creal v, creal lo, creal hi) {
  g_cnt.cmp += 2;
  return creal(v.v < lo.v ? lo.v : (v.v > hi.v ? hi.v : v.v), dmax(v.d, dmax(lo.d, hi.d)) + D_OP);
}
extern "C" {
#include "mbd_oracle_physics.c"
}

/* flops of ONE substep of `model` from `state` under `action`, as executed by the restatement */
extern "C" __attribute__((visibility("default"))) void orc_count_substep(const mbd_model_t* m, const float* state,
                                                                         const float* action, float* state_out,
                                                                         uint64_t counts[6]) {
  memset(&g_cnt, 0, sizeof(g_cnt));
  orc_substep(m, state, action, state_out);
  /* (the actuator clip/gear of orc_substep's prologue is once per CONTROL step in the rollout; it is a handful of
   * ops per actuator and is left in) */
  counts[0] = g_cnt.add; counts[1] = g_cnt.mul; counts[2] = g_cnt.fma; counts[3] = g_cnt.div;
  counts[4] = g_cnt.sqrt_; counts[5] = g_cnt.cmp;
}

/* Dependency depth of n_sub consecutive substeps from `state` (depth 0) under `action`: out[s][l] = the largest depth of
 * link l's state (position, orientation, velocities) after substep s + 1, out[s][L] = the largest over all links.  The
 * growth per substep of the last column is the length of the recurrence's critical path: the time one substep of one
 * candidate takes on hardware with unlimited lanes, in dependent issue slots (x ~4 clocks each for a lone wavefront). */
extern "C" __attribute__((visibility("default"))) void orc_depth_substeps(const mbd_model_t* m, const float* state,
                                                                           const float* action, int n_sub, uint32_t* out) {
  const int L = m->n_links;
  xf_t x[MBD_MAX_LINKS]; mo_t xd[MBD_MAX_LINKS];
  load_state(state, L, x, xd);
  real tau_rot[MBD_MAX_LINKS * 3], tau_slide[MBD_MAX_LINKS * 3];
  memset(tau_rot, 0, sizeof(tau_rot)); memset(tau_slide, 0, sizeof(tau_slide));
  for (int a = 0; a < m->n_act; ++a) { /* (env_step's prologue: once per control step, its depth is not the substep's) */
    real u = sp_clip(R(action[a]), R(m->act_lo[a]), R(m->act_hi[a])) * R(m->act_gear[a]);
    u.d = 0;
    int l = m->act_link[a], s = m->act_slot[a];
    if (s < 3) tau_rot[l * 3 + s] = u; else tau_slide[l * 3 + s - 3] = u;
  }
  for (int s = 0; s < n_sub; ++s) {
    substep(m, x, xd, tau_rot, tau_slide);
    uint32_t all = 0;
    for (int l = 0; l < L; ++l) {
      uint32_t d = 0;
      for (int i = 0; i < 3; ++i) d = dmax(d, dmax(x[l].p[i].d, dmax(xd[l].v[i].d, xd[l].w[i].d)));
      for (int i = 0; i < 4; ++i) d = dmax(d, x[l].r[i].d);
      out[(size_t)s * (L + 1) + l] = d;
      all = dmax(all, d);
    }
    out[(size_t)s * (L + 1) + L] = all;
  }
}
