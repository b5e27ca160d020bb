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

struct creal {
  float v;
  creal() = default;
  creal(float x) : v(x) {}
  creal(double x) : v((float)x) {}
  creal(int x) : v((float)x) {}
  explicit operator float() const { return v; }
  explicit operator double() const { return (double)v; }
  explicit operator int() const { return (int)v; }
};
static inline creal operator+(creal a, creal b) { ++g_cnt.add; return creal(a.v + b.v); }
static inline creal operator-(creal a, creal b) { ++g_cnt.add; return creal(a.v - b.v); }
static inline creal operator*(creal a, creal b) { ++g_cnt.mul; return creal(a.v * b.v); }
static inline creal operator/(creal a, creal b) { ++g_cnt.div; return creal(a.v / b.v); }
static inline creal operator-(creal a) { return creal(-a.v); }  /* a sign flip is an operand modifier, not an op */
static inline bool operator<(creal a, creal b) { ++g_cnt.cmp; return a.v < b.v; }
static inline bool operator>(creal a, creal b) { ++g_cnt.cmp; return a.v > b.v; }
static inline bool operator<=(creal a, creal b) { ++g_cnt.cmp; return a.v <= b.v; }
static inline bool operator>=(creal a, creal b) { ++g_cnt.cmp; return a.v >= b.v; }
static inline bool operator==(creal a, creal b) { ++g_cnt.cmp; return a.v == b.v; }
static inline bool operator!=(creal a, creal b) { ++g_cnt.cmp; return a.v != b.v; }
static inline creal& operator+=(creal& a, creal b) { a = a + b; return a; }

#define ORC_REAL creal
#define ORC_COUNT_OPS 1
#define __builtin_fmaf(a, b, c) (++g_cnt.fma, __builtin_fmaf(a, b, c))
#define __builtin_sqrtf(x) (++g_cnt.sqrt_, __builtin_sqrtf(x))
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
