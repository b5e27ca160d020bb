// mbd_math.h — arithmetic primitives of the MI355X rollout/score kernels.
//
// The numerical contract: every primitive fixes its rounding sequence with
// explicit fmaf, IEEE-exact +,-,*,/,sqrt (build flags: -ffp-contract=off, no fast-math,
// -fhip-fp32-correctly-rounded-divide-sqrt) and polynomial kernels for the transcendental functions,
// so that results are reproducible bit-for-bit across wavefronts, GPUs, shard layouts and the CPU
// checker used by the tests.  Nothing here depends on ocml's libm.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MBD_HD __host__ __device__ __forceinline__

namespace mbd {

struct v3 {
  float x, y, z;
};
struct q4 {
  float w, x, y, z;
};

MBD_HD float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MBD_HD float fsqrt(float x) { return __builtin_sqrtf(x); }
MBD_HD float fabs_(float x) { return __builtin_fabsf(x); }
// finite arguments: one v_min_f32 / v_max_f32 on the device (same value as the select)
MBD_HD float fmin_(float a, float b) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_fminf(a, b);
#else
  return a < b ? a : b;
#endif
}
MBD_HD float fmax_(float a, float b) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_fmaxf(a, b);
#else
  return a > b ? a : b;
#endif
}
// clamp to [lo, hi], lo <= hi, finite arguments: one v_med3_f32 on the device (same value as the two selects)
MBD_HD float fclip(float v, float lo, float hi) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_fmed3f(v, lo, hi);
#else
  return v < lo ? lo : (v > hi ? hi : v);
#endif
}

MBD_HD v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
MBD_HD float dot(v3 a, v3 b) { return ffma(a.x, b.x, ffma(a.y, b.y, a.z * b.z)); }
MBD_HD v3 cross(v3 a, v3 b) {
  return v3{ffma(a.y, b.z, -(a.z * b.y)), ffma(a.z, b.x, -(a.x * b.z)), ffma(a.x, b.y, -(a.y * b.x))};
}
MBD_HD v3 add(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MBD_HD v3 sub(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MBD_HD v3 scale(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
MBD_HD v3 neg(v3 a) { return v3{-a.x, -a.y, -a.z}; }
// o + s*a, one fma per component
MBD_HD v3 axpy(float s, v3 a, v3 o) { return v3{ffma(s, a.x, o.x), ffma(s, a.y, o.y), ffma(s, a.z, o.z)}; }
MBD_HD v3 sel3(bool c, v3 a, v3 b) { return v3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }

// rotate v by unit quaternion q: t = 2 (u x v); v + w t + u x t
MBD_HD v3 rot(v3 v, q4 q) {
  v3 u{q.x, q.y, q.z};
  v3 t = cross(u, v);
  t = v3{t.x + t.x, t.y + t.y, t.z + t.z};
  v3 c = cross(u, t);
  return v3{ffma(q.w, t.x, v.x) + c.x, ffma(q.w, t.y, v.y) + c.y, ffma(q.w, t.z, v.z) + c.z};
}
MBD_HD q4 sel4(bool c, q4 a, q4 b) { return q4{c ? a.w : b.w, c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
MBD_HD q4 conj(q4 q) { return q4{q.w, -q.x, -q.y, -q.z}; }
MBD_HD v3 irot(v3 v, q4 q) { return rot(v, conj(q)); }
// R(q)^T (0,0,d): inverse rotation of a vector along z, written out (about half of the general irot)
MBD_HD v3 irot_z(float d, q4 q) {
  float a = q.y * d, b = q.x * d;
  float tx = -(a + a), ty = b + b;
  float cx = q.z * ty, cy = -(q.z * tx), cz = ffma(-q.x, ty, q.y * tx);
  return v3{ffma(q.w, tx, cx), ffma(q.w, ty, cy), d + cz};
}
MBD_HD q4 qmul(q4 a, q4 b) {
  q4 o;
  o.w = ffma(-a.z, b.z, ffma(-a.y, b.y, ffma(-a.x, b.x, a.w * b.w)));
  o.x = ffma(-a.z, b.y, ffma(a.y, b.z, ffma(a.x, b.w, a.w * b.x)));
  o.y = ffma(a.z, b.x, ffma(a.y, b.w, ffma(-a.x, b.z, a.w * b.y)));
  o.z = ffma(a.z, b.w, ffma(-a.y, b.x, ffma(a.x, b.y, a.w * b.z)));
  return o;
}
// 1/sqrt(1+e) by its 4th-order series for |e| <= 0.05 (error < 1e-7), exact beyond
MBD_HD q4 qnormalize(q4 q) {
  float n2 = ffma(q.w, q.w, ffma(q.x, q.x, ffma(q.y, q.y, q.z * q.z)));
  float e = n2 - 1.0f;
  float inv = ffma(ffma(ffma(ffma(0.2734375f, e, -0.3125f), e, 0.375f), e, -0.5f), e, 1.0f);
  if (__builtin_expect(fabs_(e) > 0.05f, 0)) inv = 1.0f / fsqrt(n2);
  return q4{q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
// q + 0.5 (0,th) (x) q, not renormalised
MBD_HD q4 qrotvec_raw(q4 q, v3 th) {
  float hx = 0.5f * th.x, hy = 0.5f * th.y, hz = 0.5f * th.z;
  q4 o;
  o.w = ffma(-hz, q.z, ffma(-hy, q.y, ffma(-hx, q.x, q.w)));
  o.x = ffma(-hz, q.y, ffma(hy, q.z, ffma(hx, q.w, q.x)));
  o.y = ffma(hz, q.x, ffma(hy, q.w, ffma(-hx, q.z, q.y)));
  o.z = ffma(hz, q.w, ffma(-hy, q.x, ffma(hx, q.y, q.z)));
  return o;
}
// normalize(q + 0.5 (0,th) (x) q)
MBD_HD q4 qrotvec(q4 q, v3 th) { return qnormalize(qrotvec_raw(q, th)); }
// The same with the rare exact side of the renormalisation (|n2 - 1| > 0.05: a link turning by more than 0.45 rad in ONE
// substep) handled by the CALLER (round 6).  For a lone wavefront per SIMD the branch costs its compare-to-branch latency,
// ~40 cycles, even when it is never taken (tools/probes/probe_branch.hip).
//   QM = 1  SPECULATIVE: the series unconditionally; the largest |n2 - 1| seen is kept in `worst` (one v_max).  The rollout
//           kernels test it once per CONTROL step and re-run a control step in which it exceeded the bound, from its saved
//           start, with QM = 2
//   QM = 2  both sides computed, the exact one selected where it applies: the values of qnormalize, branch-free
template <int QM>
MBD_HD q4 qnormalize_qm(q4 q, float& worst) {
  if constexpr (QM == 0) return qnormalize(q);
  const float n2 = ffma(q.w, q.w, ffma(q.x, q.x, ffma(q.y, q.y, q.z * q.z)));
  const float e = n2 - 1.0f;
  float inv = ffma(ffma(ffma(ffma(0.2734375f, e, -0.3125f), e, 0.375f), e, -0.5f), e, 1.0f);
  if constexpr (QM == 1) {
    worst = fmax_(worst, fabs_(e));
  } else {
    const float exact = 1.0f / fsqrt(n2);
    inv = fabs_(e) > 0.05f ? exact : inv;
  }
  return q4{q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
template <int QM>
MBD_HD q4 qrotvec_qm(q4 q, v3 th, float& worst) { return qnormalize_qm<QM>(qrotvec_raw(q, th), worst); }
struct axes3 {
  v3 X, Y, Z;
};
MBD_HD axes3 qaxes(q4 q) {
  float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
  float xx = q.x * x2, yy = q.y * y2, zz = q.z * z2;
  float xy = q.x * y2, xz = q.x * z2, yz = q.y * z2;
  float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
  axes3 a;
  a.X = v3{1.0f - (yy + zz), xy + wz, xz - wy};
  a.Y = v3{xy - wz, 1.0f - (xx + zz), yz + wx};
  a.Z = v3{xz + wy, yz - wx, 1.0f - (xx + yy)};
  return a;
}

// ---- pairs: the same primitives on two operands at once (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32:
// one VALU issue slot for both).  Component-wise identical roundings to the scalar versions.
typedef float f2 __attribute__((ext_vector_type(2)));
struct v3x2 {
  f2 x, y, z;
};
struct q4x2 {
  f2 w, x, y, z;
};
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 mk2(float a, float b) { return f2{a, b}; }
__device__ __forceinline__ v3x2 pack3(v3 a, v3 b) { return v3x2{mk2(a.x, b.x), mk2(a.y, b.y), mk2(a.z, b.z)}; }
__device__ __forceinline__ q4x2 pack4(q4 a, q4 b) { return q4x2{mk2(a.w, b.w), mk2(a.x, b.x), mk2(a.y, b.y), mk2(a.z, b.z)}; }
__device__ __forceinline__ v3 lo3(v3x2 a) { return v3{a.x.x, a.y.x, a.z.x}; }
__device__ __forceinline__ v3 hi3(v3x2 a) { return v3{a.x.y, a.y.y, a.z.y}; }
__device__ __forceinline__ q4 lo4(q4x2 a) { return q4{a.w.x, a.x.x, a.y.x, a.z.x}; }
__device__ __forceinline__ v3x2 cross2(v3x2 a, v3x2 b) {
  return v3x2{fma2(a.y, b.z, -(a.z * b.y)), fma2(a.z, b.x, -(a.x * b.z)), fma2(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ v3x2 add2(v3x2 a, v3x2 b) { return v3x2{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3x2 rot2(v3x2 v, q4x2 q) {
  v3x2 u{q.x, q.y, q.z};
  v3x2 t = cross2(u, v);
  t = v3x2{t.x + t.x, t.y + t.y, t.z + t.z};
  v3x2 c = cross2(u, t);
  return v3x2{fma2(q.w, t.x, v.x) + c.x, fma2(q.w, t.y, v.y) + c.y, fma2(q.w, t.z, v.z) + c.z};
}
__device__ __forceinline__ q4x2 qmul2(q4x2 a, q4x2 b) {
  q4x2 o;
  o.w = fma2(-a.z, b.z, fma2(-a.y, b.y, fma2(-a.x, b.x, a.w * b.w)));
  o.x = fma2(-a.z, b.y, fma2(a.y, b.z, fma2(a.x, b.w, a.w * b.x)));
  o.y = fma2(a.z, b.x, fma2(a.y, b.w, fma2(-a.x, b.z, a.w * b.y)));
  o.z = fma2(a.z, b.w, fma2(-a.y, b.x, fma2(a.x, b.y, a.w * b.z)));
  return o;
}
__device__ __forceinline__ v3x2 sub2(v3x2 a, v3x2 b) { return v3x2{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3x2 bcast3(v3 a) { return v3x2{mk2(a.x, a.x), mk2(a.y, a.y), mk2(a.z, a.z)}; }
__device__ __forceinline__ v3x2 scale2(v3x2 a, f2 s) { return v3x2{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f2 dot2(v3x2 a, v3x2 b) { return fma2(a.x, b.x, fma2(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ v3x2 axpy2(f2 s, v3x2 a, v3x2 o) { return v3x2{fma2(s, a.x, o.x), fma2(s, a.y, o.y), fma2(s, a.z, o.z)}; }
struct axes3x2 {
  v3x2 X, Y, Z;
};
__device__ __forceinline__ axes3x2 qaxes2(q4x2 q) {
  f2 x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
  f2 xx = q.x * x2, yy = q.y * y2, zz = q.z * z2;
  f2 xy = q.x * y2, xz = q.x * z2, yz = q.y * z2;
  f2 wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
  const f2 one = mk2(1.0f, 1.0f);
  axes3x2 a;
  a.X = v3x2{one - (yy + zz), xy + wz, xz - wy};
  a.Y = v3x2{xy - wz, one - (xx + zz), yz + wx};
  a.Z = v3x2{xz + wy, yz - wx, one - (xx + yy)};
  return a;
}

// atan2 with a = min/max in [0,1], atan(a) = a P(a^2) (A&S 4.4.49), octant fix-ups; branch-free
MBD_HD float atan2_(float y, float x) {
  float ax = fabs_(x), ay = fabs_(y);
  float mx = fmax_(ax, ay), mn = fmin_(ax, ay);
  float a = mx == 0.0f ? 0.0f : mn / mx;
  float s = a * a;
  float p = 0.0028662257f;
  p = ffma(p, s, -0.0161657367f);
  p = ffma(p, s, 0.0429096138f);
  p = ffma(p, s, -0.0752896400f);
  p = ffma(p, s, 0.1065626393f);
  p = ffma(p, s, -0.1420889944f);
  p = ffma(p, s, 0.1999355085f);
  p = ffma(p, s, -0.3333314528f);
  float r = ffma(p * s, a, a);
  r = ay > ax ? 1.57079632679489661923f - r : r;
  r = x < 0.0f ? 3.14159265358979323846f - r : r;
  return y < 0.0f ? -r : r;
}
// ---- the solver's division -------------------------------------------------------------------------------
// numerators below 1e-28 are flushed to zero (or clamped there when non-negative by construction); with that, and
// denominators in [1e-20, 1e10], the sequence below rounds exactly like IEEE division:
//   r = rcp(d) + one Newton step  is the CORRECTLY ROUNDED reciprocal for every float32 in [1e-20, 1e20] (exhaustive,
//                                 tools/probes/probe_rcp.hip);
//   q = n r;  e = n - d q (fma);  q' = q + e r (fma)  is then the correctly rounded quotient (Markstein's theorem: a
//                                 correctly rounded reciprocal and a faithful q need ONE residual step; checked on
//                                 2.6e10 random pairs with full random mantissas and on the hard denominators —
//                                 mantissa all ones / all zeros — by tools/probes/probe_short.hip, 0 mismatches).
// It is the hardware expansion of '/' minus v_div_scale / v_div_fixup (which only act on extreme exponents) and minus
// its second residual step: 6 VALU for one quotient, as many for a pair (packed FMAs).
// 1.0f / d for d in [1e-20, 1e20]: the reciprocal itself — 3 VALU, same value as div_(1.0f, d)
__device__ __forceinline__ float rcp_exact(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = ffma(-d, r, 1.0f);
  return ffma(e, r, r);
}
__device__ __forceinline__ float div_core_(float n, float d) {
  float r = rcp_exact(d);
  float q = n * r;
  float e = ffma(-d, q, n);
  return ffma(e, r, q);
}
__device__ __forceinline__ f2 div2_core_(f2 n, f2 d) {
  f2 r = mk2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
  f2 e = fma2(-d, r, mk2(1.0f, 1.0f));
  r = fma2(e, r, r);
  f2 q = n * r;
  e = fma2(-d, q, n);
  return fma2(e, r, q);
}
__device__ __forceinline__ float div_(float n, float d) { return div_core_(fabs_(n) < 1e-28f ? 0.0f : n, d); }
// numerators that are non-negative by construction: clamped from below at 1e-28 (one v_max) instead of flushed
// (a compare + a select: three issue slots for a lone wave)
__device__ __forceinline__ float div_pos_(float n, float d) { return div_core_(fmax_(n, 1e-28f), d); }
__device__ __forceinline__ f2 div2_pos_(f2 n, f2 d) { return div2_core_(mk2(fmax_(n.x, 1e-28f), fmax_(n.y, 1e-28f)), d); }
// (signed numerator in the low half, non-negative one in the high half)
__device__ __forceinline__ f2 div2_sp_(f2 n, f2 d) {
  return div2_core_(mk2(fabs_(n.x) < 1e-28f ? 0.0f : n.x, fmax_(n.y, 1e-28f)), d);
}
__device__ __forceinline__ f2 div2_(f2 n, f2 d) {
  return div2_core_(mk2(fabs_(n.x) < 1e-28f ? 0.0f : n.x, fabs_(n.y) < 1e-28f ? 0.0f : n.y), d);
}

// ---- the solver's square root -----------------------------------------------------------------------------
// the argument is clamped from below at 1e-30 (one v_max); then the reciprocal square root and ONE correction:
// g = x rsq(x), s = g + (x - g g)(rsq(x) / 2) — bit-identical to the correctly rounded sqrtf on every float32 in
// [1e-30, FLT_MAX] (exhaustive: tools/probes/probe_short.hip; so is the longer form with a refined g,
// tools/probes/probe_sqrt.hip).  6 VALU instead of the 16 of the compiler's expansion, which also scales denormals.
__device__ __forceinline__ float sqrt_floor(float x) {
  x = fmax_(x, 1e-30f);
  float r = __builtin_amdgcn_rsqf(x);
  float g = x * r, h = 0.5f * r;
  float d = ffma(-g, g, x);
  return ffma(d, h, g);
}

// two packed divisions as interleaved chains (same arithmetic as two div2_pos_ calls)
__device__ __forceinline__ void div2x2_(f2 na, f2 da, f2 nb, f2 db, f2& qa_out, f2& qb_out) {
  na = mk2(fmax_(na.x, 1e-28f), fmax_(na.y, 1e-28f));  // (all four are squared lengths)
  nb = mk2(fmax_(nb.x, 1e-28f), fmax_(nb.y, 1e-28f));
  f2 ra = mk2(__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y));
  f2 rb = mk2(__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y));
  const f2 one = mk2(1.0f, 1.0f);
  f2 ea = fma2(-da, ra, one), eb = fma2(-db, rb, one);
  ra = fma2(ea, ra, ra); rb = fma2(eb, rb, rb);
  f2 qa = na * ra, qb = nb * rb;
  ea = fma2(-da, qa, na); eb = fma2(-db, qb, nb);
  qa_out = fma2(ea, ra, qa); qb_out = fma2(eb, rb, qb);
}

// the same with SIGNED numerators in the first pair (flushed to zero below 1e-28 like div_) and non-negative ones in the
// second (clamped like div_pos_): two div2_sp_-style quotients per half — the two contacts of a link in stage (6)
__device__ __forceinline__ void div2x2_sp_(f2 na, f2 da, f2 nb, f2 db, f2& qa_out, f2& qb_out) {
  na = mk2(fabs_(na.x) < 1e-28f ? 0.0f : na.x, fabs_(na.y) < 1e-28f ? 0.0f : na.y);
  nb = mk2(fmax_(nb.x, 1e-28f), fmax_(nb.y, 1e-28f));
  f2 ra = mk2(__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y));
  f2 rb = mk2(__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y));
  const f2 one = mk2(1.0f, 1.0f);
  f2 ea = fma2(-da, ra, one), eb = fma2(-db, rb, one);
  ra = fma2(ea, ra, ra); rb = fma2(eb, rb, rb);
  f2 qa = na * ra, qb = nb * rb;
  ea = fma2(-da, qa, na); eb = fma2(-db, qb, nb);
  qa_out = fma2(ea, ra, qa); qb_out = fma2(eb, rb, qb);
}

// angle of the near-unit vector (c, s) in (-pi, pi], division-free: asin of min(|s|,|c|) + octant fix-ups.  The
// result takes the SIGN BIT of s (one v_bfi instead of a compare and a select; a VALU compare holds a lone
// wavefront's issue port for two slots).  The reflection about pi/2 stays a compare: the same trick there
// (fma(copysign(1, c), r, pi/2 - copysign(pi/2, c))) measured 0.9 % SLOWER on the humanoid kernel.
MBD_HD float angle_unit(float s, float c) {
  float as = fabs_(s), ac = fabs_(c);
  bool swap = as > ac;
  float u = fmin_(as, ac);
  float z = u * u;
  float p = 0.11199134588241577f;
  p = ffma(p, z, -0.09445883333683014f);
  p = ffma(p, z, 0.07875244319438934f);
  p = ffma(p, z, 0.015578965656459332f);
  p = ffma(p, z, 0.04668578505516052f);
  p = ffma(p, z, 0.07486556470394135f);
  p = ffma(p, z, 0.16666975617408752f);
  float r = ffma(p * z, u, u);
  r = swap ? 1.57079632679489661923f - r : r;
  r = c < 0.0f ? 3.14159265358979323846f - r : r;
  return __builtin_copysignf(r, s);
}
// the same for c >= 0 (the middle Euler angle: c = cos b is a square root): no reflection about pi/2
MBD_HD float angle_unit_cpos(float s, float c) {
  float as = fabs_(s);
  bool swap = as > c;
  float u = fmin_(as, c);
  float z = u * u;
  float p = 0.11199134588241577f;
  p = ffma(p, z, -0.09445883333683014f);
  p = ffma(p, z, 0.07875244319438934f);
  p = ffma(p, z, 0.015578965656459332f);
  p = ffma(p, z, 0.04668578505516052f);
  p = ffma(p, z, 0.07486556470394135f);
  p = ffma(p, z, 0.16666975617408752f);
  float r = ffma(p * z, u, u);
  r = swap ? 1.57079632679489661923f - r : r;
  return __builtin_copysignf(r, s);
}
// two angle_unit() evaluations at once (the polynomial runs on packed pairs)
__device__ __forceinline__ f2 angle_unit2(f2 s, f2 c) {
  f2 as = __builtin_elementwise_abs(s), ac = __builtin_elementwise_abs(c);
  bool sw0 = as.x > ac.x, sw1 = as.y > ac.y;
  f2 u = mk2(fmin_(as.x, ac.x), fmin_(as.y, ac.y));
  f2 z = u * u;
  f2 p = mk2(0.11199134588241577f, 0.11199134588241577f);
  p = fma2(p, z, mk2(-0.09445883333683014f, -0.09445883333683014f));
  p = fma2(p, z, mk2(0.07875244319438934f, 0.07875244319438934f));
  p = fma2(p, z, mk2(0.015578965656459332f, 0.015578965656459332f));
  p = fma2(p, z, mk2(0.04668578505516052f, 0.04668578505516052f));
  p = fma2(p, z, mk2(0.07486556470394135f, 0.07486556470394135f));
  p = fma2(p, z, mk2(0.16666975617408752f, 0.16666975617408752f));
  f2 r = fma2(p * z, u, u);
  float r0 = r.x, r1 = r.y;
  r0 = sw0 ? 1.57079632679489661923f - r0 : r0;
  r1 = sw1 ? 1.57079632679489661923f - r1 : r1;
  r0 = c.x < 0.0f ? 3.14159265358979323846f - r0 : r0;
  r1 = c.y < 0.0f ? 3.14159265358979323846f - r1 : r1;
  return mk2(__builtin_copysignf(r0, s.x), __builtin_copysignf(r1, s.y));
}
MBD_HD void sincos_(float x, float* s_out, float* c_out) {
  float k = __builtin_rintf(x * 0.63661977236758134308f);
  float r = ffma(-k, 1.5703125f, x);
  r = ffma(-k, 4.837512969970703125e-4f, r);
  r = ffma(-k, 7.54978995489188216e-8f, r);
  float z = r * r;
  float ps = ffma(ffma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  float sn = ffma(ps * z, r, r);
  float pc = ffma(ffma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  float cs = ffma(pc * z, z, ffma(-0.5f, z, 1.0f));
  int q = (int)k & 3;
  float s = (q & 1) ? cs : sn, c = (q & 1) ? sn : cs;
  s = (q == 2 || q == 3) ? -s : s;
  c = (q == 1 || q == 2) ? -c : c;
  *s_out = s;
  *c_out = c;
}
MBD_HD float exp_(float x) {
  float k = __builtin_rintf(x * 1.44269504088896341f);
  float r = ffma(-k, 0.693359375f, x);
  r = ffma(-k, -2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = ffma(p, r, 1.3981999507e-3f);
  p = ffma(p, r, 8.3334519073e-3f);
  p = ffma(p, r, 4.1665795894e-2f);
  p = ffma(p, r, 1.6666665459e-1f);
  p = ffma(p, r, 5.0000001201e-1f);
  float e = ffma(p * r, r, r) + 1.0f;
  float v = __builtin_ldexpf(e, (int)k);
  v = x < -87.0f ? 0.0f : v;
  return x > 88.7f ? __builtin_inff() : v;
}
MBD_HD float log_(float x) {
  int e;
  float m = __builtin_frexpf(x, &e);
  bool lo = m < 0.707106781186547524f;
  e = lo ? e - 1 : e;
  m = lo ? m + m : m;
  float f = m - 1.0f;
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = ffma(p, f, -1.1514610310e-1f);
  p = ffma(p, f, 1.1676998740e-1f);
  p = ffma(p, f, -1.2420140846e-1f);
  p = ffma(p, f, 1.4249322787e-1f);
  p = ffma(p, f, -1.6668057665e-1f);
  p = ffma(p, f, 2.0000714765e-1f);
  p = ffma(p, f, -2.4999993993e-1f);
  p = ffma(p, f, 3.3333331174e-1f);
  float y = (p * f) * z;
  float fe = (float)e;
  y = ffma(fe, -2.12194440e-4f, y);
  y = ffma(-0.5f, z, y);
  float r = f + y;
  return ffma(fe, 0.693359375f, r);
}
MBD_HD float log1p_(float t) {  // t in (-1, 0]
  float u = 1.0f + t;
  if (u == 1.0f) return t;
  if (u <= 0.0f) return -__builtin_inff();
  return log_(u) * (t / (u - 1.0f));
}
// XLA's f32 ErfInv polynomial (Giles)
MBD_HD float erfinv_(float x) {
  float w = -log1p_(-(x * x));
  bool small = w < 5.0f;
  float ws = w - 2.5f;
  float wl = fsqrt(w) - 3.0f;
  float ww = small ? ws : wl;
  float p = small ? 2.81022636e-08f : -0.000200214257f;
  p = ffma(p, ww, small ? 3.43273939e-07f : 0.000100950558f);
  p = ffma(p, ww, small ? -3.5233877e-06f : 0.00134934322f);
  p = ffma(p, ww, small ? -4.39150654e-06f : -0.00367342844f);
  p = ffma(p, ww, small ? 0.00021858087f : 0.00573950773f);
  p = ffma(p, ww, small ? -0.00125372503f : -0.0076224613f);
  p = ffma(p, ww, small ? -0.00417768164f : 0.00943887047f);
  p = ffma(p, ww, small ? 0.246640727f : 1.00167406f);
  p = ffma(p, ww, small ? 1.50140941f : 2.83297682f);
  if (fabs_(x) == 1.0f) return x * __builtin_inff();
  return p * x;
}

// ---- threefry2x32-20 (JAX PRNG) ---------------------------------------------------------------------
MBD_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
MBD_HD void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + k0, x1 = c1 + k1;
#define MBD_TF_ROUND(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  MBD_TF_ROUND(13) MBD_TF_ROUND(15) MBD_TF_ROUND(26) MBD_TF_ROUND(6)
  x0 += k1; x1 += k2 + 1u;
  MBD_TF_ROUND(17) MBD_TF_ROUND(29) MBD_TF_ROUND(16) MBD_TF_ROUND(24)
  x0 += k2; x1 += k0 + 2u;
  MBD_TF_ROUND(13) MBD_TF_ROUND(15) MBD_TF_ROUND(26) MBD_TF_ROUND(6)
  x0 += k0; x1 += k1 + 3u;
  MBD_TF_ROUND(17) MBD_TF_ROUND(29) MBD_TF_ROUND(16) MBD_TF_ROUND(24)
  x0 += k1; x1 += k2 + 4u;
  MBD_TF_ROUND(13) MBD_TF_ROUND(15) MBD_TF_ROUND(26) MBD_TF_ROUND(6)
  x0 += k2; x1 += k0 + 5u;
#undef MBD_TF_ROUND
  o0 = x0;
  o1 = x1;
}
// jax.random.uniform's bit trick
MBD_HD float bits_to_uniform(uint32_t bits, float minval, float maxval) {
  uint32_t fb = (bits >> 9) | 0x3F800000u;
  float f = __builtin_bit_cast(float, fb) - 1.0f;
  float v = f * (maxval - minval) + minval;
  return v > minval ? v : minval;
}
// jax.random.normal from 32 random bits: sqrt(2) * erfinv(uniform(nextafter(-1,0), 1))
MBD_HD float bits_to_normal(uint32_t bits) {
  const float lo = -0.99999994f;  // nextafterf(-1, 0)
  float u = bits_to_uniform(bits, lo, 1.0f);
  return 1.41421356237309504880f * erfinv_(u);
}
// random bits for flat element j of `size` elements (legacy = jax_threefry_partitionable False)
MBD_HD uint32_t random_bits32(uint32_t k0, uint32_t k1, int impl, uint64_t j, uint64_t size) {
  uint32_t o0, o1;
  if (impl == 1) {
    threefry2x32(k0, k1, (uint32_t)(j >> 32), (uint32_t)j, o0, o1);
    return o0 ^ o1;
  }
  uint64_t half = (size + 1) / 2;
  if (j < half) {
    uint64_t c1 = j + half;
    threefry2x32(k0, k1, (uint32_t)j, c1 < size ? (uint32_t)c1 : 0u, o0, o1);
    return o0;
  }
  threefry2x32(k0, k1, (uint32_t)(j - half), (uint32_t)j, o0, o1);
  return o1;
}

}  // namespace mbd
