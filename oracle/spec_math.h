/*
 * spec_math.h — TEST INFRASTRUCTURE ONLY (CPU oracle): the numerical contract of the hot path.
 *
 * The rollout is chaotic (f32 vs f64 runs of the SAME code differ by 1e-2 in reward after 350
 * contact-rich substeps), so "within 1e-5" between two implementations can
 * only be guaranteed if both round identically.  Every primitive below therefore fixes its rounding
 * sequence: explicit fused multiply-adds (fmaf, exact by definition on both x86 and gfx950),
 * correctly-rounded +,-,*,/,sqrt, no re-association (build with -ffp-contract=off, no fast-math), and
 * the transcendental functions are fixed polynomial kernels instead of libm / ocml calls.
 * The HIP kernels implement the same contract independently (model-based-diffusion_amd/csrc/
 * mbd_math.h); tests compare the two bit-for-bit.
 *
 * With ORC_REAL=double the same expressions are evaluated in f64 (used only to quantify chaos).
 */
#ifndef ORC_SPEC_MATH_H
#define ORC_SPEC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL float
#endif
typedef ORC_REAL real;
#define R(x) ((real)(x))
#ifndef ORC_REAL_IS_F32
#define ORC_REAL_IS_F32 (sizeof(real) == 4)
#endif

/* (ORC_CUSTOM_PRIMS: oracle/count_ops.cc supplies sp_fma / sp_sqrt / sp_abs / sp_copysign on its counting wrapper type —
 * the same values, plus op counts and dependency depth) */
#ifndef ORC_CUSTOM_PRIMS
static inline real sp_fma(real a, real b, real c) {
  return ORC_REAL_IS_F32 ? (real)__builtin_fmaf((float)a, (float)b, (float)c)
                           : (real)__builtin_fma((double)a, (double)b, (double)c);
}
static inline real sp_sqrt(real x) {
  return ORC_REAL_IS_F32 ? (real)__builtin_sqrtf((float)x) : (real)__builtin_sqrt((double)x);
}
#endif
/* the solver's square root (cos of the middle Euler angle, tangential contact speed): the argument is clamped from
 * below at 1e-30 (result >= 1e-15), everything else is the correctly rounded root.  (The clamp lets the kernels
 * use the 8-instruction rsq + FMA sequence, which tools/probes/probe_sqrt.hip shows bit-identical to sqrtf on EVERY
 * float32 in [1e-30, FLT_MAX], instead of the 16-instruction expansion that also covers denormal inputs; a clamp
 * rather than a flush because one v_max is cheaper than a compare and a select.) */
static inline real sp_sqrt_floor(real x) { return sp_sqrt(x < R(1e-30) ? R(1e-30) : x); }
#ifndef ORC_CUSTOM_PRIMS
static inline real sp_abs(real x) { return ORC_REAL_IS_F32 ? (real)__builtin_fabsf((float)x) : (real)__builtin_fabs((double)x); }
/* |mag| with the SIGN BIT of sgn (so -0.0 counts as negative): one bit select on the GPU, no compare */
static inline real sp_copysign(real mag, real sgn) {
  return ORC_REAL_IS_F32 ? (real)__builtin_copysignf((float)mag, (float)sgn)
                           : (real)__builtin_copysign((double)mag, (double)sgn);
}
#endif
#ifndef ORC_CUSTOM_PRIMS
static inline real sp_min(real a, real b) { return a < b ? a : b; }
static inline real sp_max(real a, real b) { return a > b ? a : b; }
static inline real sp_clip(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }
#endif

/* the solver's division: numerators below 1e-28 in magnitude are flushed to zero (physically nothing; it keeps
 * every quotient and residual a normal number, which is what lets the GPU use the bare reciprocal/FMA
 * refinement sequence and still round exactly like this IEEE division — tools/probes/probe_div.hip) */
static inline real sp_div(real n, real d) { return (sp_abs(n) < R(1e-28) ? R(0) : n) / d; }
/* the same for numerators that are non-negative by construction (squared lengths, penetration depths, speed
 * magnitudes): clamped from below at 1e-28 instead of flushed — one max instead of a compare and a select, which
 * costs the kernels three issue slots (a VALU compare alone holds the lone wave's issue port for two).  The
 * quotient of such a floor value is at most 1e-8 and always multiplies something that vanishes with the numerator. */
static inline real sp_div_pos(real n, real d) { return (n < R(1e-28) ? R(1e-28) : n) / d; }

/* ---- vectors ------------------------------------------------------------------------------------- */
/* dot = fma(a0,b0, fma(a1,b1, a2*b2)) */
static inline real sp_dot3(const real a[3], const real b[3]) {
  return sp_fma(a[0], b[0], sp_fma(a[1], b[1], a[2] * b[2]));
}
/* cross_i = fma(a_j, b_k, -(a_k*b_j)) */
static inline void sp_cross3(const real a[3], const real b[3], real o[3]) {
  real x = sp_fma(a[1], b[2], -(a[2] * b[1]));
  real y = sp_fma(a[2], b[0], -(a[0] * b[2]));
  real z = sp_fma(a[0], b[1], -(a[1] * b[0]));
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void sp_add3(const real a[3], const real b[3], real o[3]) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void sp_sub3(const real a[3], const real b[3], real o[3]) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void sp_scale3(const real a[3], real s, real o[3]) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; }
/* o += s*a, one fma per component */
static inline void sp_axpy3(real s, const real a[3], real o[3]) {
  o[0] = sp_fma(s, a[0], o[0]); o[1] = sp_fma(s, a[1], o[1]); o[2] = sp_fma(s, a[2], o[2]);
}
static inline void sp_set3(real o[3], real x, real y, real z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void sp_copy3(const real a[3], real o[3]) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }

/* ---- quaternions (w,x,y,z), always unit ---------------------------------------------------------- */
/* rotate: t = 2*(u x v); out = v + s*t + u x t */
static inline void sp_rot(const real v[3], const real q[4], real o[3]) {
  real t[3], c[3];
  sp_cross3(q + 1, v, t);
  t[0] = t[0] + t[0]; t[1] = t[1] + t[1]; t[2] = t[2] + t[2];
  sp_cross3(q + 1, t, c);
  real x = sp_fma(q[0], t[0], v[0]) + c[0];
  real y = sp_fma(q[0], t[1], v[1]) + c[1];
  real z = sp_fma(q[0], t[2], v[2]) + c[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void sp_irot(const real v[3], const real q[4], real o[3]) {
  real qc[4] = {q[0], -q[1], -q[2], -q[3]};
  sp_rot(v, qc, o);
}
/* R(q)^T (0,0,d): the inverse rotation of a vector along z (the drop from a sphere's centre to its contact point),
 * written out — roughly half the operations of the general sp_irot */
static inline void sp_irot_z(real d, const real q[4], real o[3]) {
  real a = q[2] * d, b = q[1] * d;
  real tx = -(a + a), ty = b + b;
  real cx = q[3] * ty, cy = -(q[3] * tx), cz = sp_fma(-q[1], ty, q[2] * tx);
  real x = sp_fma(q[0], tx, cx), y = sp_fma(q[0], ty, cy), z = d + cz;
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void sp_qmul(const real a[4], const real b[4], real o[4]) {
  real w = sp_fma(-a[3], b[3], sp_fma(-a[2], b[2], sp_fma(-a[1], b[1], a[0] * b[0])));
  real x = sp_fma(-a[3], b[2], sp_fma(a[2], b[3], sp_fma(a[1], b[0], a[0] * b[1])));
  real y = sp_fma(a[3], b[1], sp_fma(a[2], b[0], sp_fma(-a[1], b[3], a[0] * b[2])));
  real z = sp_fma(a[3], b[0], sp_fma(-a[2], b[1], sp_fma(a[1], b[2], a[0] * b[3])));
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
/* q <- q / |q|.  Quaternions are renormalised after every first-order update, so e = |q|^2 - 1 is
 * tiny: 1/sqrt(1+e) = 1 - e/2 + 3e^2/8 - 5e^3/16 + 35e^4/128 (error 63/256 e^5 < 1e-7 for |e| <= 0.05,
 * i.e. up to 0.45 rad of rotation per substep); beyond that the exact sqrt + reciprocal. */
static inline void sp_qnormalize(real q[4]) {
  real n2 = sp_fma(q[0], q[0], sp_fma(q[1], q[1], sp_fma(q[2], q[2], q[3] * q[3])));
  real e = n2 - R(1);
  real inv = sp_fma(sp_fma(sp_fma(sp_fma(R(0.2734375), e, R(-0.3125)), e, R(0.375)), e, R(-0.5)), e, R(1));
  if (sp_abs(e) > R(0.05)) inv = R(1) / sp_sqrt(n2);
  q[0] = q[0] * inv; q[1] = q[1] * inv; q[2] = q[2] * inv; q[3] = q[3] * inv;
}
/* q <- normalize(q + 0.5*(0,th) (x) q): first-order update by the rotation vector th */
static inline void sp_qrotvec(real q[4], const real th[3]) {
  real h[3] = {R(0.5) * th[0], R(0.5) * th[1], R(0.5) * th[2]};
  real w = sp_fma(-h[2], q[3], sp_fma(-h[1], q[2], sp_fma(-h[0], q[1], q[0])));
  real x = sp_fma(-h[2], q[2], sp_fma(h[1], q[3], sp_fma(h[0], q[0], q[1])));
  real y = sp_fma(h[2], q[1], sp_fma(h[1], q[0], sp_fma(-h[0], q[3], q[2])));
  real z = sp_fma(h[2], q[0], sp_fma(-h[1], q[1], sp_fma(h[0], q[2], q[3])));
  q[0] = w; q[1] = x; q[2] = y; q[3] = z;
  sp_qnormalize(q);
}
/* columns of the rotation matrix of q: X = R e_x, Y = R e_y, Z = R e_z */
static inline void sp_qaxes(const real q[4], real X[3], real Y[3], real Z[3]) {
  const real w = q[0], x = q[1], y = q[2], z = q[3];
  real x2 = x + x, y2 = y + y, z2 = z + z;
  real xx = x * x2, yy = y * y2, zz = z * z2;
  real xy = x * y2, xz = x * z2, yz = y * z2;
  real wx = w * x2, wy = w * y2, wz = w * z2;
  X[0] = R(1) - (yy + zz); X[1] = xy + wz;          X[2] = xz - wy;
  Y[0] = xy - wz;          Y[1] = R(1) - (xx + zz); Y[2] = yz + wx;
  Z[0] = xz + wy;          Z[1] = yz - wx;          Z[2] = R(1) - (xx + yy);
}

/* ---- transcendental kernels ------------------------------------------------------------------------ */
/* atan2: a = min(|x|,|y|)/max(|x|,|y|) in [0,1]; atan(a) = a*P(a^2) (Abramowitz & Stegun 4.4.49,
 * |err| <= 2e-8); octant fix-ups.  atan2(0,0) = 0. */
static inline real sp_atan2(real y, real x) {
  real ax = sp_abs(x), ay = sp_abs(y);
  real mx = sp_max(ax, ay), mn = sp_min(ax, ay);
  real a = mx == R(0) ? R(0) : mn / mx;
  real s = a * a;
  real p = R(0.0028662257);
  p = sp_fma(p, s, R(-0.0161657367));
  p = sp_fma(p, s, R(0.0429096138));
  p = sp_fma(p, s, R(-0.0752896400));
  p = sp_fma(p, s, R(0.1065626393));
  p = sp_fma(p, s, R(-0.1420889944));
  p = sp_fma(p, s, R(0.1999355085));
  p = sp_fma(p, s, R(-0.3333314528));
  real r = sp_fma(p * s, a, a);
  if (ay > ax) r = R(1.57079632679489661923) - r;
  if (x < R(0)) r = R(3.14159265358979323846) - r;
  return y < R(0) ? -r : r;
}
/* angle of the (near-)unit vector (c, s), in (-pi, pi], WITHOUT a division: asin of the smaller of
 * |s|, |c| (<= 0.7072) by the odd minimax polynomial u + u^3 P(u^2) (|err| <= 5.3e-8), then octant
 * fix-ups. */
static inline real sp_angle_unit(real s, real c) {
  real as = sp_abs(s), ac = sp_abs(c);
  int swap = as > ac;
  real u = swap ? ac : as;
  real z = u * u;
  real p = R(0.11199134588241577);
  p = sp_fma(p, z, R(-0.09445883333683014));
  p = sp_fma(p, z, R(0.07875244319438934));
  p = sp_fma(p, z, R(0.015578965656459332));
  p = sp_fma(p, z, R(0.04668578505516052));
  p = sp_fma(p, z, R(0.07486556470394135));
  p = sp_fma(p, z, R(0.16666975617408752));
  real r = sp_fma(p * z, u, u);
  if (swap) r = R(1.57079632679489661923) - r;
  if (c < R(0)) r = R(3.14159265358979323846) - r;
  return sp_copysign(r, s); /* the SIGN BIT of s (r >= 0): one bit select on the GPU */
}
static inline real sp_asin(real v) { /* |v| <= 1 */
  real c2 = sp_fma(-v, v, R(1));
  return sp_angle_unit(v, sp_sqrt(c2 < R(0) ? R(0) : c2));
}
/* q <- q + 0.5*(0,th) (x) q  WITHOUT renormalisation (|q|^2 - 1 = |th|^2/4, renormalised by the next update) */
static inline void sp_qrotvec_raw(real q[4], const real th[3]) {
  real h[3] = {R(0.5) * th[0], R(0.5) * th[1], R(0.5) * th[2]};
  real w = sp_fma(-h[2], q[3], sp_fma(-h[1], q[2], sp_fma(-h[0], q[1], q[0])));
  real x = sp_fma(-h[2], q[2], sp_fma(h[1], q[3], sp_fma(h[0], q[0], q[1])));
  real y = sp_fma(h[2], q[1], sp_fma(h[1], q[0], sp_fma(-h[0], q[3], q[2])));
  real z = sp_fma(h[2], q[0], sp_fma(-h[1], q[1], sp_fma(h[0], q[2], q[3])));
  q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}

/* sin & cos, |x| < ~1e4: Cody–Waite reduction by pi/2 with fma, cephes sinf/cosf minimax kernels */
static inline void sp_sincos(real x, real* s_out, real* c_out) {
  real k = ORC_REAL_IS_F32 ? (real)__builtin_rintf((float)(x * R(0.63661977236758134308)))
                             : (real)__builtin_rint((double)(x * R(0.63661977236758134308)));
  real r = sp_fma(-k, R(1.5703125), x);                   /* pi/2 split in three parts */
  r = sp_fma(-k, R(4.837512969970703125e-4), r);
  r = sp_fma(-k, R(7.54978995489188216e-8), r);
  real z = r * r;
  real ps = sp_fma(sp_fma(R(-1.9515295891e-4), z, R(8.3321608736e-3)), z, R(-1.6666654611e-1));
  real sn = sp_fma(ps * z, r, r);
  real pc = sp_fma(sp_fma(R(2.443315711809948e-5), z, R(-1.388731625493765e-3)), z, R(4.166664568298827e-2));
  real cs = sp_fma(pc * z, z, sp_fma(R(-0.5), z, R(1)));
  int q = (int)k & 3;
  real s = (q & 1) ? cs : sn, c = (q & 1) ? sn : cs;
  if (q == 1) c = -c; else if (q == 2) { s = -s; c = -c; } else if (q == 3) s = -s;
  *s_out = s; *c_out = c;
}

/* everything below is float32-only (sampling / softmax never run in the f64 build) */
static inline float sp_exp_f32(float x) {
  /* e^x = 2^k * e^r, k = rint(x*log2e), r = x - k*ln2 (two-part), degree-6 Taylor-minimax (cephes) */
  if (x < -87.0f) return 0.0f; /* keeps the result a normal number */
  if (x > 88.7f) return INFINITY;
  float k = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(-k, 0.693359375f, x);
  r = __builtin_fmaf(-k, -2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  float e = __builtin_fmaf(p * r, r, r) + 1.0f;
  return __builtin_ldexpf(e, (int)k);
}
static inline float sp_log_f32(float x) {
  /* cephes logf: x = m * 2^e, m in [sqrt(1/2), sqrt(2)); log(m) = f - f^2/2 + f^3 P(f), f = m-1 */
  int e;
  float m = __builtin_frexpf(x, &e); /* m in [0.5, 1) */
  if (m < 0.707106781186547524f) { e = e - 1; m = m + m; }
  float f = m - 1.0f;
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = __builtin_fmaf(p, f, -1.1514610310e-1f);
  p = __builtin_fmaf(p, f, 1.1676998740e-1f);
  p = __builtin_fmaf(p, f, -1.2420140846e-1f);
  p = __builtin_fmaf(p, f, 1.4249322787e-1f);
  p = __builtin_fmaf(p, f, -1.6668057665e-1f);
  p = __builtin_fmaf(p, f, 2.0000714765e-1f);
  p = __builtin_fmaf(p, f, -2.4999993993e-1f);
  p = __builtin_fmaf(p, f, 3.3333331174e-1f);
  float y = (p * f) * z;
  float fe = (float)e;
  y = __builtin_fmaf(fe, -2.12194440e-4f, y);
  y = __builtin_fmaf(-0.5f, z, y);
  float r = f + y;
  return __builtin_fmaf(fe, 0.693359375f, r);
}
/* log1p(t) for t in (-1, 0]: u = 1+t; log(u) * t/(u-1) (the classic compensated form) */
static inline float sp_log1p_f32(float t) {
  float u = 1.0f + t;
  if (u == 1.0f) return t;
  if (u <= 0.0f) return -INFINITY;
  return sp_log_f32(u) * (t / (u - 1.0f));
}

/* canonical f32 reduction over n elements by ONE wavefront: 64 strided partials (element i goes to partial
 * i%64, in increasing i), then a xor-butterfly 32,16,8,4,2,1.  (The score kernel's workgroup-wide order —
 * 16 of these, added sequentially — and the weighted mean's 64-partial order are in mbd_oracle_core.c.) */
static inline float sp_reduce_sum64(const float* partial) {
  float a[64], b[64];
  memcpy(a, partial, sizeof(a));
  for (int off = 32; off >= 1; off >>= 1) {
    for (int j = 0; j < 64; ++j) b[j] = a[j] + a[j ^ off];
    memcpy(a, b, sizeof(a));
  }
  return a[0];
}
static inline float sp_reduce_max64(const float* partial) {
  float m = partial[0];
  for (int j = 1; j < 64; ++j) m = partial[j] > m ? partial[j] : m;
  return m;
}

#endif /* ORC_SPEC_MATH_H */
