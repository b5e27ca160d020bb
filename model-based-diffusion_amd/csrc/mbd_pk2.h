// mbd_pk2.h — the humanoid family's rollout with TWO CANDIDATES PER LANE (gfx950 packed FP32).
//
// rollout_kernel (mbd_kernels.h) keeps one link of one candidate per lane and packs (parent side, child side) pairs of
// a joint into v_pk_*_f32 — a third of its arithmetic; the rest is scalar FP32, and a wave64 VALU instruction holds its
// SIMD for one issue slot whether it carries 64 or 128 floats.  Once a launch puts more than one wavefront on a SIMD
// (N > 4096 humanoid candidates; seed sweeps as one batched launch) throughput, not latency, is what counts, and
// throughput is instructions issued per candidate.  Here a lane holds its link for the candidates (2k, 2k+1): every
// dynamic quantity is an f2 (.x = candidate 2k, .y = candidate 2k+1), every add / mul / fma is a v_pk instruction
// carrying both, and only what the ISA cannot pack — reciprocal seeds, compares and selects, clamps, sign transfers, the
// DPP row shifts — is issued once per half.  The (parent, child) pairs of the scalar kernel become two packed
// instructions (same issue cost per candidate); the scalar two thirds cost half.  Per-lane model constants (the same
// LaneRec3 records) are shared by the two candidates.
//
// The arithmetic per candidate is the scalar kernel's, operation for operation (same fma placement, same exact-division /
// square-root sequences, same order of the children's contributions): results are bit-identical to rollout_kernel and to
// the CPU checker, whatever kernel a launch picks (tests/test_gpu_parity.py::test_pk2_*).
//
// Instantiated for the humanoid-shaped models only (isotropic inertia, no slide joints, up to three children per link,
// the DPP layout (+1, -4, -6)): humanoidrun, humanoidtrack (MAXCOL = 1), humanoidstandup (MAXCOL = 5).
#pragma once

#include "mbd_kernels.h"

namespace mbd {

// (of the specification switches a tuned build may compile in, MBD_TUNED_SPEC, this kernel knows contact_avg; a build with any
// other keeps its launches on the one-candidate kernels: rollout_uses_pk2, mbd_env.hip)

struct b2 {
  bool x, y;
};
__device__ __forceinline__ f2 splat(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 sel(b2 c, f2 a, f2 b) { return mk2(c.x ? a.x : b.x, c.y ? a.y : b.y); }
__device__ __forceinline__ f2 sel(bool c, f2 a, f2 b) { return mk2(c ? a.x : b.x, c ? a.y : b.y); }
__device__ __forceinline__ v3x2 sel3(b2 c, v3x2 a, v3x2 b) { return v3x2{sel(c, a.x, b.x), sel(c, a.y, b.y), sel(c, a.z, b.z)}; }
__device__ __forceinline__ v3x2 sel3(bool c, v3x2 a, v3x2 b) { return v3x2{sel(c, a.x, b.x), sel(c, a.y, b.y), sel(c, a.z, b.z)}; }
__device__ __forceinline__ f2 fclip2(f2 v, float lo, float hi) { return mk2(fclip(v.x, lo, hi), fclip(v.y, lo, hi)); }
__device__ __forceinline__ f2 fmaxs2(f2 a, float b) { return mk2(fmax_(a.x, b), fmax_(a.y, b)); }
__device__ __forceinline__ f2 fmin2(f2 a, f2 b) { return mk2(fmin_(a.x, b.x), fmin_(a.y, b.y)); }
__device__ __forceinline__ f2 fabs2(f2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ f2 copysign2(f2 mag, f2 sgn) { return mk2(__builtin_copysignf(mag.x, sgn.x), __builtin_copysignf(mag.y, sgn.y)); }
__device__ __forceinline__ v3x2 zero3x2() { return v3x2{splat(0.0f), splat(0.0f), splat(0.0f)}; }
__device__ __forceinline__ v3x2 neg2(v3x2 a) { return v3x2{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ v3x2 scale2s(v3x2 a, float s) { return scale2(a, splat(s)); }
__device__ __forceinline__ q4x2 conj2(q4x2 q) { return q4x2{q.w, -q.x, -q.y, -q.z}; }
__device__ __forceinline__ f2 rcp_exact2(f2 d) {
  f2 r = mk2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
  f2 e = fma2(-d, r, splat(1.0f));
  return fma2(e, r, r);
}
__device__ __forceinline__ f2 sqrt_floor2(f2 x) {
  x = fmaxs2(x, 1e-30f);
  f2 r = mk2(__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y));
  f2 g = x * r, h = splat(0.5f) * r;
  f2 d = fma2(-g, g, x);
  return fma2(d, h, g);
}
__device__ __forceinline__ q4x2 qnormalize2(q4x2 q) {
  f2 n2 = fma2(q.w, q.w, fma2(q.x, q.x, fma2(q.y, q.y, q.z * q.z)));
  f2 e = n2 - splat(1.0f);
  f2 inv = fma2(fma2(fma2(fma2(splat(0.2734375f), e, splat(-0.3125f)), e, splat(0.375f)), e, splat(-0.5f)), e, splat(1.0f));
  // (ONE cold block for the pair, with one plain `if` per half inside it: out of line, and the common path pays one
  // compare pair, one exec save and one untaken branch.  As `far0 || far1` with selects inside it compiled to an if /
  // else whose common side left by a TAKEN branch per substep; as two top-level ifs to twice the exec bookkeeping.)
  float i0 = inv.x, i1 = inv.y;
  const float ax = fabs_(e.x), ay = fabs_(e.y);
  if (__builtin_expect(fmax_(ax, ay) > 0.05f, 0)) {
    if (ax > 0.05f) i0 = 1.0f / fsqrt(n2.x);
    if (ay > 0.05f) i1 = 1.0f / fsqrt(n2.y);
  }
  inv = mk2(i0, i1);
  return q4x2{q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
__device__ __forceinline__ q4x2 qrotvec_raw2(q4x2 q, v3x2 th) {
  const f2 half = splat(0.5f);
  f2 hx = half * th.x, hy = half * th.y, hz = half * th.z;
  q4x2 o;
  o.w = fma2(-hz, q.z, fma2(-hy, q.y, fma2(-hx, q.x, q.w)));
  o.x = fma2(-hz, q.y, fma2(hy, q.z, fma2(hx, q.w, q.x)));
  o.y = fma2(hz, q.x, fma2(hy, q.w, fma2(-hx, q.z, q.y)));
  o.z = fma2(hz, q.w, fma2(-hy, q.x, fma2(hx, q.y, q.z)));
  return o;
}
__device__ __forceinline__ q4x2 qrotvec2(q4x2 q, v3x2 th) { return qnormalize2(qrotvec_raw2(q, th)); }
// the pair form of qnormalize_qm (mbd_math.h): QM = 1 the series unconditionally, the largest |n2 - 1| of either half kept in
// `worst`; QM = 2 both sides computed, the exact one selected per half
template <int QM>
__device__ __forceinline__ q4x2 qnormalize2_qm(q4x2 q, float& worst) {
  if constexpr (QM == 0) return qnormalize2(q);
  f2 n2 = fma2(q.w, q.w, fma2(q.x, q.x, fma2(q.y, q.y, q.z * q.z)));
  f2 e = n2 - splat(1.0f);
  f2 inv = fma2(fma2(fma2(fma2(splat(0.2734375f), e, splat(-0.3125f)), e, splat(0.375f)), e, splat(-0.5f)), e, splat(1.0f));
  const float ax = fabs_(e.x), ay = fabs_(e.y);
  if constexpr (QM == 1) {
    worst = fmax_(worst, fmax_(ax, ay));
  } else {
    const float x0 = 1.0f / fsqrt(n2.x), x1 = 1.0f / fsqrt(n2.y);
    inv = mk2(ax > 0.05f ? x0 : inv.x, ay > 0.05f ? x1 : inv.y);
  }
  return q4x2{q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
template <int QM>
__device__ __forceinline__ q4x2 qrotvec2_qm(q4x2 q, v3x2 th, float& worst) { return qnormalize2_qm<QM>(qrotvec_raw2(q, th), worst); }
// angle_unit_cpos on a pair
__device__ __forceinline__ f2 angle_unit_cpos2(f2 s, f2 c) {
  f2 as = fabs2(s);
  const bool sw0 = as.x > c.x, sw1 = as.y > c.y;
  f2 u = fmin2(as, c);
  f2 z = u * u;
  f2 p = splat(0.11199134588241577f);
  p = fma2(p, z, splat(-0.09445883333683014f));
  p = fma2(p, z, splat(0.07875244319438934f));
  p = fma2(p, z, splat(0.015578965656459332f));
  p = fma2(p, z, splat(0.04668578505516052f));
  p = fma2(p, z, splat(0.07486556470394135f));
  p = fma2(p, z, splat(0.16666975617408752f));
  f2 r = fma2(p * z, u, u);
  float r0 = r.x, r1 = r.y;
  r0 = sw0 ? 1.57079632679489661923f - r0 : r0;
  r1 = sw1 ? 1.57079632679489661923f - r1 : r1;
  return mk2(__builtin_copysignf(r0, s.x), __builtin_copysignf(r1, s.y));
}
__device__ __forceinline__ v3x2 shfl3x2(v3x2 v, int src) {
  return v3x2{mk2(shfl(v.x.x, src), shfl(v.x.y, src)), mk2(shfl(v.y.x, src), shfl(v.y.y, src)),
              mk2(shfl(v.z.x, src), shfl(v.z.y, src))};
}
__device__ __forceinline__ q4x2 shfl4x2(q4x2 q, int src) {
  return q4x2{mk2(shfl(q.w.x, src), shfl(q.w.y, src)), mk2(shfl(q.x.x, src), shfl(q.x.y, src)),
              mk2(shfl(q.y.x, src), shfl(q.y.y, src)), mk2(shfl(q.z.x, src), shfl(q.z.y, src))};
}
__device__ __forceinline__ v3x2 mk3x2(v3 a, v3 b) { return pack3(a, b); }
__device__ __forceinline__ q4 hi4(q4x2 a) { return q4{a.w.y, a.x.y, a.y.y, a.z.y}; }

// the joint frames of one joint for both candidates (joint_frames of mbd_kernels.h, per half)
struct JointFramesP {
  v3x2 ap, ac, rp, rc;
  v3x2 Xp, Xc, Yc, Zc, ax1;
  f2 ang0, ang1, ang2;
};
template <bool MULTI = true>
__device__ __forceinline__ JointFramesP joint_frames_p(const JointConst& jc, v3x2 Pp, q4x2 Pr, v3x2 Cp, q4x2 Cr) {
  JointFramesP f;
  f.rp = rot2(bcast3(jc.ap_pos), Pr);
  f.rc = rot2(bcast3(jc.ac_pos), Cr);
  f.ap = add2(Pp, f.rp);
  f.ac = add2(Cp, f.rc);
  const axes3x2 A = qaxes2(qmul2(Pr, bcast4(jc.ap_rot)));
  const axes3x2 C = qaxes2(qmul2(Cr, bcast4(jc.ac_rot)));
  f.Xp = A.X; f.Xc = C.X; f.Yc = C.Y; f.Zc = C.Z;
  const f2 sb = fclip2(dot2(C.Z, A.X), -1.0f, 1.0f);
  const f2 cb2 = fma2(-sb, sb, splat(1.0f));
  const f2 cb = sqrt_floor2(cb2);
  const f2 inv = rcp_exact2(cb + splat(1e-10f));
  f.ang0 = angle_unit2((-dot2(C.Z, A.Y)) * inv, dot2(C.Z, A.Z) * inv);
  if constexpr (MULTI) {
    f.ang1 = angle_unit_cpos2(sb, cb);
    f.ang2 = angle_unit2((-dot2(C.Y, A.X)) * inv, dot2(C.X, A.X) * inv);
    f.ax1 = scale2(cross2(C.Z, A.X), inv);
  } else {  // single-hinge models only ever use the first Euler angle and axis (joint_frames of mbd_kernels.h)
    f.ang1 = f.ang2 = splat(0.0f);
    f.ax1 = zero3x2();
  }
  return f;
}

// children -> parent sums and the parent's pose for both halves in ONE block each (one hazard s_nop per exchange; the
// one-candidate kernel's forms, dpp_acc6x3 / dpp_fetch7<1, -4, -6>, per 32-bit half: same operations, same order)
#define MBD_PF(R, X, M, MOD) "v_fmac_f32_dpp %" #R ", %" #X ", %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define MBD_PF12(MOD, M)                                                                                         \
  MBD_PF(0, 12, M, MOD) MBD_PF(1, 13, M, MOD) MBD_PF(2, 14, M, MOD) MBD_PF(3, 15, M, MOD) MBD_PF(4, 16, M, MOD)  \
  MBD_PF(5, 17, M, MOD) MBD_PF(6, 18, M, MOD) MBD_PF(7, 19, M, MOD) MBD_PF(8, 20, M, MOD) MBD_PF(9, 21, M, MOD)  \
  MBD_PF(10, 22, M, MOD) MBD_PF(11, 23, M, MOD)
#define MBD_PK2_ACC_IO                                                                                            \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), \
        "+v"(a11)                                                                                                 \
      : "v"(x.x.x), "v"(x.y.x), "v"(x.z.x), "v"(y.x.x), "v"(y.y.x), "v"(y.z.x), "v"(x.x.y), "v"(x.y.y), "v"(x.z.y), \
        "v"(y.x.y), "v"(y.y.y), "v"(y.z.y), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3])
// FAM: the DPP layout family — 0: the humanoids (a link's children sit at lane - 1, + 4, + 6), 1: ant (- 1, + 2, + 4, + 6)
template <int FAM>
__device__ __forceinline__ void dpp_acc_p(v3x2& a, v3x2& b, v3x2 x, v3x2 y, const float (&m)[4]) {
  float a0 = a.x.x, a1 = a.y.x, a2 = a.z.x, a3 = b.x.x, a4 = b.y.x, a5 = b.z.x;
  float a6 = a.x.y, a7 = a.y.y, a8 = a.z.y, a9 = b.x.y, a10 = b.y.y, a11 = b.z.y;
  if constexpr (FAM == 0) {
    asm("s_nop 1\n\t" MBD_PF12("row_shr:1", 24) MBD_PF12("row_shl:4", 25) MBD_PF12("row_shl:6", 26) MBD_PK2_ACC_IO);
  } else {
    asm("s_nop 1\n\t" MBD_PF12("row_shr:1", 24) MBD_PF12("row_shl:2", 25) MBD_PF12("row_shl:4", 26) MBD_PF12("row_shl:6", 27)
        MBD_PK2_ACC_IO);
  }
  a = v3x2{mk2(a0, a6), mk2(a1, a7), mk2(a2, a8)};
  b = v3x2{mk2(a3, a9), mk2(a4, a10), mk2(a5, a11)};
}
#undef MBD_PK2_ACC_IO
#undef MBD_PF12
#define MBD_PF14(MOD, M)                                                                                         \
  MBD_PF(0, 14, M, MOD) MBD_PF(1, 15, M, MOD) MBD_PF(2, 16, M, MOD) MBD_PF(3, 17, M, MOD) MBD_PF(4, 18, M, MOD)  \
  MBD_PF(5, 19, M, MOD) MBD_PF(6, 20, M, MOD) MBD_PF(7, 21, M, MOD) MBD_PF(8, 22, M, MOD) MBD_PF(9, 23, M, MOD)  \
  MBD_PF(10, 24, M, MOD) MBD_PF(11, 25, M, MOD) MBD_PF(12, 26, M, MOD) MBD_PF(13, 27, M, MOD)
#define MBD_PK2_FETCH_OUT                                                                                         \
      : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+v"(o6), "+v"(o7), "+v"(o8), "+v"(o9), "+v"(o10), \
        "+v"(o11), "+v"(o12), "+v"(o13)
#define MBD_PK2_FETCH_IN                                                                                          \
      "v"(p.x.x), "v"(p.y.x), "v"(p.z.x), "v"(r.w.x), "v"(r.x.x), "v"(r.y.x), "v"(r.z.x), "v"(p.x.y), "v"(p.y.y),  \
        "v"(p.z.y), "v"(r.w.y), "v"(r.x.y), "v"(r.y.y), "v"(r.z.y)
template <int FAM>
__device__ __forceinline__ void dpp_fetch_p(v3x2 p, q4x2 r, const float (&m)[4], v3x2& Pp, q4x2& Pr) {
  const float m0 = m[0];
  float o0 = dpp_from<1>(p.x.x) * m0, o1 = dpp_from<1>(p.y.x) * m0, o2 = dpp_from<1>(p.z.x) * m0;
  float o3 = dpp_from<1>(r.w.x) * m0, o4 = dpp_from<1>(r.x.x) * m0, o5 = dpp_from<1>(r.y.x) * m0;
  float o6 = dpp_from<1>(r.z.x) * m0;
  float o7 = dpp_from<1>(p.x.y) * m0, o8 = dpp_from<1>(p.y.y) * m0, o9 = dpp_from<1>(p.z.y) * m0;
  float o10 = dpp_from<1>(r.w.y) * m0, o11 = dpp_from<1>(r.x.y) * m0, o12 = dpp_from<1>(r.y.y) * m0;
  float o13 = dpp_from<1>(r.z.y) * m0;
  if constexpr (FAM == 0) {
    asm("s_nop 1\n\t" MBD_PF14("row_shr:4", 28) MBD_PF14("row_shr:6", 29) MBD_PK2_FETCH_OUT
        : MBD_PK2_FETCH_IN, "v"(m[1]), "v"(m[2]));
  } else {  // (30 operands at most per block: the fourth slot in a block of its own)
    asm("s_nop 1\n\t" MBD_PF14("row_shr:2", 28) MBD_PF14("row_shr:4", 29) MBD_PK2_FETCH_OUT
        : MBD_PK2_FETCH_IN, "v"(m[1]), "v"(m[2]));
    asm("s_nop 1\n\t" MBD_PF14("row_shr:6", 28) MBD_PK2_FETCH_OUT : MBD_PK2_FETCH_IN, "v"(m[3]));
  }
  Pp = v3x2{mk2(o0, o7), mk2(o1, o8), mk2(o2, o9)};
  Pr = q4x2{mk2(o3, o10), mk2(o4, o11), mk2(o5, o12), mk2(o6, o13)};
}
#undef MBD_PK2_FETCH_IN
#undef MBD_PK2_FETCH_OUT
#undef MBD_PF14
#undef MBD_PF

// MAXCOL: most sphere colliders on one link (1: humanoidrun / humanoidtrack, 5: humanoidstandup)
// RK, NFR: reward kind and n_frames as compile-time constants (-1 / 0: read at run time), as in rollout_kernel
// WPE: wavefronts per SIMD the register allocation leaves room for.  1: 302 registers (256 + 46 accumulation registers
//      holding what a control step needs once), nothing in scratch — the launch of up to one wavefront per SIMD (8192
//      candidates on 256 CUs).  2: capped at 256 — 23 scratch accesses per CONTROL step, none in the substep loop — so
//      that larger launches run two wavefronts per SIMD, whose instruction fetches and hazard wait states overlap
//      (N = 32768: 3.60 -> 3.39 ms; N = 8192, one wavefront per SIMD either way: 0.97 -> 1.09 ms)
// FAM: 0 the humanoid family (three child slots, multi-dof joints), 1 ant (four child slots, single hinges, the
//      control-cost reward: the action row of the next control step travels with the other prefetched actions)
template <int MAXCOL, int RK = -1, int NFR = 0, int WPE = 1, int FAM = 0>
__global__ __launch_bounds__(256, WPE) void rollout_pk2_kernel(RolloutParams P) {
  constexpr bool MULTI = FAM == 0;
  rollout_progress(P);
  if ((int)blockIdx.x >= P.roll_blocks) {  // the next step's normals, on CUs the rollout leaves idle
    noise_blocks(P);
    return;
  }
  constexpr int LPS = 16;
  const mbd_model_t* __restrict__ Mg = P.model;
  const int lane = threadIdx.x & 63;
  const int base = lane & ~(LPS - 1);
  const int l_lane = lane & (LPS - 1);
  const LaneRec3& R = reinterpret_cast<const LaneRec3*>(P.lane_rec[1])[l_lane];
  const int L = Mg->n_links;
  const bool link_ok = R.link_ok != 0;
  const int l = R.l;
  const bool root_lane = R.root_lane != 0;
  constexpr int PPW = 64 / LPS;  // candidate PAIRS per wavefront
  const int wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int bA_raw = (wave_id * PPW + lane / LPS) * 2, bB_raw = bA_raw + 1;
  const bool okA = bA_raw < P.B, okB = bB_raw < P.B;
  const int bA = okA ? bA_raw : P.B - 1, bB = okB ? bB_raw : P.B - 1;
  const int H = P.H, Nu = Mg->n_act, nfr = NFR > 0 ? NFR : Mg->n_frames, K = Mg->n_track;

  const int nr = R.nr;
  const int plane = base + R.plane_rel;
  const float ic_inv_mass = R.ic_inv_mass, ip_inv_mass = R.ip_inv_mass, ic_ib = R.ic_ib[0], ip_ib = R.ip_ib[0];
  JointConst jc;
  jc.ap_pos = mk3(R.ap_pos[0], R.ap_pos[1], R.ap_pos[2]);
  jc.ac_pos = mk3(R.ac_pos[0], R.ac_pos[1], R.ac_pos[2]);
  jc.ap_rot = q4{R.ap_rot[0], R.ap_rot[1], R.ap_rot[2], R.ap_rot[3]};
  jc.ac_rot = q4{R.ac_rot[0], R.ac_rot[1], R.ac_rot[2], R.ac_rot[3]};
  const float ang_damp = R.ang_damp, vel_damp = R.vel_damp;
  const int nr_eff = R.nr_eff;
  float lim_lo[3], lim_hi[3], stiff[3], damp[3];
  int act_rot[3];
  float gear_rot[3], alo_rot[3], ahi_rot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lim_lo[k] = R.lim_lo[k]; lim_hi[k] = R.lim_hi[k]; stiff[k] = R.stiff[k]; damp[k] = R.damp[k];
    act_rot[k] = R.act_rot[k];
    gear_rot[k] = R.gear_rot[k]; alo_rot[k] = R.alo_rot[k]; ahi_rot[k] = R.ahi_rot[k];
  }
  float rm[4], pm[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { rm[k] = R.rm[k]; pm[k] = R.pm[k]; }
  v3 col_pos[MAXCOL];
  float col_rad[MAXCOL];
  bool col_has[MAXCOL];
#pragma unroll
  for (int j = 0; j < MAXCOL; ++j) {
    col_has[j] = R.col_has[j] != 0; col_rad[j] = R.col_rad[j];
    col_pos[j] = mk3(R.col_pos[j][0], R.col_pos[j][1], R.col_pos[j][2]);
  }
  const int track_k = R.track_k;
  const v3 com = mk3(R.com[0], R.com[1], R.com[2]);
  const float dt = Mg->dt, inv_dt = 1.0f / Mg->dt, vel_fac = Mg->vel_fac, ang_fac = Mg->ang_fac;
  const float two_inv_dt = 2.0f * inv_dt;
  const float js_pos = R.js_pos;
  const float coll_scale = Mg->collide_scale, invm_sum = R.invm_sum;
  const float kang_p = R.kang2[0], kang_c = R.kang2[1];
  const float mu = Mg->friction, elast = Mg->elasticity;
  const v3 grav = mk3(Mg->gravity[0], Mg->gravity[1], Mg->gravity[2]);
  const int rkind = RK >= 0 ? RK : Mg->reward_kind;
  const float rp0 = Mg->reward_params[0], rp1 = Mg->reward_params[1];
  const float dt_ctrl = Mg->dt * (float)nfr;

  // ---- state: both candidates start from the same state0 --------------------------------------------------
  const int pl = plan_of(P, bA);  // (sweeps: plans hold an even number of candidates, a pair never straddles two)
  const float* s0 = P.state0 + (size_t)pl * P.plan_state_stride + l * MBD_LINK_STATE;
  v3 p1 = mk3(s0[0], s0[1], s0[2]);
  q4 r1 = q4{s0[3], s0[4], s0[5], s0[6]};
  v3 v1 = mk3(s0[7], s0[8], s0[9]);
  v3 w1 = mk3(s0[10], s0[11], s0[12]);
  if (!link_ok) { p1 = mk3(0, 0, 0); r1 = q4{1, 0, 0, 0}; v1 = mk3(0, 0, 0); w1 = mk3(0, 0, 0); }
  v3x2 p = bcast3(p1), v = bcast3(v1), w = bcast3(w1);
  q4x2 r = bcast4(r1);

  const float* uA = P.us + (size_t)bA * H * Nu;
  const float* uB = P.us + (size_t)bB * H * Nu;
  const bool lazy = P.ybar != nullptr;
  const float* __restrict__ yb_row = lazy ? P.ybar + (size_t)pl * P.plan_ybar_stride : P.us;
  const float sigma = P.sigma;
  auto cand = [&](f2 e, float yb) {  // (mul, add: the sampler's roundings)
    const f2 c = fclip2(e * splat(sigma) + splat(yb), -1.0f, 1.0f);
    return lazy ? c : e;
  };
  f2 u_rot[3], un_rot[3];
  float y_rot[3] = {0.0f, 0.0f, 0.0f}, yn_rot[3] = {0.0f, 0.0f, 0.0f};
  auto load_actions = [&](int t, f2 (&ur)[3], float (&yr)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const size_t o = (size_t)t * Nu + (act_rot[k] >= 0 ? act_rot[k] : 0);
      ur[k] = mk2(uA[o], uB[o]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) yr[k] = yb_row[(size_t)t * Nu + (act_rot[k] >= 0 ? act_rot[k] : 0)];
  };
  load_actions(0, u_rot, y_rot);
  // control cost (ant): the whole action row of both candidates, in actuator order (rollout_kernel's cc_* prefetch)
  constexpr int KCC = FAM == 1 ? 8 : 0;
  f2 cc_u[KCC + 1], ccn_u[KCC + 1];
  float cc_y[KCC + 1], ccn_y[KCC + 1];
  auto load_row = [&](int t, f2 (&ru)[KCC + 1], float (&ry)[KCC + 1]) {
#pragma unroll
    for (int k = 0; k < KCC; ++k) {
      const size_t o = (size_t)t * Nu + (k < Nu ? k : 0);
      ru[k] = mk2(uA[o], uB[o]);
      ry[k] = yb_row[o];
    }
  };
#pragma unroll
  for (int k = 0; k < KCC + 1; ++k) { cc_u[k] = ccn_u[k] = splat(0.0f); cc_y[k] = ccn_y[k] = 0.0f; }
  if constexpr (KCC > 0) load_row(0, cc_u, cc_y);
  v3x2 Pp_next = shfl3x2(p, plane);
  q4x2 Pr_next = shfl4x2(r, plane);
  f2 rew_sum = splat(0.0f);

  for (int t = 0; t < H; ++t) {
    f2 tau[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u_rot[k] = cand(u_rot[k], y_rot[k]);
      tau[k] = fclip2(act_rot[k] >= 0 ? u_rot[k] : splat(0.0f), alo_rot[k], ahi_rot[k]) * splat(gear_rot[k]);
    }
    f2 ctrl_cost = splat(0.0f);
    if constexpr (KCC > 0) {
#pragma unroll
      for (int k = 0; k < KCC; ++k) {
        const f2 ua = cand(cc_u[k], cc_y[k]);
        ctrl_cost = k < Nu ? ctrl_cost + ua * ua : ctrl_cost;
      }
      for (int a = KCC; a < Nu; ++a) {
        const f2 ua = cand(mk2(uA[(size_t)t * Nu + a], uB[(size_t)t * Nu + a]), yb_row[(size_t)t * Nu + a]);
        ctrl_cost = ctrl_cost + ua * ua;
      }
    }
    load_actions(t + 1 < H ? t + 1 : t, un_rot, yn_rot);
    if constexpr (KCC > 0) load_row(t + 1 < H ? t + 1 : t, ccn_u, ccn_y);
    __builtin_amdgcn_sched_barrier(0);
    // link-frame origin before the step (the tracking reward looks at the incoming state)
    v3x2 o0 = zero3x2(), v0 = zero3x2();
    if (rkind == MBD_REW_HUMANOIDTRACK || rkind == MBD_REW_ANT) {
      const v3x2 rc0 = rot2(bcast3(com), r);
      o0 = sub2(p, rc0);
      v0 = sub2(v, cross2(w, rc0));
    }

    // (the renormalisations' rare exact side speculated like the planar kernels': -DMBD_PK2_SPECULATE; measured in round 6 —
    // see below at the substep loop)
#ifdef MBD_PK2_SPECULATE
    constexpr int QM_FAST = NFR > 1 ? 1 : 0;
#else
    constexpr int QM_FAST = 0;
#endif
    float q_worst = 0.0f;
    const v3x2 s_p = p, s_v = v, s_w = w;
    const q4x2 s_r = r;
    auto substep_qm = [&](auto qm_tag) __attribute__((always_inline)) {
      constexpr int QM = decltype(qm_tag)::value;
      // ---- (1) joints.acceleration_update ----------------------------------------------------------
      v3x2 Pv = shfl3x2(v, plane), Pw = shfl3x2(w, plane);
      shfl_issue();
      v3x2 Pp = Pp_next;
      q4x2 Pr = Pr_next;
      v3x2 fc_v, fc_w, fp_v, fp_w;
      {
        const JointFramesP f = joint_frames_p<MULTI>(jc, Pp, Pr, p, r);
        shfl_join();
        const v3x2 vp = add2(Pv, cross2(Pw, f.rp)), vc = add2(v, cross2(w, f.rc));  // anchor velocities
        const v3x2 rel_v = sub2(vc, vp), rel_w = sub2(w, Pw);
        v3x2 T = zero3x2();
        auto torque = [&](int k, v3x2 ax, f2 ang) {
          f2 qdk = dot2(rel_w, ax);
          f2 fk = fma2(splat(-stiff[k]), ang, fma2(splat(-damp[k]), qdk, tau[k]));
          fk = k < nr_eff ? fk : splat(0.0f);
          T = axpy2(fk, ax, T);
        };
        torque(0, f.Xp, f.ang0);
        if constexpr (MULTI) {
          torque(1, f.ax1, f.ang1);
          torque(2, f.Zc, f.ang2);
        }
        T = axpy2(splat(-ang_damp), rel_w, T);
        const v3x2 F = axpy2(splat(-vel_damp), rel_v, zero3x2());
        fp_v = scale2s(F, -ip_inv_mass);
        fc_v = scale2s(F, ic_inv_mass);
        const v3x2 totp = add2(T, cross2(f.rp, F)), totc = add2(T, cross2(f.rc, F));
        fp_w = neg2(scale2s(totp, ip_ib));
        fc_w = scale2s(totc, ic_ib);
      }
      // ---- (2) integrator.integrate_xdd -------------------------------------------------------------
      {
        v3x2 sv = fc_v, sw = fc_w;
        dpp_acc_p<FAM>(sv, sw, fp_v, fp_w, rm);
        const f2 vf = splat(vel_fac), af = splat(ang_fac), dt2 = splat(dt);
        v = v3x2{fma2(sv.x + splat(grav.x), dt2, vf * v.x), fma2(sv.y + splat(grav.y), dt2, vf * v.y),
                 fma2(sv.z + splat(grav.z), dt2, vf * v.z)};
        w = v3x2{fma2(sw.x, dt2, af * w.x), fma2(sw.y, dt2, af * w.y), fma2(sw.z, dt2, af * w.z)};
      }
      const v3x2 p_prev = p;
      const q4x2 r_prev = r;
      p = v3x2{fma2(v.x, splat(dt), p.x), fma2(v.y, splat(dt), p.y), fma2(v.z, splat(dt), p.z)};
      r = qrotvec2_qm<QM>(r, scale2s(w, dt), q_worst);
      // ---- (3) joints.position_update (Jacobi) ------------------------------------------------------
      dpp_fetch_p<FAM>(p, r, pm, Pp, Pr);
      {
        const JointFramesP f = joint_frames_p<MULTI>(jc, Pp, Pr, p, r);
        const v3x2 d = sub2(f.ap, f.ac);
        const f2 c2 = dot2(d, d);
        const v3x2 crp = cross2(f.rp, d), crc = cross2(f.rc, d);
        const f2 wqp = dot2(crp, scale2s(crp, ip_ib)), wqc = dot2(crc, scale2s(crc, ic_ib));
        const f2 den = fma2(splat(invm_sum), c2, wqp + wqc) + splat(1e-20f);
        // angular alignment by joint type (1: Xc || Xp; 2: Yc _|_ Xp; 3: free)
        const v3x2 A = sel3(nr == 1, f.Xc, f.Xp);
        const v3x2 Bv = sel3(nr == 1, f.Xp, f.Yc);
        const f2 dxy = dot2(f.Xp, f.Yc);
        const f2 sc = nr_eff == 1 ? splat(1.0f) : (nr_eff == 2 ? dxy : splat(0.0f));
        v3x2 E = scale2(cross2(A, Bv), sc);
        auto viol_of = [&](int k, f2 a) { return k < nr_eff ? a - fclip2(a, lim_lo[k], lim_hi[k]) : splat(0.0f); };
        E = axpy2(-viol_of(0, f.ang0), f.Xp, E);
        if constexpr (MULTI) {
          E = axpy2(-viol_of(1, f.ang1), f.ax1, E);
          E = axpy2(-viol_of(2, f.ang2), f.Zc, E);
        }
        const f2 g = div2_pos_(c2, den) * splat(js_pos);
        const v3x2 Pi = scale2(d, g);
        const v3x2 dp_p = scale2s(Pi, -ip_inv_mass);
        v3x2 dc_p = scale2s(Pi, ic_inv_mass);
        v3x2 dp_th = neg2(scale2s(cross2(f.rp, Pi), ip_ib));
        v3x2 dc_th = scale2s(cross2(f.rc, Pi), ic_ib);
        dp_th = axpy2(splat(kang_p), E, dp_th);
        dc_th = axpy2(splat(kang_c), E, dc_th);
        dpp_acc_p<FAM>(dc_p, dc_th, dp_p, dp_th, rm);
        p = add2(p, dc_p);
        r = qrotvec_raw2(r, dc_th);  // renormalised at the end of stage (4)
      }
      // ---- (4) sphere-plane contacts + collisions.resolve_position ---------------------------------
      v3x2 con_pos[MAXCOL];
      f2 con_dlam[MAXCOL];
      b2 con_act[MAXCOL];
      {
        v3x2 cd_p = zero3x2(), cd_th = zero3x2();
#pragma unroll
        for (int j = 0; j < MAXCOL; ++j) {
          const v3x2 off = rot2(bcast3(col_pos[j]), r);
          const v3x2 ctr = add2(p, off);
          const f2 pen = splat(col_rad[j]) - ctr.z;
          const b2 active{col_has[j] && pen.x > 0.0f, col_has[j] && pen.y > 0.0f};
          const f2 h = fma2(splat(-0.5f), pen, splat(col_rad[j]));
          const v3x2 pos = v3x2{ctr.x, ctr.y, ctr.z - h};
          const v3x2 rc = v3x2{off.x, off.y, off.z - h};
          const f2 cnx = rc.y, cny = -rc.x;  // crossz(rc)
          const f2 icnx = cnx * splat(ic_ib), icny = cny * splat(ic_ib);
          const f2 wn = splat(ic_inv_mass) + fma2(cnx, icnx, cny * icny);
          const v3x2 rl = add2(bcast3(col_pos[j]), irot_z2(-h, r));
          const v3x2 pprev = add2(p_prev, rot2(rl, r_prev));
          const f2 dxx = pos.x - pprev.x, dxy_ = pos.y - pprev.y;
          const f2 ct2 = fma2(dxx, dxx, dxy_ * dxy_);
          const v3x2 cnt = v3x2{-(rc.z * dxy_), rc.z * dxx, fma2(rc.x, dxy_, -(rc.y * dxx))};  // cross_bz0(rc, dx)
          const v3x2 icnt = scale2s(cnt, ic_ib);
          const f2 dent = fma2(splat(ic_inv_mass), ct2, dot2(cnt, icnt));
          const f2 q_n = div2_pos_(pen, wn), gt = div2_pos_(ct2, dent + splat(1e-20f));
          const f2 dlam = q_n * splat(coll_scale);
          const f2 lim = splat(mu) * dlam;
          const f2 lhs = (ct2 * gt) * gt, rhs = lim * lim;
          const b2 stick{lhs.x < rhs.x, lhs.y < rhs.y};
          const f2 zero = splat(0.0f);
          const v3x2 Pimp = v3x2{sel(stick, (-gt) * dxx, zero), sel(stick, (-gt) * dxy_, zero), dlam};
          const v3x2 dth = scale2s(cross2(rc, Pimp), ic_ib);
          const v3x2 ncd_p = j == 0 ? scale2s(Pimp, ic_inv_mass) : axpy2(splat(ic_inv_mass), Pimp, cd_p);
          const v3x2 ncd_th = j == 0 ? dth : add2(cd_th, dth);
          cd_p = sel3(active, ncd_p, cd_p);
          cd_th = sel3(active, ncd_th, cd_th);
          con_pos[j] = pos; con_dlam[j] = dlam; con_act[j] = active;
        }
        // (MBD_TUNED_SPEC with contact_avg, links with several colliders: the average over a candidate's active contacts —
        // two or more; one: untouched — per half, the operations of rollout_kernel's SPEC form)
        if constexpr ((MBD_TUNED_SPEC & MBD_FLAG_CONTACT_AVG) != 0 && MAXCOL > 1) {
          int nx = 0, ny = 0;
#pragma unroll
          for (int j = 0; j < MAXCOL; ++j) { nx += con_act[j].x ? 1 : 0; ny += con_act[j].y ? 1 : 0; }
          const f2 inv_n = mk2(1.0f / (float)(nx > 1 ? nx : 1), 1.0f / (float)(ny > 1 ? ny : 1));
          cd_p = scale2(cd_p, inv_n);    // (inv_n is exactly 1 for a single contact: no select needed)
          cd_th = scale2(cd_th, inv_n);
        }
        p = add2(p, cd_p);
        r = qrotvec2_qm<QM>(r, cd_th, q_worst);
        Pp_next = shfl3x2(p, plane);  // consumed by stage (1) of the next substep
        Pr_next = shfl4x2(r, plane);
        shfl_issue();
      }
      // ---- (5) integrator.project_xd ------------------------------------------------------------------
      const v3x2 v_old = v, w_old = w;
      v = v3x2{(p.x - p_prev.x) * splat(inv_dt), (p.y - p_prev.y) * splat(inv_dt), (p.z - p_prev.z) * splat(inv_dt)};
      {
        const q4x2 dq = qmul2(r, conj2(r_prev));
        const f2 s = copysign2(splat(two_inv_dt), dq.w);
        w = v3x2{dq.x * s, dq.y * s, dq.z * s};
      }
      // ---- (6) collisions.resolve_velocity (Jacobi per link: every contact sees the velocities stage (5) left) ----
      const v3x2 v6 = v, w6 = w;
#pragma unroll
      for (int j = 0; j < MAXCOL; ++j) {
        // (SKIP6 of mbd_kernels.h: ant's second collider — the ankle end of a lower leg — rarely touches; the slot's whole
        // effect is two selects on its `active` flags, so skipping it when no lane of the wavefront has one set is exact)
        if constexpr (FAM == 1) {
          if (j > 0 && __builtin_expect(__builtin_amdgcn_ballot_w64(con_act[j].x || con_act[j].y) == 0ull, 1)) continue;
        }
        const v3x2 rc = sub2(con_pos[j], p);
        const v3x2 vpt = add2(v6, cross2(w6, rc));
        f2 vn_prev = splat(0.0f);
        if (elast != 0.0f) vn_prev = add2(v_old, cross2(w_old, rc)).z;
        const f2 vn = vpt.z;
        const f2 vtn = sqrt_floor2(fma2(vpt.x, vpt.x, vpt.y * vpt.y));
        const f2 inv = rcp_exact2(vtn + splat(1e-10f));
        const f2 dirx = vpt.x * inv, diry = vpt.y * inv;
        const f2 cnx = rc.y, cny = -rc.x;
        const v3x2 cdv = v3x2{-(rc.z * diry), rc.z * dirx, fma2(rc.x, diry, -(rc.y * dirx))};  // cross_bz0(rc, dir)
        const f2 icnx = cnx * splat(ic_ib), icny = cny * splat(ic_ib);
        const v3x2 icd = scale2s(cdv, ic_ib);
        const f2 wn = splat(ic_inv_mass) + fma2(cnx, icnx, cny * icny), wt = splat(ic_inv_mass) + dot2(cdv, icd);
        const f2 rest = splat(-elast) * vn_prev;
        const f2 dvn = fmaxs2(rest, 0.0f) - vn;
        const f2 jt_max = (splat(mu) * con_dlam[j]) * splat(inv_dt);
        const f2 dvt = fmin2(jt_max * wt, vtn);
        const f2 jn = div2_(dvn, wn), jt = -div2_pos_(dvt, wt);
        const v3x2 Pimp = v3x2{dirx * jt, diry * jt, jn};
        const v3x2 nv = axpy2(splat(ic_inv_mass), Pimp, v);
        const v3x2 nw = add2(w, scale2s(cross2(rc, Pimp), ic_ib));
        v = sel3(con_act[j], nv, v);
        w = sel3(con_act[j], nw, w);
      }
      if constexpr ((MBD_TUNED_SPEC & MBD_FLAG_CONTACT_AVG) != 0 && MAXCOL > 1) {  // the average of the velocity changes: v6 + (v - v6) / n
        int nx = 0, ny = 0;
#pragma unroll
        for (int j = 0; j < MAXCOL; ++j) { nx += con_act[j].x ? 1 : 0; ny += con_act[j].y ? 1 : 0; }
        const f2 inv_n = mk2(1.0f / (float)(nx > 1 ? nx : 1), 1.0f / (float)(ny > 1 ? ny : 1));
        const b2 many{nx >= 2, ny >= 2};
        const v3x2 av_ = v3x2{fma2(v.x - v6.x, inv_n, v6.x), fma2(v.y - v6.y, inv_n, v6.y), fma2(v.z - v6.z, inv_n, v6.z)};
        const v3x2 aw_ = v3x2{fma2(w.x - w6.x, inv_n, w6.x), fma2(w.y - w6.y, inv_n, w6.y), fma2(w.z - w6.z, inv_n, w6.z)};
        v = sel3(many, av_, v);
        w = sel3(many, aw_, w);
      }
    };
    auto substep = [&]() __attribute__((always_inline)) { substep_qm(std::integral_constant<int, QM_FAST>{}); };
    if constexpr (NFR > 1) {
      for (int it = 0; it < 2; ++it) repeat_n<NFR / 2>(substep);
      if constexpr (NFR % 2 != 0) substep();
    } else {
      int fr = 0;
      for (; fr + 1 < nfr; fr += 2) { substep(); substep(); }
      if (fr < nfr) substep();
    }
    if constexpr (QM_FAST == 1) {
      if (__builtin_expect(__builtin_amdgcn_fcmpf(q_worst, 0.05f, 2 /* ogt */) != 0ull, 0)) {
        p = s_p; r = s_r; v = s_v; w = s_w;
        Pp_next = shfl3x2(p, plane);
        Pr_next = shfl4x2(r, plane);
        for (int fr = 0; fr < nfr; ++fr) substep_qm(std::integral_constant<int, 2>{});
      }
    }

    // ---- reward and tracked positions ------------------------------------------------------------------
    const v3x2 o1 = sub2(p, rot2(bcast3(com), r));
    {
      f2 rew;
      if (rkind == MBD_REW_HUMANOIDRUN) {
        rew = o1.x * splat(1.0f) - fclip2(fabs2(o1.z - splat(1.3f)), -1.0f, 1.0f) * splat(1.0f) - fabs2(o1.y) * splat(0.1f);
      } else if (rkind == MBD_REW_HUMANOIDSTANDUP) {
        rew = splat(1.5f) - fclip2(fabs2(o1.z - splat(1.3f)), -2.0f, 1.0f) - fabs2(o1.x) * splat(0.1f) - fabs2(o1.y) * splat(0.1f);
      } else if (rkind == MBD_REW_ANT) {  // (rollout_kernel's expression per half: the divisions are IEEE)
        const bool always = Mg->reward_params[5] != 0.0f;
        const float zlo = Mg->reward_params[2], zhi = Mg->reward_params[3], hv = Mg->reward_params[4];
        const f2 healthy = mk2((always || (o1.z.x >= zlo && o1.z.x <= zhi)) ? hv : 0.0f,
                               (always || (o1.z.y >= zlo && o1.z.y <= zhi)) ? hv : 0.0f);
        const f2 fwd = mk2((o1.x.x - o0.x.x) / dt_ctrl, (o1.x.y - o0.x.y) / dt_ctrl);
        rew = (splat(rp0) * fwd + healthy) - splat(rp1) * ctrl_cost;
      } else {  // MBD_REW_HUMANOIDTRACK: the reward of the INCOMING state (humanoidtrack.py:78)
        rew = splat(1.0f) + (-fabs2(v0.x - splat(1.6f)) - fabs2(o0.z - splat(1.3f)) - fabs2(o0.y) * splat(0.1f));
      }
      rew_sum = rew_sum + rew;
      if (__builtin_expect(root_lane && P.rewss != nullptr, 1)) {
        if (okA) P.rewss[(size_t)bA * H + t] = rew.x;
        if (okB) P.rewss[(size_t)bB * H + t] = rew.y;
      }
    }
    if (__builtin_expect(P.xpos != nullptr && track_k >= 0, RK == MBD_REW_HUMANOIDTRACK)) {
      if (okA) {
        float* o = P.xpos + (((size_t)bA * H + t) * K + track_k) * 3;
        o[0] = o1.x.x; o[1] = o1.y.x; o[2] = o1.z.x;
      }
      if (okB) {
        float* o = P.xpos + (((size_t)bB * H + t) * K + track_k) * 3;
        o[0] = o1.x.y; o[1] = o1.y.y; o[2] = o1.z.y;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { u_rot[k] = un_rot[k]; y_rot[k] = yn_rot[k]; }
#pragma unroll
    for (int k = 0; k < KCC; ++k) { cc_u[k] = ccn_u[k]; cc_y[k] = ccn_y[k]; }
  }  // control steps
  if (root_lane && P.rews) {
    if (okA) P.rews[bA] = rew_sum.x / (float)H;
    if (okB) P.rews[bB] = rew_sum.y / (float)H;
  }
  if (P.state_final && link_ok) {
    if (okA) {
      float* o = P.state_final + ((size_t)bA * L + l) * MBD_LINK_STATE;
      o[0] = p.x.x; o[1] = p.y.x; o[2] = p.z.x; o[3] = r.w.x; o[4] = r.x.x; o[5] = r.y.x; o[6] = r.z.x;
      o[7] = v.x.x; o[8] = v.y.x; o[9] = v.z.x; o[10] = w.x.x; o[11] = w.y.x; o[12] = w.z.x;
    }
    if (okB) {
      float* o = P.state_final + ((size_t)bB * L + l) * MBD_LINK_STATE;
      o[0] = p.x.y; o[1] = p.y.y; o[2] = p.z.y; o[3] = r.w.y; o[4] = r.x.y; o[5] = r.y.y; o[6] = r.z.y;
      o[7] = v.x.y; o[8] = v.y.y; o[9] = v.z.y; o[10] = w.x.y; o[11] = w.y.y; o[12] = w.z.y;
    }
  }
}

}  // namespace mbd
