// mbd_kernels.h — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the reverse-diffusion hot path
// of mbd/planners/mbd_planner.py:97-135:
//
//   sample_kernel        A1  eps -> Y0s = clip(eps*sigma_i + Ybar_i, -1, 1)        (:103-106)
//   noise_kernel / shift_kernel  the same in two halves: the normals of the NEXT step generated behind the current
//                        rollout (they depend on the step's key only), then only shifted by Ybar_i
//   rollout_kernel       A2/A3  vmap(rollout_us) of the positional rigid-body step (:109, utils.py:14-20)
//   car2d_rollout_kernel A2/A3  the same for the in-tree car2d env                  (car2d.py:77-93)
//   logpd_*_kernel       A5  eval_xref_logpd                                        (:118)
//   score_kernel         A4-A6  standardise, demo blend, softmax                    (:110-127)
//   wmean_kernel         A7-A8  einsum("n,nij->ij") + score update                  (:128-133)
//   wmean_partial_kernel / wmean_finish_kernel  the same for plans of >= 4096 candidates (row-major reads)
//
// Layout of the rollout kernel (the one that matters): ONE LINK PER LANE.  A candidate occupies LPS
// consecutive lanes of a wavefront (LPS = 16 for the 11-link humanoid, 8 for the 7-link cheetah, 4 for
// the hopper), 64/LPS candidates per wavefront; a workgroup is four INDEPENDENT wavefronts (that is how the
// dispatcher puts one wavefront on each SIMD of a CU), so N=1024 humanoid candidates are 256 wavefronts on
// 64 CUs.  The 13-float link state, its previous pose and ~90 per-link model constants stay in VGPRs for
// the whole H x n_frames rollout; a parent's state and the children's constraint contributions move
// between lanes by DPP row shifts when the link tree fits one of the instantiated layouts (every built-in
// model), by ds_bpermute otherwise (no LDS data, no barriers); HBM is touched only for the action fetch
// (prefetched one control step ahead) and the reward store.  All arithmetic follows mbd_math.h.
#pragma once

#include "../../include/mbd_hip.h"
#include "mbd_math.h"

namespace mbd {

constexpr int kMaxChildren = 4;

struct RolloutParams {
  const mbd_model_t* model;  // device copy of the compiled model
  const float* state0;       // [L][13]
  const float* us;           // [B][H][Nu]
  float* rewss;              // [B][H] or nullptr
  float* rews;               // [B] or nullptr: mean over H
  float* xpos;               // [B][H][K][3] or nullptr
  float* state_final;        // [B][L][13] or nullptr
  int B, H;
  int slide_limits;  // any slide dof with a finite range (wave-uniform: the limit corrections are skipped otherwise)
  int max_children;  // largest child count in the model (wave-uniform bound of the generic kernels' child loops)
  int max_rot;       // largest number of hinge dofs on one joint (1: hopper, walker2d, halfcheetah, ant, cartpole)
  int any_stiff;     // some hinge has a joint spring (rot_stiff != 0): wave-uniform, the planar kernels skip the
                     // stage-(1) hinge angle otherwise (it only feeds the spring)
  int has_weld;      // some joint has no hinge dof (slide-only / weld: the cartpole's cart): wave-uniform, the orientation
                     // lock of stage (3) is skipped otherwise (it would be discarded by its select)
  // DPP instantiations only: lane (within the 16-lane row) <-> link tables, [0..15] lane -> link (-1: padding),
  // [16..31] link -> lane.  Device memory, written by the host when the model's tree fits the shift pattern.
  const signed char* lane_tab;
  unsigned long long* dbg_clock;  // nullptr, or [grid][3] = (start tick, end tick, HW_ID|XCC<<32) (tools/probes)
};

__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
// One full LDS-counter wait after a batch of shuffles instead of a partial s_waitcnt before every consumer
// (with one wavefront per SIMD every s_waitcnt costs a 4-cycle issue slot).  ds_bpermute is not a memory
// operation to the machine scheduler, so without the two scheduling barriers it moves shuffles across the wait.
__device__ __forceinline__ void shfl_join() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
}
// "issue what was requested so far, now": keeps the scheduler from sinking a batch of shuffles towards its
// consumers, so that the LDS round trip overlaps the independent work that follows in program order
__device__ __forceinline__ void shfl_issue() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ v3 shfl3(v3 v, int src) { return v3{shfl(v.x, src), shfl(v.y, src), shfl(v.z, src)}; }
__device__ __forceinline__ q4 shfl4(q4 q, int src) {
  return q4{shfl(q.w, src), shfl(q.x, src), shfl(q.y, src), shfl(q.z, src)};
}

// ---- lane exchange without the LDS: DPP row shifts --------------------------------------------------------
// A candidate occupies one 16-lane DPP row.  When the link tree can be laid out so that every s-th child (in link
// order) sits at lane(parent) - DS for a fixed shift DS per slot s, parent<->child traffic is a row shift executed
// by the VALU (v_mul_f32_dpp): no LDS issue slots (a ds_bpermute_b32 costs two), no round trip to wait for.
// dpp_from<K>(x): the value of x in lane i+K of the same row (0 when that lane is outside the row).
template <int K>
__device__ __forceinline__ float dpp_from(float x) {
  static_assert(K != 0 && K > -16 && K < 16, "row shift");
  constexpr int ctrl = K > 0 ? 0x100 + K /* row_shl:K */ : 0x110 - K /* row_shr:-K */;
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true));
}
// v_fmac_f32 with a DPP source is not something the compiler forms (it keeps v_mov_b32_dpp + v_fmac_f32), so the
// accumulating forms are written out.  acc += x(lane i+K) * m is bit-identical to an add when m is 1.0f and a
// no-op when m is 0.0f.  The leading "s_nop 1" covers the hazard of a DPP read within two wait states of a VALU
// write of the same register, which the compiler cannot see inside an asm block; the accumulators are only ever
// consumed by ordinary VALU instructions.
// (the modifier text must be a literal inside the asm string: one specialisation per shift in use)
template <int K>
__device__ __forceinline__ void dpp_acc6(v3& a, v3& b, v3 x, v3 y, float m);
#define MBD_DPP_ACC6(K, MOD)                                                                                  \
  template <>                                                                                                 \
  __device__ __forceinline__ void dpp_acc6<K>(v3 & a, v3 & b, v3 x, v3 y, float m) {                          \
    asm("s_nop 1\n\t"                                                                                         \
        "v_fmac_f32_dpp %0, %6, %12 " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
        "v_fmac_f32_dpp %1, %7, %12 " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
        "v_fmac_f32_dpp %2, %8, %12 " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
        "v_fmac_f32_dpp %3, %9, %12 " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
        "v_fmac_f32_dpp %4, %10, %12 " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
        "v_fmac_f32_dpp %5, %11, %12 " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1"                          \
        : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z)                                    \
        : "v"(x.x), "v"(x.y), "v"(x.z), "v"(y.x), "v"(y.y), "v"(y.z), "v"(m));                                \
  }
MBD_DPP_ACC6(-1, "row_shr:1")
MBD_DPP_ACC6(4, "row_shl:4")
MBD_DPP_ACC6(6, "row_shl:6")
MBD_DPP_ACC6(3, "row_shl:3")
MBD_DPP_ACC6(2, "row_shl:2")
#undef MBD_DPP_ACC6
// The three child slots of the humanoid layout (shifts -1, +4, +6) in ONE block: one hazard s_nop for 18 accumulations
// (an "s_nop 1" costs a lone wavefront two whole issue slots, tools/probes/probe_issue.hip).
#define MBD_F6(MOD, M)                                                                                        \
  "v_fmac_f32_dpp %0, %6, %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
  "v_fmac_f32_dpp %1, %7, %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
  "v_fmac_f32_dpp %2, %8, %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
  "v_fmac_f32_dpp %3, %9, %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
  "v_fmac_f32_dpp %4, %10, %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                       \
  "v_fmac_f32_dpp %5, %11, %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void dpp_acc6x3(v3& a, v3& b, v3 x, v3 y, float m0, float m1, float m2) {
  asm("s_nop 1\n\t" MBD_F6("row_shr:1", 12) MBD_F6("row_shl:4", 13) MBD_F6("row_shl:6", 14)
      : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z)
      : "v"(x.x), "v"(x.y), "v"(x.z), "v"(y.x), "v"(y.y), "v"(y.z), "v"(m0), "v"(m1), "v"(m2));
}
// The same with two independent products t0 = f * u0, t1 = f * u1 riding in place of the s_nop: they are the two
// wait states between any earlier write of x / y and the first DPP read (t0, t1 are early-clobber outputs, so they
// never alias a DPP source), and they are work the caller needs anyway (v_mul_f32 rounds like the compiler's product).
__device__ __forceinline__ void dpp_acc6x3_mul2(v3& a, v3& b, v3 x, v3 y, float m0, float m1, float m2, float f,
                                                float u0, float u1, float& t0, float& t1) {
  asm("v_mul_f32_e32 %6, %17, %18\n\tv_mul_f32_e32 %7, %17, %19\n\t"
      "v_fmac_f32_dpp %0, %8, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %9, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %2, %10, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %3, %11, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %4, %12, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %5, %13, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %8, %15 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %9, %15 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %2, %10, %15 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %3, %11, %15 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %4, %12, %15 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %5, %13, %15 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %8, %16 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %9, %16 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %2, %10, %16 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %3, %11, %16 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %4, %12, %16 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %5, %13, %16 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z), "=&v"(t0), "=&v"(t1)
      : "v"(x.x), "v"(x.y), "v"(x.z), "v"(y.x), "v"(y.y), "v"(y.z), "v"(m0), "v"(m1), "v"(m2), "v"(f), "v"(u0), "v"(u1));
}
#undef MBD_F6
// the parent's pose for the s-th child (mask m_s = 1): seven values, r = x(i+K0) m0 + x(i+K1) m1 + x(i+K2) m2
template <int K0, int K1, int K2>
__device__ __forceinline__ void dpp_fetch7(v3 p, q4 r, float m0, float m1, float m2, v3& Pp, q4& Pr);
#define MBD_DPP_F(R, X, M, MOD) "v_fmac_f32_dpp %" #R ", %" #X ", %" #M " " MOD " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define MBD_DPP_FETCH7(K0, K1, K2, MOD1, MOD2)                                                                \
  template <>                                                                                                 \
  __device__ __forceinline__ void dpp_fetch7<K0, K1, K2>(v3 p, q4 r, float m0, float m1, float m2, v3 & Pp,   \
                                                         q4 & Pr) {                                           \
    float o0 = dpp_from<K0>(p.x) * m0, o1 = dpp_from<K0>(p.y) * m0, o2 = dpp_from<K0>(p.z) * m0;              \
    float o3 = dpp_from<K0>(r.w) * m0, o4 = dpp_from<K0>(r.x) * m0, o5 = dpp_from<K0>(r.y) * m0;              \
    float o6 = dpp_from<K0>(r.z) * m0;                                                                        \
    asm("s_nop 1\n\t" MBD_DPP_F(0, 7, 14, MOD1) MBD_DPP_F(1, 8, 14, MOD1) MBD_DPP_F(2, 9, 14, MOD1)           \
            MBD_DPP_F(3, 10, 14, MOD1) MBD_DPP_F(4, 11, 14, MOD1) MBD_DPP_F(5, 12, 14, MOD1)                  \
                MBD_DPP_F(6, 13, 14, MOD1) MBD_DPP_F(0, 7, 15, MOD2) MBD_DPP_F(1, 8, 15, MOD2)                \
                    MBD_DPP_F(2, 9, 15, MOD2) MBD_DPP_F(3, 10, 15, MOD2) MBD_DPP_F(4, 11, 15, MOD2)           \
                        MBD_DPP_F(5, 12, 15, MOD2) MBD_DPP_F(6, 13, 15, MOD2)                                 \
        : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+v"(o6)                                \
        : "v"(p.x), "v"(p.y), "v"(p.z), "v"(r.w), "v"(r.x), "v"(r.y), "v"(r.z), "v"(m1), "v"(m2));            \
    Pp = v3{o0, o1, o2};                                                                                      \
    Pr = q4{o3, o4, o5, o6};                                                                                  \
  }
MBD_DPP_FETCH7(1, -4, -6, "row_shr:4", "row_shr:6")
MBD_DPP_FETCH7(1, -2, -4, "row_shr:2", "row_shr:4")
#undef MBD_DPP_FETCH7
// a fourth slot (ant: four legs on the torso): one more masked term on top of dpp_fetch7
template <int K>
__device__ __forceinline__ void dpp_fmac7(v3 p, q4 r, float m, v3& Pp, q4& Pr);
template <>
__device__ __forceinline__ void dpp_fmac7<-6>(v3 p, q4 r, float m, v3& Pp, q4& Pr) {
  asm("s_nop 1\n\t" MBD_DPP_F(0, 7, 14, "row_shr:6") MBD_DPP_F(1, 8, 14, "row_shr:6") MBD_DPP_F(2, 9, 14, "row_shr:6")
          MBD_DPP_F(3, 10, 14, "row_shr:6") MBD_DPP_F(4, 11, 14, "row_shr:6") MBD_DPP_F(5, 12, 14, "row_shr:6")
              MBD_DPP_F(6, 13, 14, "row_shr:6")
      : "+v"(Pp.x), "+v"(Pp.y), "+v"(Pp.z), "+v"(Pr.w), "+v"(Pr.x), "+v"(Pr.y), "+v"(Pr.z)
      : "v"(p.x), "v"(p.y), "v"(p.z), "v"(r.w), "v"(r.x), "v"(r.y), "v"(r.z), "v"(m));
}
// trees with at most two children per link (walker2d, halfcheetah: D = (+1, -3)) and chains (hopper, cartpole)
template <>
__device__ __forceinline__ void dpp_fetch7<1, -3, 0>(v3 p, q4 r, float m0, float m1, float, v3& Pp, q4& Pr) {
  float o0 = dpp_from<1>(p.x) * m0, o1 = dpp_from<1>(p.y) * m0, o2 = dpp_from<1>(p.z) * m0;
  float o3 = dpp_from<1>(r.w) * m0, o4 = dpp_from<1>(r.x) * m0, o5 = dpp_from<1>(r.y) * m0;
  float o6 = dpp_from<1>(r.z) * m0;
  asm("s_nop 1\n\t" MBD_DPP_F(0, 7, 14, "row_shr:3") MBD_DPP_F(1, 8, 14, "row_shr:3") MBD_DPP_F(2, 9, 14, "row_shr:3")
          MBD_DPP_F(3, 10, 14, "row_shr:3") MBD_DPP_F(4, 11, 14, "row_shr:3") MBD_DPP_F(5, 12, 14, "row_shr:3")
              MBD_DPP_F(6, 13, 14, "row_shr:3")
      : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+v"(o6)
      : "v"(p.x), "v"(p.y), "v"(p.z), "v"(r.w), "v"(r.x), "v"(r.y), "v"(r.z), "v"(m1));
  Pp = v3{o0, o1, o2};
  Pr = q4{o3, o4, o5, o6};
}
template <>
__device__ __forceinline__ void dpp_fetch7<1, 0, 0>(v3 p, q4 r, float m0, float, float, v3& Pp, q4& Pr) {
  Pp = v3{dpp_from<1>(p.x) * m0, dpp_from<1>(p.y) * m0, dpp_from<1>(p.z) * m0};
  Pr = q4{dpp_from<1>(r.w) * m0, dpp_from<1>(r.x) * m0, dpp_from<1>(r.y) * m0, dpp_from<1>(r.z) * m0};
}

template <bool ISO>
struct Inert {
  float inv_mass;
  float ib[ISO ? 1 : 6];
};
// World-frame inverse inertia W = R Ib R^T (xx yy zz xy xz yz) of a link at orientation r: T = R Ib, then
// the six unique entries of T R^T.  Refreshed at the head of every stage that applies it — (1), (3), (4),
// (6) — and applied as a symmetric 3x3 product (9 FMAs instead of two quaternion rotations around the
// body-frame product).  Isotropic models (ib0 * identity) carry nothing.
template <bool ISO>
struct WInert {
  float w[ISO ? 1 : 6];
};
// DIAG: every body-frame tensor of the model is exactly diagonal (axis-aligned capsules / boxes): the products
// with the zero off-diagonal entries are dropped — same values, 27 instructions fewer per tensor.
// AXI: every body-frame tensor is axisymmetric about one of the link's axes — diag with two equal entries (capsules:
// hopper, walker2d): Ib = a Id + (c - a) u u^T, so the world tensor is a Id + (c - a) U U^T with U = R u.  Only U is
// rebuilt per stage (one rotation of a constant instead of R Ib R^T) and the tensor is applied as
// a v + (c - a)(U.v) U — a specification of its own (not the same roundings as R Ib R^T), shared with the checker.
// The lane constants are remapped once: ib = (a, a, c, u.x, u.y, u.z).
template <bool ISO>
__device__ __forceinline__ float axi_k(const Inert<ISO>& in) { return in.ib[ISO ? 0 : 2] - in.ib[0]; }
template <bool ISO>
__device__ __forceinline__ void axi_remap(Inert<ISO>& in) {
  if constexpr (!ISO) {
    const float xx = in.ib[0], yy = in.ib[1], zz = in.ib[2];
    const bool ez = xx == yy, ex = !ez && yy == zz;  // otherwise xx == zz: the axis is y
    const float a = ez ? xx : (ex ? yy : xx), c = ez ? zz : (ex ? xx : yy);
    in.ib[0] = a; in.ib[1] = a; in.ib[2] = c;
    in.ib[3] = ex ? 1.0f : 0.0f; in.ib[4] = (!ez && !ex) ? 1.0f : 0.0f; in.ib[5] = ez ? 1.0f : 0.0f;
  }
}
template <bool ISO, bool DIAG, bool AXI = false>
__device__ __forceinline__ WInert<ISO> world_inertia(const Inert<ISO>& in, q4 r) {
  WInert<ISO> W;
  if constexpr (ISO) {
    W.w[0] = 0.0f;
  } else if constexpr (AXI) {
    const v3 U = rot(v3{in.ib[3], in.ib[4], in.ib[5]}, r);
    W.w[0] = U.x; W.w[1] = U.y; W.w[2] = U.z;
    W.w[3] = W.w[4] = W.w[5] = 0.0f;
  } else if constexpr (DIAG) {
    const axes3 A = qaxes(r);
    const float xx = in.ib[0], yy = in.ib[1], zz = in.ib[2];
    const v3 T0 = v3{A.X.x * xx, A.Y.x * yy, A.Z.x * zz}, T1 = v3{A.X.y * xx, A.Y.y * yy, A.Z.y * zz};
    const v3 T2 = v3{A.X.z * xx, A.Y.z * yy, A.Z.z * zz};
    auto ent = [&](v3 T, float X, float Y, float Z) { return ffma(T.z, Z, ffma(T.y, Y, T.x * X)); };
    W.w[0] = ent(T0, A.X.x, A.Y.x, A.Z.x);
    W.w[1] = ent(T1, A.X.y, A.Y.y, A.Z.y);
    W.w[2] = ent(T2, A.X.z, A.Y.z, A.Z.z);
    W.w[3] = ent(T0, A.X.y, A.Y.y, A.Z.y);
    W.w[4] = ent(T0, A.X.z, A.Y.z, A.Z.z);
    W.w[5] = ent(T1, A.X.z, A.Y.z, A.Z.z);
  } else {
    const axes3 A = qaxes(r);
    const float xx = in.ib[0], yy = in.ib[1], zz = in.ib[2], xy = in.ib[3], xz = in.ib[4], yz = in.ib[5];
    auto row = [&](float X, float Y, float Z) {  // row i of T = R Ib
      return v3{ffma(Z, xz, ffma(Y, xy, X * xx)), ffma(Z, yz, ffma(Y, yy, X * xy)), ffma(Z, zz, ffma(Y, yz, X * xz))};
    };
    const v3 T0 = row(A.X.x, A.Y.x, A.Z.x), T1 = row(A.X.y, A.Y.y, A.Z.y), T2 = row(A.X.z, A.Y.z, A.Z.z);
    auto ent = [&](v3 T, float X, float Y, float Z) { return ffma(T.z, Z, ffma(T.y, Y, T.x * X)); };
    W.w[0] = ent(T0, A.X.x, A.Y.x, A.Z.x);
    W.w[1] = ent(T1, A.X.y, A.Y.y, A.Z.y);
    W.w[2] = ent(T2, A.X.z, A.Y.z, A.Z.z);
    W.w[3] = ent(T0, A.X.y, A.Y.y, A.Z.y);
    W.w[4] = ent(T0, A.X.z, A.Y.z, A.Z.z);
    W.w[5] = ent(T1, A.X.z, A.Y.z, A.Z.z);
  }
  return W;
}
// world-frame inverse inertia applied to v
template <bool ISO, bool AXI = false>
__device__ __forceinline__ v3 iinv(const Inert<ISO>& in, const WInert<ISO>& W, v3 v) {
  if constexpr (ISO) {
    return scale(v, in.ib[0]);
  } else if constexpr (AXI) {
    const v3 Z = v3{W.w[0], W.w[1], W.w[2]};
    const float kd = axi_k<ISO>(in) * dot(Z, v);
    return v3{ffma(kd, Z.x, in.ib[0] * v.x), ffma(kd, Z.y, in.ib[0] * v.y), ffma(kd, Z.z, in.ib[0] * v.z)};
  } else {
    v3 m;
    m.x = ffma(W.w[4], v.z, ffma(W.w[3], v.y, W.w[0] * v.x));
    m.y = ffma(W.w[5], v.z, ffma(W.w[1], v.y, W.w[3] * v.x));
    m.z = ffma(W.w[2], v.z, ffma(W.w[5], v.y, W.w[4] * v.x));
    return m;
  }
}

struct JointFrames {
  v3 ap, ac;
  v3x2 anchor;  // (ap, ac) packed
  v3x2 arm;     // (rp, rc): the anchor offsets rotated into the world — the lever arms of the joint's impulses
  q4 aprot, acrot;
  v3 Xp, Xc, Yc, Zc, ax1;
  float ang0, ang1, ang2;
};

struct JointConst {
  v3 ap_pos, ac_pos;
  q4 ap_rot, ac_rot;
};

// multi (wave-uniform): some joint of the model has more than one hinge dof.  Single-hinge models only ever use
// the first Euler angle and axis; the other two would be masked to exact zeros downstream.
__device__ __forceinline__ JointFrames joint_frames(const JointConst& jc, v3 Pp, q4 Pr, v3 Cp, q4 Cr, bool multi) {
  JointFrames f;
  // parent side in the low halves, child side in the high halves of packed pairs
  const q4x2 R2 = pack4(Pr, Cr);
  const v3x2 arm = rot2(pack3(jc.ap_pos, jc.ac_pos), R2);
  const v3x2 anchor = add2(pack3(Pp, Cp), arm);
  f.arm = arm;
  const q4x2 arot = qmul2(R2, pack4(jc.ap_rot, jc.ac_rot));
  const axes3x2 AX = qaxes2(arot);
  f.ap = lo3(anchor); f.ac = hi3(anchor);
  f.anchor = anchor;
  f.aprot = lo4(arot);
  f.acrot = q4{arot.w.y, arot.x.y, arot.y.y, arot.z.y};
  struct { v3 X, Y, Z; } A{lo3(AX.X), lo3(AX.Y), lo3(AX.Z)}, C{hi3(AX.X), hi3(AX.Y), hi3(AX.Z)};
  f.Xp = A.X; f.Xc = C.X; f.Yc = C.Y; f.Zc = C.Z;
  // sin b = Zc.Xp; (sin a, cos a) and (sin c, cos c) both have length cos b: one reciprocal for all
  // (opaque: keeps the SLP vectoriser from pairing these five dot products across register pairs, which
  // costs more v_mov than the packed ops save — 15 instructions per substep)
  auto opq = [](float x) { asm("" : "+v"(x)); return x; };
  float sb = fclip(opq(dot(C.Z, A.X)), -1.0f, 1.0f);
  float cb2 = ffma(-sb, sb, 1.0f);
  float cb = sqrt_floor(cb2);
  float inv = div_(1.0f, cb + 1e-10f);
  if (multi) {
    const f2 a02 = angle_unit2(mk2(-opq(dot(C.Z, A.Y)) * inv, -opq(dot(C.Y, A.X)) * inv),
                               mk2(opq(dot(C.Z, A.Z)) * inv, opq(dot(C.X, A.X)) * inv));
    f.ang0 = a02.x;
    f.ang1 = angle_unit_cpos(sb, cb);
    f.ang2 = a02.y;
    v3 n = cross(C.Z, A.X);
    f.ax1 = scale(n, inv);
  } else {
    f.ang0 = angle_unit(-opq(dot(C.Z, A.Y)) * inv, opq(dot(C.Z, A.Z)) * inv);
    f.ang1 = 0.0f;
    f.ang2 = 0.0f;
    f.ax1 = mk3(0.0f, 0.0f, 0.0f);
  }
  return f;
}

// (parent, child) pair of world tensors and of inverse-inertia applications: low half the parent's, high half
// the child's
template <bool ISO>
struct WInert2 {
  f2 w[ISO ? 1 : 6];
};
template <bool ISO, bool DIAG, bool AXI = false>
__device__ __forceinline__ WInert2<ISO> world_inertia2(const Inert<ISO>& ip, const Inert<ISO>& ic, q4x2 R2) {
  WInert2<ISO> W;
  if constexpr (ISO) {
    W.w[0] = mk2(0.0f, 0.0f);
  } else if constexpr (AXI) {
    const v3x2 U = rot2(pack3(v3{ip.ib[3], ip.ib[4], ip.ib[5]}, v3{ic.ib[3], ic.ib[4], ic.ib[5]}), R2);
    W.w[0] = U.x; W.w[1] = U.y; W.w[2] = U.z;
    W.w[3] = W.w[4] = W.w[5] = mk2(0.0f, 0.0f);
  } else if constexpr (DIAG) {
    const axes3x2 A = qaxes2(R2);
    const f2 xx = mk2(ip.ib[0], ic.ib[0]), yy = mk2(ip.ib[1], ic.ib[1]), zz = mk2(ip.ib[2], ic.ib[2]);
    const v3x2 T0 = v3x2{A.X.x * xx, A.Y.x * yy, A.Z.x * zz}, T1 = v3x2{A.X.y * xx, A.Y.y * yy, A.Z.y * zz};
    const v3x2 T2 = v3x2{A.X.z * xx, A.Y.z * yy, A.Z.z * zz};
    auto ent = [&](v3x2 T, f2 X, f2 Y, f2 Z) { return fma2(T.z, Z, fma2(T.y, Y, T.x * X)); };
    W.w[0] = ent(T0, A.X.x, A.Y.x, A.Z.x);
    W.w[1] = ent(T1, A.X.y, A.Y.y, A.Z.y);
    W.w[2] = ent(T2, A.X.z, A.Y.z, A.Z.z);
    W.w[3] = ent(T0, A.X.y, A.Y.y, A.Z.y);
    W.w[4] = ent(T0, A.X.z, A.Y.z, A.Z.z);
    W.w[5] = ent(T1, A.X.z, A.Y.z, A.Z.z);
  } else {
    const axes3x2 A = qaxes2(R2);
    const f2 xx = mk2(ip.ib[0], ic.ib[0]), yy = mk2(ip.ib[1], ic.ib[1]), zz = mk2(ip.ib[2], ic.ib[2]);
    const f2 xy = mk2(ip.ib[3], ic.ib[3]), xz = mk2(ip.ib[4], ic.ib[4]), yz = mk2(ip.ib[5], ic.ib[5]);
    auto row = [&](f2 X, f2 Y, f2 Z) {
      return v3x2{fma2(Z, xz, fma2(Y, xy, X * xx)), fma2(Z, yz, fma2(Y, yy, X * xy)), fma2(Z, zz, fma2(Y, yz, X * xz))};
    };
    const v3x2 T0 = row(A.X.x, A.Y.x, A.Z.x), T1 = row(A.X.y, A.Y.y, A.Z.y), T2 = row(A.X.z, A.Y.z, A.Z.z);
    auto ent = [&](v3x2 T, f2 X, f2 Y, f2 Z) { return fma2(T.z, Z, fma2(T.y, Y, T.x * X)); };
    W.w[0] = ent(T0, A.X.x, A.Y.x, A.Z.x);
    W.w[1] = ent(T1, A.X.y, A.Y.y, A.Z.y);
    W.w[2] = ent(T2, A.X.z, A.Y.z, A.Z.z);
    W.w[3] = ent(T0, A.X.y, A.Y.y, A.Z.y);
    W.w[4] = ent(T0, A.X.z, A.Y.z, A.Z.z);
    W.w[5] = ent(T1, A.X.z, A.Y.z, A.Z.z);
  }
  return W;
}
template <bool ISO, bool AXI = false>
__device__ __forceinline__ v3x2 iinv2(const Inert<ISO>& ip, const Inert<ISO>& ic, const WInert2<ISO>& W, v3x2 v) {
  if constexpr (ISO) {
    return scale2(v, mk2(ip.ib[0], ic.ib[0]));
  } else if constexpr (AXI) {
    const v3x2 Z = v3x2{W.w[0], W.w[1], W.w[2]};
    const f2 a = mk2(ip.ib[0], ic.ib[0]);
    const f2 kd = mk2(axi_k<ISO>(ip), axi_k<ISO>(ic)) * dot2(Z, v);
    return v3x2{fma2(kd, Z.x, a * v.x), fma2(kd, Z.y, a * v.y), fma2(kd, Z.z, a * v.z)};
  } else {
    v3x2 m;
    m.x = fma2(W.w[4], v.z, fma2(W.w[3], v.y, W.w[0] * v.x));
    m.y = fma2(W.w[5], v.z, fma2(W.w[1], v.y, W.w[3] * v.x));
    m.z = fma2(W.w[2], v.z, fma2(W.w[5], v.y, W.w[4] * v.x));
    return m;
  }
}
// one angular positional correction (rotate child by +e, parent by -e): I^-1 e * |e|^2 /
// (e.I_p^-1 e + e.I_c^-1 e) — one division, no square root.  Split in two so that independent corrections
// share a packed division: prepare -> (I^-1 e pair, numerator, denominator); the caller divides; apply.
struct AngPrep {
  v3x2 in2;  // (I_p^-1 e, I_c^-1 e)
  float num, den;
};
template <bool ISO, bool AXI = false>
__device__ __forceinline__ AngPrep ang_prepare(v3 e, const Inert<ISO>& ip, const Inert<ISO>& ic,
                                               const WInert2<ISO>& W2) {
  AngPrep a;
  v3x2 e2 = bcast3(e);
  a.in2 = iinv2<ISO, AXI>(ip, ic, W2, e2);
  f2 d2 = dot2(e2, a.in2);
  a.den = (d2.x + d2.y) + 1e-20f;
  a.num = dot(e, e);
  return a;
}
__device__ __forceinline__ void ang_apply(const AngPrep& a, float quot, float sc, v3x2& dth2) {
  float g = quot * sc;
  dth2 = axpy2(mk2(-g, g), a.in2, dth2);
}
// contact normal = +z of the floor plane
__device__ __forceinline__ v3 crossz(v3 a) { return v3{a.y, -a.x, 0.0f}; }
// the same for vectors whose z component is an exact zero: value-identical to the generic helpers, minus the
// products with that zero
__device__ __forceinline__ float dot_az0(v3 a, v3 b) { return ffma(a.x, b.x, a.y * b.y); }
__device__ __forceinline__ v3 cross_bz0(v3 a, v3 b) {
  return v3{-(a.z * b.y), a.z * b.x, ffma(a.x, b.y, -(a.y * b.x))};
}
template <bool ISO, bool AXI = false>
__device__ __forceinline__ v3 iinv_z0(const Inert<ISO>& in, const WInert<ISO>& W, v3 v) {
  if constexpr (ISO) {
    return v3{v.x * in.ib[0], v.y * in.ib[0], 0.0f};
  } else if constexpr (AXI) {
    const float kd = axi_k<ISO>(in) * ffma(W.w[0], v.x, W.w[1] * v.y);
    return v3{ffma(kd, W.w[0], in.ib[0] * v.x), ffma(kd, W.w[1], in.ib[0] * v.y), kd * W.w[2]};
  } else {
    return v3{ffma(W.w[3], v.y, W.w[0] * v.x), ffma(W.w[1], v.y, W.w[3] * v.x), ffma(W.w[5], v.y, W.w[4] * v.x)};
  }
}

// ---- two colliders of ONE link at once (stage (4) is a Jacobi solve: every contact sees the same pose): the same
// primitives on (collider 0, collider 1) pairs, component-wise identical roundings
__device__ __forceinline__ q4x2 bcast4(q4 q) { return q4x2{mk2(q.w, q.w), mk2(q.x, q.x), mk2(q.y, q.y), mk2(q.z, q.z)}; }
__device__ __forceinline__ v3x2 irot_z2(f2 d, q4x2 q) {
  f2 a = q.y * d, b = q.x * d;
  f2 tx = -(a + a), ty = b + b;
  f2 cx = q.z * ty, cy = -(q.z * tx), cz = fma2(-q.x, ty, q.y * tx);
  return v3x2{fma2(q.w, tx, cx), fma2(q.w, ty, cy), d + cz};
}
template <bool ISO, bool AXI = false>
__device__ __forceinline__ v3x2 iinv_s2(const Inert<ISO>& in, const WInert<ISO>& W, v3x2 v) {
  if constexpr (ISO) {
    return scale2(v, mk2(in.ib[0], in.ib[0]));
  } else if constexpr (AXI) {
    const v3x2 Z = bcast3(v3{W.w[0], W.w[1], W.w[2]});
    const f2 a = mk2(in.ib[0], in.ib[0]), k = mk2(axi_k<ISO>(in), axi_k<ISO>(in));
    const f2 kd = k * dot2(Z, v);
    return v3x2{fma2(kd, Z.x, a * v.x), fma2(kd, Z.y, a * v.y), fma2(kd, Z.z, a * v.z)};
  } else {
    auto b = [&](int k) { return mk2(W.w[k], W.w[k]); };
    v3x2 m;
    m.x = fma2(b(4), v.z, fma2(b(3), v.y, b(0) * v.x));
    m.y = fma2(b(5), v.z, fma2(b(1), v.y, b(3) * v.x));
    m.z = fma2(b(2), v.z, fma2(b(5), v.y, b(4) * v.x));
    return m;
  }
}
template <bool ISO, bool AXI = false>
__device__ __forceinline__ v3x2 iinv_z0_s2(const Inert<ISO>& in, const WInert<ISO>& W, v3x2 v) {
  if constexpr (ISO) {
    const f2 ib = mk2(in.ib[0], in.ib[0]);
    return v3x2{v.x * ib, v.y * ib, mk2(0.0f, 0.0f)};
  } else if constexpr (AXI) {
    const f2 Zx = mk2(W.w[0], W.w[0]), Zy = mk2(W.w[1], W.w[1]), Zz = mk2(W.w[2], W.w[2]);
    const f2 a = mk2(in.ib[0], in.ib[0]), k = mk2(axi_k<ISO>(in), axi_k<ISO>(in));
    const f2 kd = k * fma2(Zx, v.x, Zy * v.y);
    return v3x2{fma2(kd, Zx, a * v.x), fma2(kd, Zy, a * v.y), kd * Zz};
  } else {
    auto b = [&](int k) { return mk2(W.w[k], W.w[k]); };
    return v3x2{fma2(b(3), v.y, b(0) * v.x), fma2(b(1), v.y, b(3) * v.x), fma2(b(5), v.y, b(4) * v.x)};
  }
}
__device__ __forceinline__ f2 dot_az0_2(v3x2 a, v3x2 b) { return fma2(a.x, b.x, a.y * b.y); }
__device__ __forceinline__ v3x2 cross_bz0_2(v3x2 a, v3x2 b) {
  return v3x2{-(a.z * b.y), a.z * b.x, fma2(a.x, b.y, -(a.y * b.x))};
}

// LPS   lanes per candidate (power of two >= n_links)
// ISO   model-wide isotropic inverse inertia (spring_inertia_scale = 1 models: the humanoid)
// SLIDES any slide dof in the model (planar roots of hopper / halfcheetah)
// MAXCH max children of any link; MAXCOL max sphere colliders on any link
// D0..D3: DPP layout, lane(parent) = lane(s-th child) + Ds (D0 = 0: off; a trailing 0: the model has no such
// slot).  Groups of LPS lanes never straddle a 16-lane DPP row, and the 0/1 masks discard whatever a shift
// pulls in from a neighbouring candidate of the same row.
// NS    slide slots in use (the model's largest slide count, 1..3): the loops over slide dofs stop there
// SLIDEW every joint with a slide dof hangs off the WORLD (planar roots, the cartpole's cart): its parent-side joint
//       frame is the constant ap_rot (identity (x) ap_rot, up to the sign of zeros), so the world-frame slide axes
//       rot(slide_axis, aprot) are per-lane constants instead of a rotation per slot, stage and substep
template <int LPS, bool ISO, bool SLIDES, int MAXCH, int MAXCOL, int D0 = 0, int D1 = 0, int D2 = 0, int D3 = 0,
          bool DIAG = false, bool MULTI = true, int NS = 3, bool SLIDEW = false, bool AXI = false>
__global__ __launch_bounds__(256) void rollout_kernel(RolloutParams P) {
  constexpr bool DPP = D0 != 0;
  static_assert(!DPP || ((D1 != 0 || D2 == 0) && (D2 != 0 || D3 == 0) && (D3 == 0 || MAXCH >= 4)),
                "DPP layout: slots are filled in order");
  const mbd_model_t* __restrict__ M = P.model;
  const unsigned long long dbg_t0 = P.dbg_clock ? __builtin_amdgcn_s_memtime() : 0ull;
  const int lane = threadIdx.x & 63;
  const int base = lane & ~(LPS - 1);
  const int l_lane = lane & (LPS - 1);
  const int L = M->n_links;
  const int l_link = DPP ? (int)P.lane_tab[l_lane] : l_lane;  // the link this lane holds
  const bool link_ok = l_link >= 0 && l_link < L;
  const int l = link_ok ? l_link : 0;
  auto lane_of = [&](int link) { return base + (DPP ? (int)P.lane_tab[16 + link] : link); };
  const bool root_lane = link_ok && l == 0;  // the lane that owns link 0 (rewards, control cost)
  constexpr int SPW = 64 / LPS;
  // a workgroup is 1 or 4 INDEPENDENT wavefronts (no LDS, no barrier): four-wave workgroups are how a launch
  // of >= 1024 wavefronts gets one wavefront on every SIMD of a CU (dispatch)
  const int wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int b_raw = wave_id * SPW + lane / LPS;
  const bool b_ok = b_raw < P.B;
  const int b = b_ok ? b_raw : P.B - 1;
  const int H = P.H, Nu = M->n_act, nfr = M->n_frames, K = M->n_track;

  // ---- per-lane model constants -----------------------------------------------------------------------
  const int parent = M->parent[l];
  const int nr = M->n_rot[l];
  const bool is_joint = link_ok && nr >= 0;
  const int ns = SLIDES ? M->n_slide[l] : 0;
  const int plane = parent >= 0 ? lane_of(parent) : lane;  // lane holding the parent (self if world)
  const bool world_parent = parent < 0;
  Inert<ISO> ic, ip;
  ic.inv_mass = M->inv_mass[l];
  ip.inv_mass = world_parent ? 0.0f : M->inv_mass[parent >= 0 ? parent : 0];
#pragma unroll
  for (int k = 0; k < (ISO ? 1 : 6); ++k) {
    ic.ib[k] = M->inv_inertia[l][k];
    ip.ib[k] = world_parent ? 0.0f : M->inv_inertia[parent >= 0 ? parent : 0][k];
  }
  if constexpr (AXI) { axi_remap<ISO>(ic); axi_remap<ISO>(ip); }
  JointConst jc;
  jc.ap_pos = mk3(M->ap_pos[l][0], M->ap_pos[l][1], M->ap_pos[l][2]);
  jc.ac_pos = mk3(M->ac_pos[l][0], M->ac_pos[l][1], M->ac_pos[l][2]);
  jc.ap_rot = q4{M->ap_rot[l][0], M->ap_rot[l][1], M->ap_rot[l][2], M->ap_rot[l][3]};
  jc.ac_rot = q4{M->ac_rot[l][0], M->ac_rot[l][1], M->ac_rot[l][2], M->ac_rot[l][3]};
  // non-joint lanes (free root, padding) are masked through their scalars: every contribution they
  // compute is then an exact zero
  const float ang_damp = is_joint ? M->ang_damp[l] : 0.0f, vel_damp = is_joint ? M->vel_damp[l] : 0.0f;
  const int nr_eff = is_joint ? nr : -1;
  const int zero_lane = M->n_rot[0] < 0 ? lane_of(0) : (L < LPS ? base + L : -1);  // a lane contributing zeros
  const bool need_child_mask = !(M->n_rot[0] < 0) && !(L < LPS);           // wave-uniform (scalar) flag
  float lim_lo[3], lim_hi[3], stiff[3], damp[3];
  v3 saxis[3];
  float sl_lo[3], sl_hi[3], sl_damp[3];
  int act_rot[3], act_sl[3];
  float gear_rot[3], gear_sl[3], alo_rot[3], ahi_rot[3], alo_sl[3], ahi_sl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lim_lo[k] = M->rot_lo[l][k]; lim_hi[k] = M->rot_hi[l][k];
    stiff[k] = M->rot_stiff[l][k]; damp[k] = M->rot_damp[l][k];
    saxis[k] = mk3(M->slide_axis[l][k][0], M->slide_axis[l][k][1], M->slide_axis[l][k][2]);
    sl_lo[k] = M->slide_lo[l][k]; sl_hi[k] = M->slide_hi[l][k]; sl_damp[k] = M->slide_damp[l][k];
    act_rot[k] = -1; act_sl[k] = -1;
    gear_rot[k] = gear_sl[k] = 0.0f; alo_rot[k] = alo_sl[k] = 0.0f; ahi_rot[k] = ahi_sl[k] = 0.0f;
  }
  for (int a = 0; a < Nu; ++a) {
    if (M->act_link[a] != l || !link_ok) continue;
    const int s = M->act_slot[a];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (s == k) { act_rot[k] = a; gear_rot[k] = M->act_gear[a]; alo_rot[k] = M->act_lo[a]; ahi_rot[k] = M->act_hi[a]; }
      if (s == 3 + k) { act_sl[k] = a; gear_sl[k] = M->act_gear[a]; alo_sl[k] = M->act_lo[a]; ahi_sl[k] = M->act_hi[a]; }
    }
  }
  // Slide slots a joint does not have are masked in the CONSTANTS, once: a zero axis makes the slot's velocity,
  // force, projection and limit terms exact zeros (everything it multiplies is finite).  Hinge slots are NOT
  // masked that way: the Euler angles of a slot the joint lacks are meaningless and can overflow near the gimbal
  // singularity, so they are discarded by selects (0 * inf would be NaN).
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool has_sl = is_joint && k < ns;
    saxis[k] = sel3(has_sl, saxis[k], mk3(0.0f, 0.0f, 0.0f));
    gear_sl[k] = has_sl ? gear_sl[k] : 0.0f;
  }
  v3 saxis_w[3];  // SLIDEW: the slide axes in the world frame, once
#pragma unroll
  for (int k = 0; k < 3; ++k) saxis_w[k] = rot(saxis[k], jc.ap_rot);
  auto slide_dir = [&](int k, q4 aprot) { return SLIDEW ? saxis_w[k] : rot(saxis[k], aprot); };
  int child_lane[MAXCH];
  {
    int nc = 0;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) child_lane[c] = -1;
    for (int c = l + 1; c < L; ++c)
      if (M->parent[c] == l && link_ok) {
#pragma unroll
        for (int j = 0; j < MAXCH; ++j)
          if (j == nc) child_lane[j] = lane_of(c);
        ++nc;
      }
  }
  int child_src[MAXCH];  // lane to pull child c's contribution from (a zero lane when there is none)
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) child_src[c] = child_lane[c] >= 0 ? child_lane[c] : (zero_lane >= 0 ? zero_lane : lane);
  // the generic kernels (MAXCH = 4) bound their child loops by the model's largest child count: a scalar
  // branch per slot; a skipped slot would have added exact zeros
  const int max_children = P.max_children;
  // MULTI: some joint of the model has more than one hinge dof.  A template parameter, not a wave-uniform flag:
  // around small blocks the compiler turns a uniform branch into selects and executes both sides.
  constexpr bool multi = MULTI;
  auto child_slot = [&](int c) { return MAXCH <= 3 || c < max_children; };
  // DPP layout: 0/1 masks — rm[s]: this link has an s-th child (it sits at lane - Ds); pm[s]: this link is the
  // s-th child of its parent (which sits at lane + Ds)
  float rm[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pm[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (DPP) {
    int myslot = -1;
    if (link_ok && parent >= 0) {
      myslot = 0;
      for (int c = 0; c < l; ++c) myslot += M->parent[c] == parent ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < (MAXCH < 4 ? MAXCH : 4); ++k) {
      rm[k] = child_lane[k] >= 0 ? 1.0f : 0.0f;
      pm[k] = myslot == k ? 1.0f : 0.0f;
    }
  }
  v3 col_pos[MAXCOL];
  float col_rad[MAXCOL];
  bool col_has[MAXCOL];
  {
    int nc = 0;
#pragma unroll
    for (int j = 0; j < MAXCOL; ++j) { col_has[j] = false; col_rad[j] = 0.0f; col_pos[j] = mk3(0, 0, 0); }
    for (int k = 0; k < M->n_col; ++k)
      if (M->col_link[k] == l && link_ok) {
#pragma unroll
        for (int j = 0; j < MAXCOL; ++j)
          if (j == nc) {
            col_has[j] = true; col_rad[j] = M->col_radius[k];
            col_pos[j] = mk3(M->col_pos[k][0], M->col_pos[k][1], M->col_pos[k][2]);
          }
        ++nc;
      }
  }
  int track_k = -1;
  for (int k = 0; k < K; ++k)
    if (M->track_link[k] == l && link_ok) track_k = k;
  const v3 com = mk3(M->com[l][0], M->com[l][1], M->com[l][2]);
  const float dt = M->dt, inv_dt = 1.0f / M->dt, vel_fac = M->vel_fac, ang_fac = M->ang_fac;
  const float two_inv_dt = 2.0f * inv_dt;
  const float js_pos = is_joint ? M->joint_scale_pos : 0.0f, js_ang = is_joint ? M->joint_scale_ang : 0.0f;
  const float coll_scale = M->collide_scale, invm_sum = ip.inv_mass + ic.inv_mass;
  // isotropic models: the shares of an angular correction, (-ib_p, ib_c)/(ib_p + ib_c) * joint_scale_ang
  // (zero on non-joint lanes through js_ang)
  f2 kang2 = mk2(0.0f, 0.0f);
  if constexpr (ISO) {
    const float ibs = ip.ib[0] + ic.ib[0];
    kang2 = mk2(-((ip.ib[0] / ibs) * js_ang), (ic.ib[0] / ibs) * js_ang);
  }
  const float mu = M->friction, elast = M->elasticity;
  const v3 grav = mk3(M->gravity[0], M->gravity[1], M->gravity[2]);
  const int rkind = M->reward_kind;
  const float rp0 = M->reward_params[0], rp1 = M->reward_params[1];
  const float dt_ctrl = M->dt * (float)nfr;

  // ---- state -------------------------------------------------------------------------------------------
  const float* s0 = P.state0 + l * MBD_LINK_STATE;
  v3 p = mk3(s0[0], s0[1], s0[2]);
  q4 r = q4{s0[3], s0[4], s0[5], s0[6]};
  v3 v = mk3(s0[7], s0[8], s0[9]);
  v3 w = mk3(s0[10], s0[11], s0[12]);
  if (!link_ok) { p = mk3(0, 0, 0); r = q4{1, 0, 0, 0}; v = mk3(0, 0, 0); w = mk3(0, 0, 0); }

  const float* u_row = P.us + (size_t)b * H * Nu;
  float u_rot[3], u_sl[3];     // actions of the current control step
  float un_rot[3], un_sl[3];   // actions of the next one, in flight while the substeps run
  auto load_actions = [&](int t, float (&ur)[3], float (&us)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ur[k] = u_row[(size_t)t * Nu + (act_rot[k] >= 0 ? act_rot[k] : 0)];
      if constexpr (SLIDES) us[k] = u_row[(size_t)t * Nu + (act_sl[k] >= 0 ? act_sl[k] : 0)];
      else us[k] = 0.0f;
    }
  };
  load_actions(0, u_rot, u_sl);
  // the parent's POSE for the next substep is fetched as soon as it is final (end of stage 4), so that its
  // LDS round trip overlaps the velocity stages; only the parent's velocities are fetched at the top
  v3 Pp_next = shfl3(p, plane);
  q4 Pr_next = shfl4(r, plane);
  float rew_sum = 0.0f;

  for (int t = 0; t < H; ++t) {
    // actuator.to_tau: clip to ctrlrange, times gear
    float tau[3], tau_sl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      tau[k] = fclip(act_rot[k] >= 0 ? u_rot[k] : 0.0f, alo_rot[k], ahi_rot[k]) * gear_rot[k];
      tau_sl[k] = SLIDES ? fclip(act_sl[k] >= 0 ? u_sl[k] : 0.0f, alo_sl[k], ahi_sl[k]) * gear_sl[k] : 0.0f;
    }
    float ctrl_cost = 0.0f;
    if ((rkind == MBD_REW_HALFCHEETAH || rkind == MBD_REW_ANT) && root_lane) {
      for (int a = 0; a < Nu; ++a) {
        float ua = u_row[(size_t)t * Nu + a];
        ctrl_cost = ctrl_cost + ua * ua;
      }
    }
    // issue the next control step's action loads now (clamped index: no branch) and keep them in flight
    // across the n_frames substeps; they are consumed at the top of the next iteration
    load_actions(t + 1 < H ? t + 1 : t, un_rot, un_sl);
    __builtin_amdgcn_sched_barrier(0);
    // link-frame origin before the step (rewards that look at the incoming state / finite differences)
    const v3 o0 = sub(p, rot(com, r));
    const v3 v0 = sub(v, cross(w, rot(com, r)));

    for (int fr = 0; fr < nfr; ++fr) {
      // ---- (1) joints.acceleration_update ----------------------------------------------------------
      v3 Pv = shfl3(v, plane), Pw = shfl3(w, plane);
      shfl_issue();  // in flight across the joint frames, joined before the anchor velocities
      v3 Pp = Pp_next;  // (no join here: the velocities are first needed after the joint frames)
      q4 Pr = Pr_next;
      // a link hanging off the world sees the static identity frame. Only models with a jointed root need
      // it (the generic kernels): a FREE root's joint is masked out, whatever parent data it computes with.
      if constexpr (SLIDES) {
        Pp = sel3(world_parent, mk3(0, 0, 0), Pp); Pv = sel3(world_parent, mk3(0, 0, 0), Pv);
        Pw = sel3(world_parent, mk3(0, 0, 0), Pw); Pr = sel4(world_parent, q4{1, 0, 0, 0}, Pr);
      }
      v3 fc_v, fc_w, fp_v, fp_w;
      {
        JointFrames f = joint_frames(jc, Pp, Pr, p, r, multi);
        const WInert2<ISO> W2 = world_inertia2<ISO, DIAG, AXI>(ip, ic, pack4(Pr, r));
        const v3x2 arm = f.arm;  // (rp, rc)
        shfl_join();
        const v3x2 va = add2(pack3(Pv, v), cross2(pack3(Pw, w), arm));  // anchor velocities (vp, vc)
        v3 rel_v = sub(hi3(va), lo3(va)), rel_w = sub(w, Pw);
        v3 T = mk3(0, 0, 0), F = mk3(0, 0, 0);
        auto torque = [&](int k, v3 ax, float ang) {
          float qdk = dot(rel_w, ax);
          float fk = ffma(-stiff[k], ang, ffma(-damp[k], qdk, tau[k]));
          fk = k < nr_eff ? fk : 0.0f;  // (a select: ang is garbage, possibly non-finite, for a slot the joint lacks)
          T = axpy(fk, ax, T);
        };
        torque(0, f.Xp, f.ang0);
        if (multi) {
          torque(1, f.ax1, f.ang1);
          torque(2, f.Zc, f.ang2);
        }
        if constexpr (SLIDES) {
#pragma unroll
          for (int k = 0; k < NS; ++k) {
            v3 s = slide_dir(k, f.aprot);
            float vs = dot(rel_v, s);
            float fk = ffma(-sl_damp[k], vs, tau_sl[k]);  // motor + MJCF joint damping of the slide dof
            F = axpy(fk, s, F);  // (s = 0 for a slot the joint lacks)
            rel_v = axpy(-vs, s, rel_v);
          }
        }
        T = axpy(-ang_damp, rel_w, T);
        F = axpy(-vel_damp, rel_v, F);
        // child: +F at its anchor, +T; parent: -F at its anchor, -T
        const v3x2 F2 = bcast3(F);
        const v3x2 lin = scale2(F2, mk2(-ip.inv_mass, ic.inv_mass));       // (fp_v, fc_v)
        const v3x2 tot = add2(bcast3(T), cross2(arm, F2));
        const v3x2 ang = scale2(iinv2<ISO, AXI>(ip, ic, W2, tot), mk2(-1.0f, 1.0f));  // (fp_w, fc_w)
        fc_v = hi3(lin); fp_v = lo3(lin);
        fc_w = hi3(ang); fp_w = lo3(ang);
      }
      // ---- (2) integrator.integrate_xdd -------------------------------------------------------------
      v3x2 acc = pack3(fc_v, fc_w);  // (linear, angular) acceleration, packed
      float damped_vx = 0.0f, damped_vy = 0.0f;  // vel_fac * v.{x,y}: computed inside the humanoids' exchange block
      bool have_damped = false;
      {
        v3 cv[MAXCH], cw[MAXCH];
        if constexpr (DPP) {  // ((own + child 0) + child 1) + child 2, as in the shuffle path
          v3 sv = fc_v, sw = fc_w;
          if constexpr (D0 == 1 && D1 == -4 && D2 == -6 && D3 == 0) {  // (the humanoids: one block, no s_nop)
            dpp_acc6x3_mul2(sv, sw, fp_v, fp_w, rm[0], rm[1], rm[2], vel_fac, v.x, v.y, damped_vx, damped_vy);
            have_damped = true;
          } else {
            dpp_acc6<-D0>(sv, sw, fp_v, fp_w, rm[0]);
            if constexpr (D1 != 0) dpp_acc6<-D1>(sv, sw, fp_v, fp_w, rm[1]);
            if constexpr (D2 != 0) dpp_acc6<-D2>(sv, sw, fp_v, fp_w, rm[2]);
            if constexpr (D3 != 0) dpp_acc6<-D3>(sv, sw, fp_v, fp_w, rm[3]);
          }
          acc = pack3(sv, sw);
        } else {
#pragma unroll
          for (int c = 0; c < MAXCH; ++c) {
            if (child_slot(c)) { cv[c] = shfl3(fp_v, child_src[c]); cw[c] = shfl3(fp_w, child_src[c]); }
          }
          shfl_join();
        }
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
          if (DPP || !child_slot(c)) continue;
          if (need_child_mask) {  // no zero lane in this model: mask missing children (scalar branch)
            cv[c] = sel3(child_lane[c] >= 0, cv[c], mk3(0, 0, 0));
            cw[c] = sel3(child_lane[c] >= 0, cw[c], mk3(0, 0, 0));
          }
          acc = add2(acc, pack3(cv[c], cw[c]));
        }
      }
      const v3 av = lo3(acc), aw = hi3(acc);
      if (!have_damped) { damped_vx = vel_fac * v.x; damped_vy = vel_fac * v.y; }
      v = mk3(ffma(av.x + grav.x, dt, damped_vx), ffma(av.y + grav.y, dt, damped_vy),
              ffma(av.z + grav.z, dt, vel_fac * v.z));
      w = mk3(ffma(aw.x, dt, ang_fac * w.x), ffma(aw.y, dt, ang_fac * w.y), ffma(aw.z, dt, ang_fac * w.z));
      const v3 p_prev = p;
      const q4 r_prev = r;
      p = mk3(ffma(v.x, dt, p.x), ffma(v.y, dt, p.y), ffma(v.z, dt, p.z));
      r = qrotvec(r, scale(w, dt));
      // ---- (3) joints.position_update (Jacobi) ------------------------------------------------------
      if constexpr (DPP) {
        dpp_fetch7<D0, D1, D2>(p, r, pm[0], pm[1], pm[2], Pp, Pr);
        if constexpr (D3 != 0) dpp_fmac7<D3>(p, r, pm[3], Pp, Pr);
      } else {
        Pp = shfl3(p, plane);
        Pr = shfl4(r, plane);
        shfl_join();
      }
      if constexpr (SLIDES) { Pp = sel3(world_parent, mk3(0, 0, 0), Pp); Pr = sel4(world_parent, q4{1, 0, 0, 0}, Pr); }
      v3 dc_p, dc_th, dp_p, dp_th;
      v3 cp[MAXCH], cth[MAXCH];
      {
        JointFrames f = joint_frames(jc, Pp, Pr, p, r, multi);
        v3 d = sub(f.ap, f.ac);
        if constexpr (SLIDES) {
#pragma unroll
          for (int k = 0; k < NS; ++k) {
            v3 s = slide_dir(k, f.aprot);
            float cf = -dot(d, s);
            d = axpy(cf, s, d);
          }
        }
        const WInert2<ISO> W2 = world_inertia2<ISO, DIAG, AXI>(ip, ic, pack4(Pr, r));
        const v3x2 arm = f.arm;  // (rp, rc)
        float c2 = dot(d, d);
        const v3x2 d2 = bcast3(d);
        const v3x2 cr = cross2(arm, d2);                       // (rp x d, rc x d)
        const f2 wq = dot2(cr, iinv2<ISO, AXI>(ip, ic, W2, cr));
        float den = ffma(invm_sum, c2, wq.x + wq.y) + 1e-20f;
        // angular alignment by joint type (0 hinges: weld; 1: Xc || Xp; 2: Yc _|_ Xp; 3: free)
        v3 A = sel3(nr == 1, f.Xc, f.Xp);
        v3 Bv = sel3(nr == 1, f.Xp, f.Yc);
        const float dxy = dot(f.Xp, f.Yc);
        float sc = nr_eff == 1 ? 1.0f : (nr_eff == 2 ? dxy : 0.0f);
        v3 e = scale(cross(A, Bv), sc);
        if (SLIDES && P.has_weld) {  // joints without a hinge dof keep the child's orientation locked to the parent's
          q4 qe = qmul(f.aprot, conj(f.acrot));
          float sg = __builtin_copysignf(2.0f, qe.w);
          e = sel3(nr_eff == 0, mk3(sg * qe.x, sg * qe.y, sg * qe.z), e);
        }
        // a - clamp(a, lo, hi): a-lo below, a-hi above, 0 inside; a select discards the slots the joint lacks
        auto viol_of = [&](int k, float a) { return k < nr_eff ? a - fclip(a, lim_lo[k], lim_hi[k]) : 0.0f; };
        // Angular corrections (alignment + up to three Euler-angle limits).  Anisotropic inertia: one
        // I^-1 e |e|^2 / (e.I_p^-1 e + e.I_c^-1 e) each (quotients (0,1) share a packed division, 2 goes alone).
        // Isotropic inertia ib*Id: that expression is e * ib_c/(ib_p + ib_c) — linear in e — so the errors are
        // summed first and applied once with the per-lane constants kang2 = (-k_p, k_c): no division at all.
        AngPrep ca, c0, c1, c2_;
        v3 E = e;
        auto limits_prepare = [&] {
          c0 = ang_prepare<ISO, AXI>(scale(f.Xp, -viol_of(0, f.ang0)), ip, ic, W2);
          if (multi) {
            c1 = ang_prepare<ISO, AXI>(scale(f.ax1, -viol_of(1, f.ang1)), ip, ic, W2);
            c2_ = ang_prepare<ISO, AXI>(scale(f.Zc, -viol_of(2, f.ang2)), ip, ic, W2);
          }
        };
        f2 q_ta, q01;  // (translation, alignment) and (limit 0, limit 1) quotients
        float q2;
        if constexpr (ISO) {
          E = axpy(-viol_of(0, f.ang0), f.Xp, E);
          if (multi) {
            E = axpy(-viol_of(1, f.ang1), f.ax1, E);
            E = axpy(-viol_of(2, f.ang2), f.Zc, E);
          }
          q_ta = mk2(div_pos_(c2, den), 0.0f);
        } else if constexpr (DPP) {
          // no exchange latency to hide here: all divisions run as interleaved independent chains (a dependent
          // packed FMA costs a wait state, which the compiler fills with s_nop when nothing else is at hand)
          ca = ang_prepare<ISO, AXI>(e, ip, ic, W2);
          limits_prepare();
          if (multi) {
            div2x2_(mk2(c2, ca.num), mk2(den, ca.den), mk2(c0.num, c1.num), mk2(c0.den, c1.den), q_ta, q01);
            q2 = div_pos_(c2_.num, c2_.den);
          } else {
            q_ta = div2_pos_(mk2(c2, ca.num), mk2(den, ca.den));
            q01 = mk2(div_pos_(c0.num, c0.den), 0.0f);
            q2 = 0.0f;
          }
        } else {
          ca = ang_prepare<ISO, AXI>(e, ip, ic, W2);
          q_ta = div2_pos_(mk2(c2, ca.num), mk2(den, ca.den));
        }
        float g = q_ta.x * js_pos;
        const v3x2 P2 = bcast3(scale(d, g));
        const v3x2 lin = scale2(P2, mk2(-ip.inv_mass, ic.inv_mass));  // (dp_p, dc_p)
        v3x2 dth2 = scale2(iinv2<ISO, AXI>(ip, ic, W2, cross2(arm, P2)), mk2(-1.0f, 1.0f));  // (dp_th, dc_th)
        v3x2 lin2 = lin;
        // slide limits: push the child back along the slide axis by the violation. Models without a limited
        // slide (planar roots of hopper / walker2d / halfcheetah) skip the block: it would add exact zeros.
        if (SLIDES && P.slide_limits) {
#pragma unroll
          for (int k = 0; k < NS; ++k) {
            v3 sx = slide_dir(k, f.aprot);
            float qs = dot(sub(f.ac, f.ap), sx);
            float viol = qs - fclip(qs, sl_lo[k], sl_hi[k]);
            v3 dl = scale(sx, -viol);
            float l2 = dot(dl, dl);
            const v3x2 dl2 = bcast3(dl);
            const v3x2 lcr = cross2(arm, dl2);
            const f2 lw = dot2(lcr, iinv2<ISO, AXI>(ip, ic, W2, lcr));
            float dens = ffma(invm_sum, l2, lw.x + lw.y);
            float gs = div_pos_(l2, dens + 1e-20f) * js_pos;
            const v3x2 Ps2 = bcast3(scale(dl, gs));
            lin2 = add2(lin2, scale2(Ps2, mk2(-ip.inv_mass, ic.inv_mass)));
            dth2 = add2(dth2, scale2(iinv2<ISO, AXI>(ip, ic, W2, cross2(arm, Ps2)), mk2(-1.0f, 1.0f)));
          }
        }
        // the translational corrections are final: the parent's share leaves now and its round trip hides
        // behind the angular limit corrections
        dc_p = hi3(lin2); dp_p = lo3(lin2);
        if constexpr (!DPP) {
#pragma unroll
          for (int c = 0; c < MAXCH; ++c) {
            if (child_slot(c)) cp[c] = shfl3(dp_p, child_src[c]);
          }
          shfl_issue();
        }
        if constexpr (ISO) {
          dth2 = axpy2(kang2, bcast3(E), dth2);
        } else {
          ang_apply(ca, q_ta.y, js_ang, dth2);
          if constexpr (!DPP) {  // (the shuffled kernels keep this work behind the translational exchange)
            limits_prepare();
            if (multi) {
              q01 = div2_pos_(mk2(c0.num, c1.num), mk2(c0.den, c1.den));
              q2 = div_pos_(c2_.num, c2_.den);
            } else {
              q01 = mk2(div_pos_(c0.num, c0.den), 0.0f);
              q2 = 0.0f;
            }
          }
          ang_apply(c0, q01.x, js_ang, dth2);
          if (multi) {
            ang_apply(c1, q01.y, js_ang, dth2);
            ang_apply(c2_, q2, js_ang, dth2);
          }
        }
        dc_th = hi3(dth2); dp_th = lo3(dth2);
      }
      {
        v3x2 acc = pack3(dc_p, dc_th);  // (translation, rotation vector), packed
        if constexpr (DPP) {
          v3 sp = dc_p, sth = dc_th;
          if constexpr (D0 == 1 && D1 == -4 && D2 == -6 && D3 == 0) {
            dpp_acc6x3(sp, sth, dp_p, dp_th, rm[0], rm[1], rm[2]);
          } else {
            dpp_acc6<-D0>(sp, sth, dp_p, dp_th, rm[0]);
            if constexpr (D1 != 0) dpp_acc6<-D1>(sp, sth, dp_p, dp_th, rm[1]);
            if constexpr (D2 != 0) dpp_acc6<-D2>(sp, sth, dp_p, dp_th, rm[2]);
            if constexpr (D3 != 0) dpp_acc6<-D3>(sp, sth, dp_p, dp_th, rm[3]);
          }
          acc = pack3(sp, sth);
        } else {
#pragma unroll
          for (int c = 0; c < MAXCH; ++c) {
            if (child_slot(c)) cth[c] = shfl3(dp_th, child_src[c]);
          }
          shfl_join();
        }
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
          if (DPP || !child_slot(c)) continue;
          if (need_child_mask) {
            cp[c] = sel3(child_lane[c] >= 0, cp[c], mk3(0, 0, 0));
            cth[c] = sel3(child_lane[c] >= 0, cth[c], mk3(0, 0, 0));
          }
          acc = add2(acc, pack3(cp[c], cth[c]));
        }
        p = add(p, lo3(acc));
        r = qrotvec_raw(r, hi3(acc));  // renormalised at the end of stage (4)
      }
      // ---- (4) sphere-plane contacts + collisions.resolve_position ---------------------------------
      v3 con_pos[MAXCOL];
      float con_dlam[MAXCOL];
      bool con_act[MAXCOL];
      {
        const WInert<ISO> Wc = world_inertia<ISO, DIAG, AXI>(ic, r);  // (r not yet renormalised, like the contact points)
        v3 cd_p = mk3(0, 0, 0), cd_th = mk3(0, 0, 0);
        if constexpr (MAXCOL == 2) {
          // both colliders of the link as one packed pair (the solve is Jacobi: each sees the pose of the stage's
          // start); their corrections are then added in collider order, exactly like the loop below
          const q4x2 R2 = bcast4(r);
          const v3x2 cpos = pack3(col_pos[0], col_pos[1]);
          const f2 rad = mk2(col_rad[0], col_rad[1]);
          const v3x2 off = rot2(cpos, R2);
          const v3x2 ctr = add2(bcast3(p), off);
          const f2 pen = rad - ctr.z;
          const bool act0 = col_has[0] && pen.x > 0.0f, act1 = col_has[1] && pen.y > 0.0f;
          const f2 h = fma2(mk2(-0.5f, -0.5f), pen, rad);
          const v3x2 pos = v3x2{ctr.x, ctr.y, ctr.z - h};
          const v3x2 rc = v3x2{off.x, off.y, off.z - h};
          const v3x2 cn = v3x2{rc.y, -rc.x, mk2(0.0f, 0.0f)};
          const v3x2 icn = iinv_z0_s2<ISO, AXI>(ic, Wc, cn);
          const f2 wn = mk2(ic.inv_mass, ic.inv_mass) + dot_az0_2(cn, icn);
          const v3x2 rl = add2(cpos, irot_z2(-h, R2));
          const v3x2 pprev = add2(bcast3(p_prev), rot2(rl, bcast4(r_prev)));
          v3x2 dx = sub2(pos, pprev);
          dx.z = mk2(0.0f, 0.0f);
          const f2 ct2 = fma2(dx.x, dx.x, dx.y * dx.y);
          const v3x2 cnt = cross_bz0_2(rc, dx);
          const v3x2 icnt = iinv_s2<ISO, AXI>(ic, Wc, cnt);
          const f2 dent = fma2(mk2(ic.inv_mass, ic.inv_mass), ct2, dot2(cnt, icnt));
          f2 q_n, q_g;  // (dlam / collide_scale, gt) of both colliders
          div2x2_(pen, wn, ct2, dent + mk2(1e-20f, 1e-20f), q_n, q_g);
          const f2 dlam = q_n * mk2(coll_scale, coll_scale);
          const f2 lim = mk2(mu, mu) * dlam;
          const f2 lhs = (ct2 * q_g) * q_g, rhs = lim * lim;
          const bool st0 = lhs.x < rhs.x, st1 = lhs.y < rhs.y;
          const f2 px = (-q_g) * dx.x, py = (-q_g) * dx.y;
          const v3x2 Pimp = v3x2{mk2(st0 ? px.x : 0.0f, st1 ? px.y : 0.0f), mk2(st0 ? py.x : 0.0f, st1 ? py.y : 0.0f), dlam};
          const v3x2 dth = iinv_s2<ISO, AXI>(ic, Wc, cross2(rc, Pimp));
          const v3 P0 = lo3(Pimp), P1 = hi3(Pimp), dth0 = lo3(dth), dth1 = hi3(dth);
          cd_p = sel3(act0, scale(P0, ic.inv_mass), cd_p);
          cd_th = sel3(act0, dth0, cd_th);
          cd_p = sel3(act1, axpy(ic.inv_mass, P1, cd_p), cd_p);
          cd_th = sel3(act1, add(cd_th, dth1), cd_th);
          con_pos[0] = lo3(pos); con_pos[1] = hi3(pos);
          con_dlam[0] = dlam.x; con_dlam[1] = dlam.y;
          con_act[0] = act0; con_act[1] = act1;
        } else {
#pragma unroll
        for (int j = 0; j < MAXCOL; ++j) {
          const v3 off = rot(col_pos[j], r);
          v3 ctr = add(p, off);
          float pen = col_rad[j] - ctr.z;
          bool active = col_has[j] && pen > 0.0f;
          // the contact point sits h below the sphere's centre: lever arm = rotated offset minus that drop,
          // link-frame coordinates = collider offset plus the drop rotated back
          const float h = ffma(-0.5f, pen, col_rad[j]);
          v3 pos = mk3(ctr.x, ctr.y, ctr.z - h);
          v3 rc = mk3(off.x, off.y, off.z - h);
          v3 cn = crossz(rc);
          v3 icn = iinv_z0<ISO, AXI>(ic, Wc, cn);
          float wn = ic.inv_mass + dot_az0(cn, icn);
          // (dlam and gt share one packed division below)
          v3 rl = add(col_pos[j], irot_z(-h, r));
          v3 pprev = add(p_prev, rot(rl, r_prev));
          v3 dx = sub(pos, pprev);
          dx.z = 0.0f;
          float ct2 = ffma(dx.x, dx.x, dx.y * dx.y);
          v3 cnt = cross_bz0(rc, dx);
          v3 icnt = iinv<ISO, AXI>(ic, Wc, cnt);
          float dent = ffma(ic.inv_mass, ct2, dot(cnt, icnt));
          const f2 q_ng = div2_pos_(mk2(pen, ct2), mk2(wn, dent + 1e-20f));
          float dlam = q_ng.x * coll_scale;
          float gt = q_ng.y;
          v3 Pimp = mk3(0.0f, 0.0f, dlam);
          float lim = mu * dlam;
          const bool stick = (ct2 * gt) * gt < lim * lim;
          Pimp.x = stick ? (-gt) * dx.x : 0.0f;
          Pimp.y = stick ? (-gt) * dx.y : 0.0f;
          const v3 dth = iinv<ISO, AXI>(ic, Wc, cross(rc, Pimp));
          // (the first collider adds to exact zeros: skipped)
          v3 ncd_p = j == 0 ? scale(Pimp, ic.inv_mass) : axpy(ic.inv_mass, Pimp, cd_p);
          v3 ncd_th = j == 0 ? dth : add(cd_th, dth);
          cd_p = sel3(active, ncd_p, cd_p);
          cd_th = sel3(active, ncd_th, cd_th);
          con_pos[j] = pos; con_dlam[j] = dlam; con_act[j] = active;
        }
        }
        p = add(p, cd_p);  // zero corrections on links without colliders
        r = qrotvec(r, cd_th);
        Pp_next = shfl3(p, plane);  // consumed by stage (1) of the next substep
        Pr_next = shfl4(r, plane);
        shfl_issue();  // the round trip hides behind stages (5) and (6)
      }
      // ---- (5) integrator.project_xd ------------------------------------------------------------------
      const v3 v_old = v, w_old = w;
      v = mk3((p.x - p_prev.x) * inv_dt, (p.y - p_prev.y) * inv_dt, (p.z - p_prev.z) * inv_dt);
      {
        q4 dq = qmul(r, conj(r_prev));
        float s = __builtin_copysignf(two_inv_dt, dq.w);
        w = mk3(dq.x * s, dq.y * s, dq.z * s);
      }
      // ---- (6) collisions.resolve_velocity --------------------------------------------------------------
      const WInert<ISO> Wc = world_inertia<ISO, DIAG, AXI>(ic, r);
#pragma unroll
      for (int j = 0; j < MAXCOL; ++j) {
        v3 rc = sub(con_pos[j], p);
        v3 vpt = add(v, cross(w, rc));
        // restitution needs the pre-solve normal velocity only when elasticity != 0 (wave-uniform); with
        // e = 0 the term min(-e*vn_prev, 0) is exactly 0
        float vn_prev = 0.0f;
        if (elast != 0.0f) vn_prev = add(v_old, cross(w_old, rc)).z;
        float vn = vpt.z;
        v3 vt = mk3(vpt.x, vpt.y, 0.0f);
        float vtn = sqrt_floor(ffma(vt.x, vt.x, vt.y * vt.y));
        float inv = div_(1.0f, vtn + 1e-10f);
        v3 dir = mk3(vt.x * inv, vt.y * inv, 0.0f);
        v3 cn = crossz(rc), cdv = cross_bz0(rc, dir);
        v3 icn = iinv_z0<ISO, AXI>(ic, Wc, cn), icd = iinv<ISO, AXI>(ic, Wc, cdv);
        float wn = ic.inv_mass + dot_az0(cn, icn), wt = ic.inv_mass + dot(cdv, icd);
        float rest = -elast * vn_prev;
        float dvn = fmin_(rest, 0.0f) - vn;
        float jt_max = (mu * con_dlam[j]) * inv_dt;
        float dvt = fmin_(jt_max * wt, vtn);
        const f2 q_nt = div2_sp_(mk2(dvn, dvt), mk2(wn, wt));
        float jn = q_nt.x, jt = -q_nt.y;
        v3 Pimp = mk3(dir.x * jt, dir.y * jt, jn);
        v3 nv = axpy(ic.inv_mass, Pimp, v);
        v3 nw = add(w, iinv<ISO, AXI>(ic, Wc, cross(rc, Pimp)));
        v = sel3(con_act[j], nv, v);
        w = sel3(con_act[j], nw, w);
      }
    }  // substeps

    // ---- reward (env wrapper's _get_reward) and tracked positions ------------------------------------
    const v3 o1 = sub(p, rot(com, r));
    float cart_cos = 0.0f, cart_vs = 0.0f;
    if (rkind == MBD_REW_CARTPOLE) {  // cartpole.py:45: cos(q[1]) - |qd[0]| (wave-uniform branch)
      v3 Pp = shfl3(p, plane);
      q4 Pr = shfl4(r, plane);
      if (world_parent) { Pp = mk3(0, 0, 0); Pr = q4{1, 0, 0, 0}; }
      JointFrames f = joint_frames(jc, Pp, Pr, p, r, multi);
      v3 sx = slide_dir(0, f.aprot);
      v3 vc = add(v, cross(w, sub(f.ac, p)));  // link 0 hangs off the static world
      float sn, cs;
      sincos_(f.ang0, &sn, &cs);
      cart_vs = dot(vc, sx);                    // own slide-0 velocity: used on lane 0
      cart_cos = shfl(cs, lane_of(1));          // cos of link 1's hinge angle, fetched by the root lane
    }
    if (root_lane) {
      float rew;
      if (rkind == MBD_REW_HUMANOIDRUN) {
        rew = o1.x * 1.0f - fclip(fabs_(o1.z - 1.3f), -1.0f, 1.0f) * 1.0f - fabs_(o1.y) * 0.1f;
      } else if (rkind == MBD_REW_HOPPER) {
        rew = o1.x - fclip(fabs_(o1.z - rp0), -1.0f, 1.0f) * rp1;
      } else if (rkind == MBD_REW_HALFCHEETAH) {
        rew = rp0 * ((o1.x - o0.x) / dt_ctrl) - rp1 * ctrl_cost;
      } else if (rkind == MBD_REW_ANT) {
        // reward_params[5] != 0: terminate_when_unhealthy (the stock setting) — the healthy term is unconditional
        float healthy = (M->reward_params[5] != 0.0f || (o1.z >= M->reward_params[2] && o1.z <= M->reward_params[3]))
                            ? M->reward_params[4] : 0.0f;
        rew = (rp0 * ((o1.x - o0.x) / dt_ctrl) + healthy) - rp1 * ctrl_cost;
      } else if (rkind == MBD_REW_CARTPOLE) {
        rew = cart_cos - fabs_(cart_vs);
      } else if (rkind == MBD_REW_HUMANOIDSTANDUP) {
        rew = 1.5f - fclip(fabs_(o1.z - 1.3f), -2.0f, 1.0f) - fabs_(o1.x) * 0.1f - fabs_(o1.y) * 0.1f;
      } else {
        rew = 1.0f + (-fabs_(v0.x - 1.6f) - fabs_(o0.z - 1.3f) - fabs_(o0.y) * 0.1f);
      }
      rew_sum = rew_sum + rew;
      if (b_ok && P.rewss) P.rewss[(size_t)b * H + t] = rew;
    }
    if (P.xpos && track_k >= 0 && b_ok) {
      float* o = P.xpos + (((size_t)b * H + t) * K + track_k) * 3;
      o[0] = o1.x; o[1] = o1.y; o[2] = o1.z;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { u_rot[k] = un_rot[k]; u_sl[k] = un_sl[k]; }
  }  // control steps
  if (P.dbg_clock && lane == 0) {
    P.dbg_clock[wave_id * 3 + 0] = dbg_t0;
    P.dbg_clock[wave_id * 3 + 1] = __builtin_amdgcn_s_memtime();
    P.dbg_clock[wave_id * 3 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |
                                      ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
  }
  if (root_lane && b_ok && P.rews) P.rews[b] = rew_sum / (float)H;
  if (P.state_final && link_ok && b_ok) {
    float* o = P.state_final + ((size_t)b * L + l) * MBD_LINK_STATE;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = r.w; o[4] = r.x; o[5] = r.y; o[6] = r.z;
    o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = w.x; o[11] = w.y; o[12] = w.z;
  }
}

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
  float d = fsqrt(dx * dx + dy * dy);
  d = fclip(d, 0.0f, 0.2f);
  float t = d / 0.2f;
  return 1.0f - t * t;
}

__global__ __launch_bounds__(64) void car2d_rollout_kernel(Car2dParams P) {
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
      collide = collide || (fsqrt(dx * dx + dy * dy) < 0.3f);
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
__global__ __launch_bounds__(256) void sample_kernel(uint32_t k0, uint32_t k1, int impl, int N, int HNu,
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

// The sampler in two halves, for plans that PREFETCH the noise of the next diffusion step behind the current rollout
// (mbd_plan_prefetch_noise): eps does not depend on the step's result, only the shift by Ybar does.
//   noise_kernel  eps[e] = normal(key)[e] for the flat elements [e_begin, e_begin + e_count) — the threefry counters,
//                 layouts and the ErfInv polynomial of sample_kernel
//   shift_kernel  Y0s[e] = clip(eps[e] * sigma + Ybar[e mod HNu], -1, 1) — the same two roundings as sample_kernel
__global__ __launch_bounds__(256) void noise_kernel(uint32_t k0, uint32_t k1, int impl, int N, int HNu,
                                                     unsigned long long e_begin, unsigned long long e_count,
                                                     float* __restrict__ eps) {
  const uint64_t size = (uint64_t)N * (uint64_t)HNu;
  const uint64_t half = (size + 1) / 2;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (impl == 1) {
    if (tid >= e_count) return;
    const uint64_t e = e_begin + tid;
    uint32_t o0, o1;
    threefry2x32(k0, k1, (uint32_t)(e >> 32), (uint32_t)e, o0, o1);
    eps[e] = bits_to_normal(o0 ^ o1);
    return;
  }
  if (e_begin == 0 && e_count == size) {  // whole tensor, legacy layout: one thread per threefry block
    if (tid >= half) return;
    const uint64_t j1 = tid + half;
    uint32_t o0, o1;
    threefry2x32(k0, k1, (uint32_t)tid, j1 < size ? (uint32_t)j1 : 0u, o0, o1);
    eps[tid] = bits_to_normal(o0);
    if (j1 < size) eps[j1] = bits_to_normal(o1);
    return;
  }
  if (tid >= e_count) return;
  const uint64_t e = e_begin + tid;
  const uint64_t j0 = e < half ? e : e - half, j1 = j0 + half;
  uint32_t o0, o1;
  threefry2x32(k0, k1, (uint32_t)j0, j1 < size ? (uint32_t)j1 : 0u, o0, o1);
  eps[e] = bits_to_normal(e < half ? o0 : o1);
}
__global__ __launch_bounds__(256) void shift_kernel(const float* __restrict__ eps, int HNu, unsigned long long e_begin,
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
// HumanoidTrack.eval_xref_logpd (humanoidtrack.py:98-106): xpos [B][H][K][3], xref [K][H][3]
__global__ __launch_bounds__(64) void logpd_track_kernel(const float* __restrict__ xpos,
                                                         const float* __restrict__ xref, int B, int H, int K,
                                                         float* __restrict__ lp) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float acc = 0.0f;
  for (int k = 0; k < K; ++k)
    for (int t = 0; t < H; ++t) {
      const float* a = xpos + (((size_t)b * H + t) * K + k) * 3;
      const float* c = xref + ((size_t)k * H + t) * 3;
      float ex = a[0] - c[0], ey = a[1] - c[1], ez = a[2] - c[2];
      float d = fsqrt(ex * ex + ey * ey + ez * ez);
      d = fclip(d, 0.0f, 0.5f);
      float s = d / 0.5f;
      acc = acc + s * s;
    }
  lp[b] = 0.0f - acc / (float)(H * K);
}
// Car2d.eval_xref_logpd (car2d.py:95-102): qs [B][H][3], xref [H][2]
__global__ __launch_bounds__(64) void logpd_car2d_kernel(const float* __restrict__ qs,
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

__global__ __launch_bounds__(kScoreThreads) void score_kernel(const float* __restrict__ rews,
                                                              const float* __restrict__ lp_demo, int N,
                                                              float rew_xref, float temp, int std_guard,
                                                              float* __restrict__ weights,
                                                              float* __restrict__ rew_mean_out,
                                                              float* __restrict__ lg_global) {
  // logp0 [N]: in LDS while it fits (every plan of the reference's sizes), in a plan-owned global scratch beyond
  // (each thread only ever touches its own entries, so the scratch needs no synchronisation either)
  extern __shared__ __attribute__((aligned(16))) float lg_lds[];
  float* __restrict__ lg = lg_global ? lg_global : lg_lds;
  __shared__ float red[kScoreThreads / 64];
  const int tid = threadIdx.x;
  float part = 0.0f;
  for (int i = tid; i < N; i += kScoreThreads) part = part + rews[i];
  const float rew_mean = block_sum(part, red) / (float)N;
  part = 0.0f;
  for (int i = tid; i < N; i += kScoreThreads) {
    float d = rews[i] - rew_mean;
    part = ffma(d, d, part);
  }
  float rew_std = fsqrt(block_sum(part, red) / (float)N);
  rew_std = (std_guard && rew_std < 1e-4f) ? 1.0f : rew_std;  // mbd_planner.py:112; path_integral.py:123 has none
  for (int i = tid; i < N; i += kScoreThreads) lg[i] = ((rews[i] - rew_mean) / rew_std) / temp;
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
  float mx = -__builtin_inff();
  for (int i = tid; i < N; i += kScoreThreads) mx = fmax_(mx, lg[i]);
  mx = block_max(mx, red);
  part = 0.0f;
  for (int i = tid; i < N; i += kScoreThreads) {
    float e = exp_(lg[i] - mx);
    lg[i] = e;
    part = part + e;
  }
  const float den = block_sum(part, red);
  for (int i = tid; i < N; i += kScoreThreads) weights[i] = lg[i] / den;
  if (tid == 0) *rew_mean_out = rew_mean;
}

// ---- A7-A8: weighted mean + score update (mbd_planner.py:128-133) ---------------------------------------
// A workgroup owns kWmE = 16 consecutive outputs e of [H][Nu] and splits the candidates into 64 groups:
// thread (g, j) runs a sequential fma over n = g, g+64, ... for output j (a wavefront reads four 64-byte row
// segments per load instruction); the 64 partials of an output are then added sequentially in g
// ("wsum64" of the contract).  ceil(HNu/16) workgroups of 1024 threads: every load of a thread is in flight at once.
constexpr int kWmE = 16, kWmG = 64;
__global__ __launch_bounds__(kWmE * kWmG) void wmean_kernel(const float* __restrict__ weights,
                                                            const float* __restrict__ Y0s, int N, int HNu,
                                                            const float* __restrict__ Ybar_i, float alpha_i,
                                                            float alpha_bar_i, float alpha_bar_im1, int literal,
                                                            float* __restrict__ Ybar_im1) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // the N weights: read once, shared by the 16 outputs
  __shared__ float red[kWmG][kWmE + 1];
  const int j = threadIdx.x & (kWmE - 1), g = threadIdx.x / kWmE;
  const int e_raw = blockIdx.x * kWmE + j;
  const int e = e_raw < HNu ? e_raw : HNu - 1;
  const float* __restrict__ col = Y0s + e;
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
    for (int k = 0; k < 32; ++k) acc = ffma(wl[n + k * kWmG], y[k], acc);
  }
  for (; n + 15 * kWmG < N; n += 16 * kWmG) {
    float y[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] = col[(size_t)(n + k * kWmG) * HNu];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = ffma(wl[n + k * kWmG], y[k], acc);
  }
  for (; n < N; n += kWmG) acc = ffma(wl[n], col[(size_t)n * HNu], acc);
  red[g][j] = acc;
  __syncthreads();
  if (g != 0 || e_raw >= HNu) return;
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

// The same weighted mean for LARGE N (multi-GPU plans average over all N_total candidates on every rank).  The tile
// kernel above reads 64-byte pieces of rows 3.4 KB apart (1.4 TB/s at N = 8192); here a workgroup owns ONE candidate
// group g and 256 consecutive outputs, so a wavefront reads 256 contiguous bytes of a row per load and the weight is a
// scalar.  The (group, output) partials go through a [64][HNu] scratch and wmean_finish_kernel adds the 64 partials
// of an output in group order and applies the update: the same chains and the same final order as wmean_kernel —
// the same bits.
constexpr int kWmT = 256;  // outputs per workgroup of the row-major variant
__global__ __launch_bounds__(kWmT) void wmean_partial_kernel(const float* __restrict__ weights,
                                                             const float* __restrict__ Y0s, int N, int HNu,
                                                             float* __restrict__ partial) {
  const int g = blockIdx.y;
  const int e_raw = blockIdx.x * kWmT + threadIdx.x;
  const int e = e_raw < HNu ? e_raw : HNu - 1;
  const float* __restrict__ col = Y0s + e;
  float acc = 0.0f;
  int n = g;
  for (; n + 31 * kWmG < N; n += 32 * kWmG) {
    float y[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) y[k] = col[(size_t)(n + k * kWmG) * HNu];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = ffma(weights[n + k * kWmG], y[k], acc);
  }
  for (; n < N; n += kWmG) acc = ffma(weights[n], col[(size_t)n * HNu], acc);
  if (e_raw < HNu) partial[(size_t)g * HNu + e_raw] = acc;
}
__global__ __launch_bounds__(64) void wmean_finish_kernel(const float* __restrict__ partial, int HNu,
                                                          const float* __restrict__ Ybar_i, float alpha_i,
                                                          float alpha_bar_i, float alpha_bar_im1, int literal,
                                                          float* __restrict__ Ybar_im1) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= HNu) return;
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
__global__ __launch_bounds__(64) void cma_spread_kernel(const float* __restrict__ weights,
                                                        const float* __restrict__ Y0s, int N, int HNu,
                                                        const float* __restrict__ mu_t, float* __restrict__ s_out) {
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
__global__ __launch_bounds__(64) void cma_sigma_kernel(const float* __restrict__ s, int HNu, float* __restrict__ sigma) {
  float part = 0.0f;
  for (int i = threadIdx.x; i < HNu; i += 64) part = part + s[i];
  float sig = (wave_sum(part) / (float)HNu) * (*sigma);
  if (threadIdx.x == 0) *sigma = sig > 1e-3f ? sig : 1e-3f;
}
// cem: indices of the K (<= 10) largest weights, ties towards the higher index (argsort()[::-1][:10])
__global__ __launch_bounds__(64) void cem_select_kernel(const float* __restrict__ weights, int N, int K,
                                                        int* __restrict__ idx_out, float* __restrict__ wl_global) {
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
__global__ __launch_bounds__(64) void cem_mean_kernel(const int* __restrict__ idx, int K, const float* __restrict__ Y0s,
                                                      int HNu, float* __restrict__ mu_out) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= HNu) return;
  float acc = 0.0f;
  for (int k = 0; k < K; ++k) acc = acc + Y0s[(size_t)idx[k] * HNu + e];
  mu_out[e] = acc / (float)K;
}

}  // namespace mbd
