// mbd_kernels.h — the hand-written rollout kernel (gfx950 / CDNA4, wave64) of the reverse-diffusion hot path,
// mbd/planners/mbd_planner.py:109 — jax.vmap(rollout_us) of the positional rigid-body step (utils.py:14-20; A2/A3 of
// SURVEY.md §8) — and what every rollout unit shares: RolloutParams, the DPP exchange, the noise generator that rides in
// spare workgroups, the per-lane constant records.  The other kernels of a step (sampling, score, weighted mean, sweeps'
// batched forms, path-integral updates, car2d) are in mbd_step_kernels.h; the two-candidates-per-lane and planar rollouts
// in mbd_pk2.h / mbd_planar.h.
//
// Layout of the rollout kernel (the one that matters): ONE LINK PER LANE.  A candidate occupies LPS
// consecutive lanes of a wavefront (LPS = 16 for the 11-link humanoid, 8 for the 7-link cheetah, 4 for
// the hopper), 64/LPS candidates per wavefront; a workgroup is four INDEPENDENT wavefronts (that is how the
// dispatcher puts one wavefront on each SIMD of a CU), so N=1024 humanoid candidates are 256 wavefronts on
// 64 CUs.  The 13-float link state, its previous pose and ~90 per-link model constants stay in VGPRs for
// the whole H x n_frames rollout; a parent's state and the children's constraint contributions move
// between lanes by DPP row shifts when the link tree fits one of the instantiated layouts (every built-in
// model), by ds_bpermute otherwise (no barriers); HBM is touched only for the action fetch (the step's normal and
// its Ybar_i entry, prefetched one control step ahead: candidates are formed lazily) and the reward store.  Workgroups
// beyond the rollout's generate the NEXT step's normals on CUs the rollout leaves idle.  What the issue model of a lone
// wavefront per SIMD taught: one 4-cycle slot per instruction, one per 32 bytes of code, 8-14 per
// taken branch — hence compile-time model switches, unrolled substep loops and predicated gathers in the prologue.
// All arithmetic follows mbd_math.h.
#pragma once

#include <type_traits>

#include "../../include/mbd_hip.h"
#include "mbd_math.h"

// The specification the TUNED instantiations compile in: a word of mbd_model_flags' switches, MBD_DEFAULT_SPEC
// (contact_avg) in the shipped library.  A model whose switches equal it runs the tuned kernels; any other word runs the general SPEC instantiations, which
// read the switches at run time — at a third to a half of the speed (profiles/r06_spec_cost.txt).  So the day a golden vector of
// Brax decides a switch the other way, the answer is a REBUILD with -DMBD_TUNED_SPEC=<word> (tools/build_variant.py; build()
// keeps lib/variants/libmbd_hip_sum.so = word 0, the summed contacts of rounds 1-5, held bit-exact to the checker by the GPU suite).
// Switches the tuned kernels can compile in: contact_avg, contact6_gauss_seidel, friction_vel_bound, restitution_min.
#ifndef MBD_TUNED_SPEC
#define MBD_TUNED_SPEC MBD_DEFAULT_SPEC
#endif
static_assert((MBD_TUNED_SPEC & ~(MBD_FLAG_CONTACT_AVG | MBD_FLAG_CONTACT6_GAUSS_SEIDEL | MBD_FLAG_FRICTION_VEL_BOUND | MBD_FLAG_RESTITUTION_MIN)) == 0,
              "MBD_TUNED_SPEC: euler_extrinsic and gyroscopic exist in the SPEC instantiations only");

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
  unsigned long long* dbg_clock;  // nullptr, or [waves][6] = (start tick, end tick, HW_ID|XCC<<32, prologue done, first
                                  // control step done, second control step done) (tools/probes/rollout_timeline.py)
  // LAZY candidates (plans): ybar != nullptr means `us` holds the step's NORMALS eps [B][H][Nu] and the action is
  // formed at the fetch, clip(eps * sigma + ybar[t][a], -1, 1) (mbd_planner.py:106, the sampler's two roundings) —
  // Y0s is never materialised; the weighted mean forms the same values from the same inputs
  const float* ybar;  // [H][Nu] = Ybar_i, or nullptr: `us` holds the actions themselves
  float sigma;        // sigma_i
  // The normals of the NEXT diffusion step (they depend on that step's key only), generated by the workgroups
  // blockIdx.x >= roll_blocks of THIS launch: a rollout of up to ~3000 humanoid candidates leaves most CUs without a
  // workgroup, so the sampler costs no launch, no event and no time on the step's critical path
  float* nz_out;      // [nz_N][nz_HNu] or nullptr
  uint32_t nz_k0, nz_k1;
  int nz_impl, nz_N, nz_HNu;
  int roll_blocks;    // workgroups of the rollout proper (the grid may be larger by the noise workgroups)
  // per-lane constants of the 3-D kernels (struct LaneRec3 [LPS], device): [0] lane = link, [1] the env's DPP layout,
  // [2] the DPP layout with helper lanes (HELP instantiations; nullptr when the model has no use for them)
  const void* lane_rec[3];
  // SWEEPS (mbd_sweep_*: several plans of one env in ONE launch): candidate b belongs to plan b / plan_N, which starts
  // from state0 + plan * plan_state_stride and (lazy) shifts by ybar + plan * plan_ybar_stride.  plan_N = 0: one plan.
  int plan_N, plan_state_stride, plan_ybar_stride;
  // PROGRESS word (pinned host memory, or nullptr): the launch stores progress_val into it as it STARTS — by stream
  // order everything enqueued before it on its stream has then finished.  The host side uses it to reuse a buffer of
  // normals without events on the step's stream (mbd_plan.hip: eps ring)
  int* progress;
  int progress_val;
  // XCD PINNING of small launches (round 5): workgroup i of a launch lands on XCD i mod 8, each XCD with its own L2.  A rollout
  // of at most 32 workgroups (one XCD's CUs) is launched 8 x as long: the multiples of 8 are its workgroups — all behind ONE L2,
  // which then fetches the kernel's code (30-40 KB) and the model once instead of once per XCD touched (hopper512: 0.47 MB of
  // its 1.26 MB per launch) — and the others carry the next step's normals, or leave at once.  0: not pinned; 1 + x: the
  // workgroups with residue x roll out (the host derives x from the process and the stream: fixed for a plan — its CUs keep
  // the code in their instruction caches from launch to launch — and different for concurrent small plans on other streams or
  // in other processes, which would otherwise all pile onto XCD 0).
  int xcd_pin;
  // CANDIDATES PER WAVEFRONT of an early-out launch (round 6; mbd_planar.h EO, rollout_kernel EO): a power of two below the
  // 64 / LPS the layout holds — a launch that would leave SIMDs idle anyway spreads its candidates over more wavefronts, and a
  // wavefront whose few candidates all have no sphere below the plane in a substep branches around the contact code.  0: the
  // launch fills its wavefronts (every instantiation without EO ignores the field).
  int cpw;
  // FUSED DEMO LOG-DENSITY (round 6; HumanoidTrack.eval_xref_logpd, humanoidtrack.py:98-106): lp != nullptr asks the
  // instantiations that compile the tracking reward in (RK = MBD_REW_HUMANOIDTRACK) to accumulate, on each tracked link's
  // lane and per control step, (clip(|x.pos - xref[k][t]|, 0, .5) / .5)^2 and to store lp[b] = 0 - (S_0 + ... + S_{K-1}) / (K H)
  // (S_k: link k's sum over t, in t order) — the values and the order of logpd_track_kernel, without the [B][H][K][3] round
  // trip through HBM and its launch.  xref: [K][H][3] (device).  The launcher tells its caller whether it happened.
  float* lp;
  const float* xref;
};
__device__ __forceinline__ void rollout_progress(const RolloutParams& P) {
  if (P.progress && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(P.progress, P.progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the plan a candidate belongs to (sweeps) — 0 for every ordinary launch
__device__ __forceinline__ int plan_of(const RolloutParams& P, int b) { return P.plan_N > 0 ? b / P.plan_N : 0; }

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
// The four child slots of the ant layout (shifts -1, +2, +4, +6) in ONE block: one hazard s_nop for 24 accumulations
__device__ __forceinline__ void dpp_acc6x4_ant(v3& a, v3& b, v3 x, v3 y, float m0, float m1, float m2, float m3) {
  asm("s_nop 1\n\t" MBD_F6("row_shr:1", 12) MBD_F6("row_shl:2", 13) MBD_F6("row_shl:4", 14) MBD_F6("row_shl:6", 15)
      : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z)
      : "v"(x.x), "v"(x.y), "v"(x.z), "v"(y.x), "v"(y.y), "v"(y.z), "v"(m0), "v"(m1), "v"(m2), "v"(m3));
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
// ant (D = (+1, -2, -4, -6)): the four slots in one block
__device__ __forceinline__ void dpp_fetch7x4_ant(v3 p, q4 r, float m0, float m1, float m2, float m3, v3& Pp, q4& Pr) {
  float o0 = dpp_from<1>(p.x) * m0, o1 = dpp_from<1>(p.y) * m0, o2 = dpp_from<1>(p.z) * m0;
  float o3 = dpp_from<1>(r.w) * m0, o4 = dpp_from<1>(r.x) * m0, o5 = dpp_from<1>(r.y) * m0;
  float o6 = dpp_from<1>(r.z) * m0;
  asm("s_nop 1\n\t" MBD_DPP_F(0, 7, 14, "row_shr:2") MBD_DPP_F(1, 8, 14, "row_shr:2") MBD_DPP_F(2, 9, 14, "row_shr:2")
          MBD_DPP_F(3, 10, 14, "row_shr:2") MBD_DPP_F(4, 11, 14, "row_shr:2") MBD_DPP_F(5, 12, 14, "row_shr:2")
              MBD_DPP_F(6, 13, 14, "row_shr:2") MBD_DPP_F(0, 7, 15, "row_shr:4") MBD_DPP_F(1, 8, 15, "row_shr:4")
                  MBD_DPP_F(2, 9, 15, "row_shr:4") MBD_DPP_F(3, 10, 15, "row_shr:4") MBD_DPP_F(4, 11, 15, "row_shr:4")
                      MBD_DPP_F(5, 12, 15, "row_shr:4") MBD_DPP_F(6, 13, 15, "row_shr:4") MBD_DPP_F(0, 7, 16, "row_shr:6")
                          MBD_DPP_F(1, 8, 16, "row_shr:6") MBD_DPP_F(2, 9, 16, "row_shr:6") MBD_DPP_F(3, 10, 16, "row_shr:6")
                              MBD_DPP_F(4, 11, 16, "row_shr:6") MBD_DPP_F(5, 12, 16, "row_shr:6")
                                  MBD_DPP_F(6, 13, 16, "row_shr:6")
      : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+v"(o6)
      : "v"(p.x), "v"(p.y), "v"(p.z), "v"(r.w), "v"(r.x), "v"(r.y), "v"(r.z), "v"(m1), "v"(m2), "v"(m3));
  Pp = v3{o0, o1, o2};
  Pr = q4{o3, o4, o5, o6};
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

// jax.random.normal(key, (N, HNu)) in flat element order, by the threads tid0, tid0 + stride, ...: legacy layout —
// one threefry block per thread-item, pairing element j with j + half (both outputs used); partitionable layout —
// one block per element.  Used by noise_kernel and by the noise workgroups of a rollout launch.
__device__ __forceinline__ void noise_fill(uint32_t k0, uint32_t k1, int impl, uint64_t size, uint64_t tid0,
                                           uint64_t stride, float* __restrict__ eps) {
  if (impl == 1) {
    for (uint64_t e = tid0; e < size; e += stride) {
      uint32_t o0, o1;
      threefry2x32(k0, k1, (uint32_t)(e >> 32), (uint32_t)e, o0, o1);
      eps[e] = bits_to_normal(o0 ^ o1);
    }
    return;
  }
  const uint64_t half = (size + 1) / 2;
  for (uint64_t j0 = tid0; j0 < half; j0 += stride) {
    const uint64_t j1 = j0 + half;
    uint32_t o0, o1;
    threefry2x32(k0, k1, (uint32_t)j0, j1 < size ? (uint32_t)j1 : 0u, o0, o1);
    eps[j0] = bits_to_normal(o0);
    if (j1 < size) eps[j1] = bits_to_normal(o1);
  }
}
// the noise workgroups of a rollout launch (blockIdx.x >= P.roll_blocks)
__device__ __forceinline__ void noise_blocks(const RolloutParams& P) {
  const uint64_t nb = gridDim.x - P.roll_blocks;
  noise_fill(P.nz_k0, P.nz_k1, P.nz_impl, (uint64_t)P.nz_N * (uint64_t)P.nz_HNu,
             (uint64_t)(blockIdx.x - P.roll_blocks) * blockDim.x + threadIdx.x, nb * blockDim.x, P.nz_out);
}
// Which workgroup of the rollout this one is — or -1: it was a noise workgroup (done) or has nothing to do.  Wave-uniform.
__device__ __forceinline__ int rollout_block(const RolloutParams& P) {
  if (P.xcd_pin) {  // (the grid is 8 x roll_blocks: RolloutParams::xcd_pin)
    const int q = blockIdx.x >> 3, r = ((int)(blockIdx.x & 7) - (P.xcd_pin - 1)) & 7;  // r: residue relative to the pinned XCD
    if (r == 0) return q;
    if (P.nz_out)
      noise_fill(P.nz_k0, P.nz_k1, P.nz_impl, (uint64_t)P.nz_N * (uint64_t)P.nz_HNu,
                 (uint64_t)(q * 7 + r - 1) * blockDim.x + threadIdx.x, (uint64_t)(7 * P.roll_blocks) * blockDim.x, P.nz_out);
    return -1;
  }
  if ((int)blockIdx.x >= P.roll_blocks) {  // the next step's normals, on CUs the rollout leaves idle
    noise_blocks(P);
    return -1;
  }
  return (int)blockIdx.x;
}

// ---- code placement (tools/tune_phase.py) -----------------------------------------------------------------------------
#if defined(MBD_PHASE_TUNING) || !__has_include("mbd_phase_gen.inc")
constexpr int mbd_pad_3d(int, int) { return 0; }
constexpr int mbd_pad_planar(int, int, int, int, int, int, int) { return 0; }
#else
#include "mbd_phase_gen.inc"
#endif
template <int N, class F>
__device__ __forceinline__ void repeat_n(F& f) {  // f() N times, straight-line
  if constexpr (N > 0) {
    f();
    repeat_n<N - 1>(f);
  }
}
template <int N>
__device__ __forceinline__ void phase_pad() {  // N x s_nop 0: shifts the code behind it by 4 N bytes
  if constexpr (N >= 1) {
    asm volatile("s_nop 0");
    phase_pad<N - 1>();
  }
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
  float inv = rcp_exact(cb + 1e-10f);  // (1e-10 <= cb + 1e-10 <= 1 + 1e-10)
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

// ---- per-lane constants of rollout_kernel, gathered ONCE per env (mbd_env_create) ---------------------------------------
// What a lane of a candidate needs to know about its link: which link, where its parent and children sit, its joint,
// inertia, actuators, colliders.  Gathering it is ~50 dependent reads of the model (lane table -> link -> parent, loops
// over actuators / children / colliders).  Done by the kernel itself it cost a launch ~16 us as global round trips, ~8 us
// through a per-wavefront LDS copy of the model, ~3 us as this record (tools/gpu_hscale.py, gpu_ab2.sh: -5 us against
// the LDS copy on the metric's rollout, 1 %).  lane_setup3_kernel does it once per env and layout into LaneRec3 records (one per lane of a
// candidate group, relative lanes); rollout_kernel loads its record in one batch of independent reads.  (The planar
// kernels keep their own gathers: the same record there measured no gain on hopper and -1 % on halfcheetah.)  Array sizes are
// the maxima of every instantiation (4 child slots, 5 colliders, 6 inertia entries, 3 slide slots).
struct LaneRec3 {
  int l, link_ok, root_lane, parent, nr, is_joint, ns, world_parent, nr_eff;
  int plane_rel;          // lane (within the candidate's group) holding the parent; the own lane for a world parent
  int need_child_mask, lane_of1_rel;
  int child_lane_rel[4];  // -1: no such child
  int child_src_rel[4];   // lane to pull child c's contribution from (a zero lane / the own lane when there is none)
  int act_rot[3], act_sl[3], track_k;
  int col_has[5];
  float ic_inv_mass, ip_inv_mass, ic_ib[6], ip_ib[6];  // (axisymmetric models: already remapped, axi_remap)
  float ap_pos[3], ac_pos[3], ap_rot[4], ac_rot[4];
  float ang_damp, vel_damp;
  float lim_lo[3], lim_hi[3], stiff[3], damp[3];
  float saxis[3][3], saxis_w[3][3], sl_lo[3], sl_hi[3], sl_damp[3];
  float gear_rot[3], gear_sl[3], alo_rot[3], ahi_rot[3], alo_sl[3], ahi_sl[3];
  float rm[4], pm[4];
  float col_pos[5][3], col_rad[5];
  float com[3];
  float js_pos, js_ang, invm_sum, kang2[2];
  // HELPER layout (lane_setup3_kernel, helpers != 0): a link with more than two sphere colliders (humanoidstandup's
  // torso: five) keeps colliders 0, 1 on its own lane and lends 2, 3 and 4 to two of the candidate's idle lanes, which
  // carry them in their slots 0, 1 together with the owner's inertia.  help_src_rel: the lane whose pose a lane's
  // colliders ride on (the owner for a helper, the lane itself otherwise); help_from_rel[2]: the helper lanes of an
  // owner (the lane itself otherwise); is_owner / is_helper.
  int help_src_rel, help_from_rel[2], is_owner, is_helper;
};
#ifndef MBD_SHARED_ONLY  // (translation units that only want the shared pieces: mbd_pk2.hip)
// lps lanes; dpp: the lane <-> link table of a DPP layout is in force (else lane = link); axi: axisymmetric inertia
// instantiation (the tensors are remapped to (a, a, c, u)).  The code of the former kernel prologue, unchanged.
static __global__ void lane_setup3_kernel(const mbd_model_t* __restrict__ M, const signed char* __restrict__ lane_tab, int LPS,
                                   int dpp, int axi, LaneRec3* __restrict__ out, int helpers = 0) {
  const int l_lane = threadIdx.x;
  if (l_lane >= LPS) return;
  const bool DPP = dpp != 0;
  LaneRec3 R;
  const int L = M->n_links, Nu = M->n_act, K = M->n_track;
  const int l_link = DPP ? (int)lane_tab[l_lane] : l_lane;  // the link this lane holds
  const bool link_ok = l_link >= 0 && l_link < L;
  const int l = link_ok ? l_link : 0;
  auto lane_of = [&](int link) { return DPP ? (int)lane_tab[16 + link] : link; };  // (relative to the group's first lane)
  const int lane = l_lane;
  const int parent = M->parent[l];
  const int nr = M->n_rot[l];
  const bool is_joint = link_ok && nr >= 0;
  const int ns = M->n_slide[l];  // (kernels without slide joints ignore everything slide-related)
  const bool world_parent = parent < 0;
  R.l = l; R.link_ok = link_ok; R.root_lane = link_ok && l == 0; R.parent = parent; R.nr = nr; R.is_joint = is_joint;
  R.ns = ns; R.world_parent = world_parent; R.nr_eff = is_joint ? nr : -1;
  R.plane_rel = parent >= 0 ? lane_of(parent) : lane;
  Inert<false> ic, ip;
  ic.inv_mass = M->inv_mass[l];
  ip.inv_mass = world_parent ? 0.0f : M->inv_mass[parent >= 0 ? parent : 0];
  for (int k = 0; k < 6; ++k) {
    ic.ib[k] = M->inv_inertia[l][k];
    ip.ib[k] = world_parent ? 0.0f : M->inv_inertia[parent >= 0 ? parent : 0][k];
  }
  if (axi) { axi_remap<false>(ic); axi_remap<false>(ip); }
  R.ic_inv_mass = ic.inv_mass; R.ip_inv_mass = ip.inv_mass;
  for (int k = 0; k < 6; ++k) { R.ic_ib[k] = ic.ib[k]; R.ip_ib[k] = ip.ib[k]; }
  for (int k = 0; k < 3; ++k) { R.ap_pos[k] = M->ap_pos[l][k]; R.ac_pos[k] = M->ac_pos[l][k]; }
  for (int k = 0; k < 4; ++k) { R.ap_rot[k] = M->ap_rot[l][k]; R.ac_rot[k] = M->ac_rot[l][k]; }
  const q4 ap_rot = q4{M->ap_rot[l][0], M->ap_rot[l][1], M->ap_rot[l][2], M->ap_rot[l][3]};
  // non-joint lanes (free root, padding) are masked through their scalars: every contribution they
  // compute is then an exact zero
  R.ang_damp = is_joint ? M->ang_damp[l] : 0.0f; R.vel_damp = is_joint ? M->vel_damp[l] : 0.0f;
  const int zero_lane = M->n_rot[0] < 0 ? lane_of(0) : (L < LPS ? L : -1);  // a lane contributing zeros
  R.need_child_mask = !(M->n_rot[0] < 0) && !(L < LPS);
  R.lane_of1_rel = lane_of(1);
  v3 saxis[3];
  int act_rot[3], act_sl[3];
  float gear_rot[3], gear_sl[3], alo_rot[3], ahi_rot[3], alo_sl[3], ahi_sl[3];
  for (int k = 0; k < 3; ++k) {
    R.lim_lo[k] = M->rot_lo[l][k]; R.lim_hi[k] = M->rot_hi[l][k];
    R.stiff[k] = M->rot_stiff[l][k]; R.damp[k] = M->rot_damp[l][k];
    saxis[k] = mk3(M->slide_axis[l][k][0], M->slide_axis[l][k][1], M->slide_axis[l][k][2]);
    R.sl_lo[k] = M->slide_lo[l][k]; R.sl_hi[k] = M->slide_hi[l][k]; R.sl_damp[k] = M->slide_damp[l][k];
    act_rot[k] = -1; act_sl[k] = -1;
    gear_rot[k] = gear_sl[k] = 0.0f; alo_rot[k] = alo_sl[k] = 0.0f; ahi_rot[k] = ahi_sl[k] = 0.0f;
  }
  for (int a = 0; a < Nu; ++a) {
    if (M->act_link[a] != l || !link_ok) continue;
    const int s = M->act_slot[a];
    for (int k = 0; k < 3; ++k) {
      if (s == k) { act_rot[k] = a; gear_rot[k] = M->act_gear[a]; alo_rot[k] = M->act_lo[a]; ahi_rot[k] = M->act_hi[a]; }
      if (s == 3 + k) { act_sl[k] = a; gear_sl[k] = M->act_gear[a]; alo_sl[k] = M->act_lo[a]; ahi_sl[k] = M->act_hi[a]; }
    }
  }
  // Slide slots a joint does not have are masked in the CONSTANTS, once: a zero axis makes the slot's velocity,
  // force, projection and limit terms exact zeros (everything it multiplies is finite).  Hinge slots are NOT
  // masked that way: the Euler angles of a slot the joint lacks are meaningless and can overflow near the gimbal
  // singularity, so they are discarded by selects (0 * inf would be NaN).
  for (int k = 0; k < 3; ++k) {
    const bool has_sl = is_joint && k < ns;
    saxis[k] = sel3(has_sl, saxis[k], mk3(0.0f, 0.0f, 0.0f));
    gear_sl[k] = has_sl ? gear_sl[k] : 0.0f;
    const v3 sw = rot(saxis[k], ap_rot);  // SLIDEW: the slide axes in the world frame, once
    R.saxis[k][0] = saxis[k].x; R.saxis[k][1] = saxis[k].y; R.saxis[k][2] = saxis[k].z;
    R.saxis_w[k][0] = sw.x; R.saxis_w[k][1] = sw.y; R.saxis_w[k][2] = sw.z;
    R.act_rot[k] = act_rot[k]; R.act_sl[k] = act_sl[k];
    R.gear_rot[k] = gear_rot[k]; R.gear_sl[k] = gear_sl[k];
    R.alo_rot[k] = alo_rot[k]; R.ahi_rot[k] = ahi_rot[k]; R.alo_sl[k] = alo_sl[k]; R.ahi_sl[k] = ahi_sl[k];
  }
  int child_lane[4] = {-1, -1, -1, -1};
  {
    int nc = 0;
    for (int c = l + 1; c < L; ++c)
      if (M->parent[c] == l && link_ok) {
        if (nc < 4) child_lane[nc] = lane_of(c);
        ++nc;
      }
  }
  for (int c = 0; c < 4; ++c) {
    R.child_lane_rel[c] = child_lane[c];
    R.child_src_rel[c] = child_lane[c] >= 0 ? child_lane[c] : (zero_lane >= 0 ? zero_lane : lane);
  }
  // DPP layout: 0/1 masks — rm[s]: this link has an s-th child (it sits at lane - Ds); pm[s]: this link is the
  // s-th child of its parent (which sits at lane + Ds)
  int myslot = -1;
  if (DPP && link_ok && parent >= 0) {
    myslot = 0;
    for (int c = 0; c < l; ++c) myslot += M->parent[c] == parent ? 1 : 0;
  }
  for (int k = 0; k < 4; ++k) {
    R.rm[k] = (DPP && child_lane[k] >= 0) ? 1.0f : 0.0f;
    R.pm[k] = (DPP && myslot == k) ? 1.0f : 0.0f;
  }
  {
    int nc = 0;
    for (int j = 0; j < 5; ++j) { R.col_has[j] = 0; R.col_rad[j] = 0.0f; R.col_pos[j][0] = R.col_pos[j][1] = R.col_pos[j][2] = 0.0f; }
    for (int k = 0; k < M->n_col; ++k)
      if (M->col_link[k] == l && link_ok) {
        if (nc < 5) {
          R.col_has[nc] = 1; R.col_rad[nc] = M->col_radius[k];
          R.col_pos[nc][0] = M->col_pos[k][0]; R.col_pos[nc][1] = M->col_pos[k][1]; R.col_pos[nc][2] = M->col_pos[k][2];
        }
        ++nc;
      }
  }
  R.help_src_rel = lane; R.help_from_rel[0] = R.help_from_rel[1] = lane; R.is_owner = 0; R.is_helper = 0;
  if (helpers && DPP) {
    // the one link with more than two colliders, its first two idle lanes (the host checked that all of this exists)
    int owner = -1;
    for (int c = 0; c < L && owner < 0; ++c) {
      int n = 0;
      for (int k = 0; k < M->n_col; ++k) n += M->col_link[k] == c ? 1 : 0;
      if (n > 2) owner = c;
    }
    int idle[2] = {-1, -1}, ni = 0;
    for (int q = 0; q < LPS && ni < 2; ++q)
      if ((int)lane_tab[q] < 0) idle[ni++] = q;
    if (owner >= 0 && ni == 2) {
      const int owner_lane = lane_of(owner);
      const int which = lane == idle[0] ? 0 : (lane == idle[1] ? 1 : -1);
      if (link_ok && l == owner) {
        R.is_owner = 1; R.help_from_rel[0] = idle[0]; R.help_from_rel[1] = idle[1];
        for (int j = 2; j < 5; ++j) R.col_has[j] = 0;  // (stage (4) of the owner runs slots 0, 1; the rest come from the helpers)
      } else if (which >= 0) {
        R.is_helper = 1; R.help_src_rel = owner_lane;
        int nc = 0;
        for (int k = 0; k < M->n_col; ++k)
          if (M->col_link[k] == owner) {
            const int slot = nc - 2 - 2 * which;  // helper 0: colliders 2, 3; helper 1: collider 4 (and 5)
            if (slot >= 0 && slot < 2) {
              R.col_has[slot] = 1; R.col_rad[slot] = M->col_radius[k];
              R.col_pos[slot][0] = M->col_pos[k][0]; R.col_pos[slot][1] = M->col_pos[k][1]; R.col_pos[slot][2] = M->col_pos[k][2];
            }
            ++nc;
          }
        R.ic_inv_mass = M->inv_mass[owner];
        for (int k = 0; k < 6; ++k) R.ic_ib[k] = M->inv_inertia[owner][k];
      }
    }
  }
  int track_k = -1;
  for (int k = 0; k < K; ++k)
    if (M->track_link[k] == l && link_ok) track_k = k;
  R.track_k = track_k;
  for (int k = 0; k < 3; ++k) R.com[k] = M->com[l][k];
  R.js_pos = is_joint ? M->joint_scale_pos : 0.0f; R.js_ang = is_joint ? M->joint_scale_ang : 0.0f;
  R.invm_sum = ip.inv_mass + ic.inv_mass;
  // isotropic models: the shares of an angular correction, (-ib_p, ib_c)/(ib_p + ib_c) * joint_scale_ang
  // (zero on non-joint lanes through js_ang)
  {
    const float ibs = ip.ib[0] + ic.ib[0];
    R.kang2[0] = -((ip.ib[0] / ibs) * R.js_ang); R.kang2[1] = (ic.ib[0] / ibs) * R.js_ang;
  }
  out[l_lane] = R;
}

#endif  // MBD_SHARED_ONLY

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
// RK    the model's reward kind as a COMPILE-TIME constant (-1: read at run time).  Outside its substeps a control step
//       is ~200 instructions and, with the reward kind a run-time value, a chain of eight taken branches (each 30-60
//       cycles for a lone wavefront) plus work only some rewards need (the incoming link-frame origin and velocity):
//       0.48 us per control step on the humanoid, 4 % of its rollout (kernel time against n_frames, tools/gpu_nfr.py)
// SKIP6 stage (6) of the collider slots j >= 1 sits behind a wave-uniform test "some lane has an active contact in slot j"
//       (the slot's whole effect is two selects on its `active` flag, so skipping it is exact): for models whose second
//       collider rarely touches — ant: the ankle end of a lower leg's capsule, 88 of 880 instructions per substep.  The
//       block is placed out of line (a lone wavefront falls through when the slot is idle, and pays two taken branches
//       when it is not): worth it only where idle is the rule
//       (the same for the FIRST collider of the humanoids — feet that are off the floor in a tenth of a four-candidate
//       wavefront's substeps — with the active side falling through: tried, the two tests cut the scheduler's regions,
//       846 -> 869 instructions on the active side, 548 -> 553 us: not built)
// NFR   the model's n_frames as a compile-time constant (0: read at run time): the substep loop then runs exactly twice
//       over NFR / 2 substeps in line, plus one in line for an odd count, and leaves by falling through — two taken
//       branches per control step instead of four (n_frames = 7)
// SPEC  the model's specification switches (mbd_model_flags: contact_avg, contact6_gauss_seidel, friction_vel_bound,
//       restitution_min, euler_extrinsic, gyroscopic — DESIGN.md §9) are read at run time and honoured, each exactly as
//       the checker's 3-D restatement states it.  Only the general shuffle-exchange instantiations are built with it (models
//       with any such bit run there: launch_rollout); every other instantiation compiles the default specification in.
// ---- helpers of the SPEC instantiations ---------------------------------------------------------------------------------
// MBD_FLAG_EULER_EXTRINSIC: the joint-frame Euler angles / gimbal axes of R_rel = Rz(c) Ry(b) Rx(a) — the default
// decomposition with the roles of the parent's and the child's joint frame exchanged, angles negated
__device__ __forceinline__ void euler_extrinsic(q4 aprot, q4 acrot, float& a0, float& a1, float& a2, v3& x0, v3& x1, v3& x2) {
  const axes3 A = qaxes(aprot), C = qaxes(acrot);
  const float sb = fclip(dot(A.Z, C.X), -1.0f, 1.0f);
  const float cb = sqrt_floor(ffma(-sb, sb, 1.0f));
  const float inv = rcp_exact(cb + 1e-10f);
  a0 = -angle_unit(-dot(A.Z, C.Y) * inv, dot(A.Z, C.Z) * inv);
  a1 = -angle_unit(sb, cb);
  a2 = -angle_unit(-dot(A.Y, C.X) * inv, dot(A.X, C.X) * inv);
  x0 = C.X; x2 = A.Z;
  x1 = scale(cross(A.Z, C.X), inv);
}
// MBD_FLAG_GYROSCOPIC: body-frame inertia from its inverse (adjugate / determinant), and -I^-1 (w x I w) in the body frame
struct Inertia6 { float v[6]; };
__device__ __forceinline__ Inertia6 inertia_from_inverse(const float b[6]) {
  const float xx = b[0], yy = b[1], zz = b[2], xy = b[3], xz = b[4], yz = b[5];
  const float c0 = yy * zz - yz * yz, c1 = xz * yz - xy * zz, c2 = xy * yz - xz * yy;
  const float det = (xx * c0 + xy * c1) + xz * c2;
  const float id = 1.0f / det;
  Inertia6 I;
  I.v[0] = c0 * id; I.v[1] = (xx * zz - xz * xz) * id; I.v[2] = (xx * yy - xy * xy) * id;
  I.v[3] = c1 * id; I.v[4] = c2 * id; I.v[5] = (xy * xz - xx * yz) * id;
  return I;
}
__device__ __forceinline__ v3 gyro_accel(const float ib[6], const Inertia6& I, q4 r, v3 w) {
  const v3 wb = irot(w, r);
  const v3 Lb = v3{ffma(I.v[4], wb.z, ffma(I.v[3], wb.y, I.v[0] * wb.x)), ffma(I.v[5], wb.z, ffma(I.v[1], wb.y, I.v[3] * wb.x)),
                   ffma(I.v[2], wb.z, ffma(I.v[5], wb.y, I.v[4] * wb.x))};
  const v3 g = cross(wb, Lb);
  const v3 ab = v3{-ffma(ib[4], g.z, ffma(ib[3], g.y, ib[0] * g.x)), -ffma(ib[5], g.z, ffma(ib[1], g.y, ib[3] * g.x)),
                   -ffma(ib[2], g.z, ffma(ib[5], g.y, ib[4] * g.x))};
  return rot(ab, r);
}

template <int LPS, bool ISO, bool SLIDES, int MAXCH, int MAXCOL, int D0 = 0, int D1 = 0, int D2 = 0, int D3 = 0,
          bool DIAG = false, bool MULTI = true, int NS = 3, bool SLIDEW = false, bool AXI = false, int RK = -1, int NFR = 0,
          bool HELP = false, bool SKIP6 = false, bool SPEC = false>
__global__ __launch_bounds__(256) void rollout_kernel(RolloutParams P) {
  constexpr bool DPP = D0 != 0;
  static_assert(!SPEC || (!DPP && !HELP && !SKIP6 && MULTI), "SPEC: the general shuffle-exchange instantiations");
  // HELP: the link with more than two colliders runs stage (4) on slots 0, 1 and gets the corrections of its other
  // colliders from two helper lanes (LaneRec3); stage (6) stays MAXCOL slots in a row, every one from the velocities stage (5) left (Jacobi per link)
  static_assert(!HELP || (DPP && ISO && MAXCOL > 2 && MAXCOL <= 5), "helper lanes: isotropic DPP models with 3..5 colliders on one link");
  static_assert(!DPP || ((D1 != 0 || D2 == 0) && (D2 != 0 || D3 == 0) && (D3 == 0 || MAXCH >= 4)),
                "DPP layout: slots are filled in order");
  rollout_progress(P);  // (workgroup 0, whatever its job: the launch has started)
  const int rblock = rollout_block(P);  // (noise workgroups and the idle ones of a pinned launch are done here)
  if (rblock < 0) return;
  const unsigned long long dbg_t0 = P.dbg_clock ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long dbg_t1 = 0ull, dbg_t2 = 0ull, dbg_t3 = 0ull;  // (probes: prologue / first / second control step done)
  const mbd_model_t* __restrict__ Mg = P.model;  // the wave-uniform scalars of the model
  const int lane = threadIdx.x & 63;
  const int base = lane & ~(LPS - 1);
  const int l_lane = lane & (LPS - 1);
  // ---- per-lane model constants: the record lane_setup3_kernel gathered at env creation, one batch of reads --------
  const LaneRec3& R = reinterpret_cast<const LaneRec3*>(P.lane_rec[HELP ? 2 : (DPP ? 1 : 0)])[l_lane];
  const int L = Mg->n_links;
  const bool link_ok = R.link_ok != 0;
  const int l = R.l;
  const bool root_lane = R.root_lane != 0;  // the lane that owns link 0 (rewards, control cost)
  constexpr int SPW = 64 / LPS;
  // a workgroup is 1 or 4 INDEPENDENT wavefronts (nothing shared, no barrier): four-wave workgroups are how a launch
  // of >= 1024 wavefronts gets one wavefront on every SIMD of a CU (dispatch)
  const int wave_id = rblock * (blockDim.x >> 6) + (threadIdx.x >> 6);
#ifdef MBD_PROBE_3D_CPW1
  // (TIMING PROBE, variant builds only — docs/experiments.md §15: ONE candidate per wavefront, the other groups repeat it)
  const int b_raw = wave_id;
  const bool b_ok = b_raw < P.B && lane / LPS == 0;
  const int b = b_raw < P.B ? b_raw : P.B - 1;
#else
  const int b_raw = wave_id * SPW + lane / LPS;
  const bool b_ok = b_raw < P.B;
  const int b = b_ok ? b_raw : P.B - 1;
#endif
  const int H = P.H, Nu = Mg->n_act, nfr = NFR > 0 ? NFR : Mg->n_frames, K = Mg->n_track;

  const int nr = R.nr;
  const int plane = base + R.plane_rel;  // lane holding the parent (self if world)
  const bool world_parent = R.world_parent != 0;
  Inert<ISO> ic, ip;
  ic.inv_mass = R.ic_inv_mass;
  ip.inv_mass = R.ip_inv_mass;
#pragma unroll
  for (int k = 0; k < (ISO ? 1 : 6); ++k) { ic.ib[k] = R.ic_ib[k]; ip.ib[k] = R.ip_ib[k]; }  // (AXI: stored remapped)
  JointConst jc;
  jc.ap_pos = mk3(R.ap_pos[0], R.ap_pos[1], R.ap_pos[2]);
  jc.ac_pos = mk3(R.ac_pos[0], R.ac_pos[1], R.ac_pos[2]);
  jc.ap_rot = q4{R.ap_rot[0], R.ap_rot[1], R.ap_rot[2], R.ap_rot[3]};
  jc.ac_rot = q4{R.ac_rot[0], R.ac_rot[1], R.ac_rot[2], R.ac_rot[3]};
  // non-joint lanes (free root, padding) are masked through their scalars: every contribution they
  // compute is then an exact zero
  const float ang_damp = R.ang_damp, vel_damp = R.vel_damp;
  const int nr_eff = R.nr_eff;
  const bool need_child_mask = Mg->n_rot[0] >= 0 && L >= LPS;  // no lane contributes zeros: wave-uniform (scalar) flag
  float lim_lo[3], lim_hi[3], stiff[3], damp[3];
  v3 saxis[3], saxis_w[3];  // (slots a joint lacks: zero axes; saxis_w: the axes in the world frame, SLIDEW)
  float sl_lo[3], sl_hi[3], sl_damp[3];
  int act_rot[3], act_sl[3];
  float gear_rot[3], gear_sl[3], alo_rot[3], ahi_rot[3], alo_sl[3], ahi_sl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lim_lo[k] = R.lim_lo[k]; lim_hi[k] = R.lim_hi[k]; stiff[k] = R.stiff[k]; damp[k] = R.damp[k];
    saxis[k] = mk3(R.saxis[k][0], R.saxis[k][1], R.saxis[k][2]);
    saxis_w[k] = mk3(R.saxis_w[k][0], R.saxis_w[k][1], R.saxis_w[k][2]);
    sl_lo[k] = R.sl_lo[k]; sl_hi[k] = R.sl_hi[k]; sl_damp[k] = R.sl_damp[k];
    act_rot[k] = R.act_rot[k]; act_sl[k] = R.act_sl[k];
    gear_rot[k] = R.gear_rot[k]; gear_sl[k] = R.gear_sl[k];
    alo_rot[k] = R.alo_rot[k]; ahi_rot[k] = R.ahi_rot[k]; alo_sl[k] = R.alo_sl[k]; ahi_sl[k] = R.ahi_sl[k];
  }
  auto slide_dir = [&](int k, q4 aprot) { return SLIDEW ? saxis_w[k] : rot(saxis[k], aprot); };
  int child_lane[MAXCH], child_src[MAXCH];  // child_src: lane to pull child c's contribution from (a zero lane when none)
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    child_lane[c] = R.child_lane_rel[c] >= 0 ? base + R.child_lane_rel[c] : -1;
    child_src[c] = base + R.child_src_rel[c];
  }
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
#pragma unroll
    for (int k = 0; k < (MAXCH < 4 ? MAXCH : 4); ++k) { rm[k] = R.rm[k]; pm[k] = R.pm[k]; }
  }
  v3 col_pos[MAXCOL];
  float col_rad[MAXCOL];
  bool col_has[MAXCOL];
#pragma unroll
  for (int j = 0; j < MAXCOL; ++j) {
    col_has[j] = R.col_has[j] != 0; col_rad[j] = R.col_rad[j];
    col_pos[j] = mk3(R.col_pos[j][0], R.col_pos[j][1], R.col_pos[j][2]);
  }
  const int track_k = R.track_k;
  // (HELP: whose pose this lane's colliders ride on, where an owner's helpers sit — LaneRec3)
  const int help_src = base + (HELP ? R.help_src_rel : l_lane);
  const bool help_owner = HELP && R.is_owner != 0, help_helper = HELP && R.is_helper != 0;
  const v3 com = mk3(R.com[0], R.com[1], R.com[2]);
  const float dt = Mg->dt, inv_dt = 1.0f / Mg->dt, vel_fac = Mg->vel_fac, ang_fac = Mg->ang_fac;
  const float two_inv_dt = 2.0f * inv_dt;
  const float js_pos = R.js_pos, js_ang = R.js_ang;
  const float coll_scale = Mg->collide_scale, invm_sum = R.invm_sum;
  // isotropic models: the shares of an angular correction, (-ib_p, ib_c)/(ib_p + ib_c) * joint_scale_ang
  // (zero on non-joint lanes through js_ang)
  const f2 kang2 = ISO ? mk2(R.kang2[0], R.kang2[1]) : mk2(0.0f, 0.0f);
  const float mu = Mg->friction, elast = Mg->elasticity;
  const v3 grav = mk3(Mg->gravity[0], Mg->gravity[1], Mg->gravity[2]);
  const int rkind = RK >= 0 ? RK : Mg->reward_kind;
  const float rp0 = Mg->reward_params[0], rp1 = Mg->reward_params[1];
  const float dt_ctrl = Mg->dt * (float)nfr;
  // SPEC: the model's specification switches, wave-uniform (0 in every other instantiation: the tests below fold away)
  const int spec = SPEC ? (Mg->flags & MBD_SPEC_FLAGS) : MBD_TUNED_SPEC;  // (a constant in the tuned instantiations)
  constexpr bool SPEC_AVG = SPEC || (MBD_TUNED_SPEC & MBD_FLAG_CONTACT_AVG) != 0;  // (code that only this switch needs)
  const bool sp_avg = (spec & MBD_FLAG_CONTACT_AVG) != 0, sp_gs = (spec & MBD_FLAG_CONTACT6_GAUSS_SEIDEL) != 0;
  const bool sp_fvel = (spec & MBD_FLAG_FRICTION_VEL_BOUND) != 0, sp_rmin = (spec & MBD_FLAG_RESTITUTION_MIN) != 0;
  const bool sp_ext = (spec & MBD_FLAG_EULER_EXTRINSIC) != 0, sp_gyro = !ISO && (spec & MBD_FLAG_GYROSCOPIC) != 0;
  float g_ib[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // the link's own inverse inertia as the model has it (AXI remaps ic.ib)
  Inertia6 g_I;
#pragma unroll
  for (int k = 0; k < 6; ++k) g_I.v[k] = 0.0f;
  if constexpr (SPEC && !ISO) {
    if (sp_gyro) {
#pragma unroll
      for (int k = 0; k < 6; ++k) g_ib[k] = Mg->inv_inertia[l][k];
      g_I = inertia_from_inverse(g_ib);
    }
  }

  // ---- state -------------------------------------------------------------------------------------------
  const int pl = plan_of(P, b);
  const float* s0 = P.state0 + (size_t)pl * P.plan_state_stride + l * MBD_LINK_STATE;
  v3 p = mk3(s0[0], s0[1], s0[2]);
  q4 r = q4{s0[3], s0[4], s0[5], s0[6]};
  v3 v = mk3(s0[7], s0[8], s0[9]);
  v3 w = mk3(s0[10], s0[11], s0[12]);
  if (!link_ok) { p = mk3(0, 0, 0); r = q4{1, 0, 0, 0}; v = mk3(0, 0, 0); w = mk3(0, 0, 0); }

  const float* u_row = P.us + (size_t)b * H * Nu;
  // lazy candidates (wave-uniform): u_row holds normals, the action is clip(eps * sigma + Ybar_i[t][a], -1, 1)
  // (branch-free: the Ybar loads are unconditional — from P.us itself, ignored, when the launch is not lazy — and the
  // choice is a select; branches around loads make the compiler wait for every load in flight at the joins)
  const bool lazy = P.ybar != nullptr;
  const float* __restrict__ yb_row = lazy ? P.ybar + (size_t)pl * P.plan_ybar_stride : P.us;
  const float sigma = P.sigma;
  auto cand = [&](float e, float yb) {  // (mul, add: the sampler's roundings)
    const float c = fclip(e * sigma + yb, -1.0f, 1.0f);
    return lazy ? c : e;
  };
  float u_rot[3], u_sl[3];     // actions of the current control step (lazy: its normals)
  float un_rot[3], un_sl[3];   // actions of the next one, in flight while the substeps run
  float y_rot[3] = {0.0f, 0.0f, 0.0f}, y_sl[3] = {0.0f, 0.0f, 0.0f};      // lazy: Ybar_i of the current control step
  float yn_rot[3] = {0.0f, 0.0f, 0.0f}, yn_sl[3] = {0.0f, 0.0f, 0.0f};    // ... of the next one
  auto load_actions = [&](int t, float (&ur)[3], float (&us)[3], float (&yr)[3], float (&ys)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ur[k] = u_row[(size_t)t * Nu + (act_rot[k] >= 0 ? act_rot[k] : 0)];
      if constexpr (SLIDES) us[k] = u_row[(size_t)t * Nu + (act_sl[k] >= 0 ? act_sl[k] : 0)];
      else us[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      yr[k] = yb_row[(size_t)t * Nu + (act_rot[k] >= 0 ? act_rot[k] : 0)];
      if constexpr (SLIDES) ys[k] = yb_row[(size_t)t * Nu + (act_sl[k] >= 0 ? act_sl[k] : 0)];
    }
  };
  load_actions(0, u_rot, u_sl, y_rot, y_sl);
  // control cost (halfcheetah, ant: sum of squares of the WHOLE action row, in actuator order): the row of the next
  // control step travels with the other prefetched actions — fetched where it is used, its Nu dependent round trips
  // per control step were 10 % of the halfcheetah rollout.  Every lane of a candidate loads the same row (a broadcast).
  // (only in the single-hinge instantiations — ant, halfcheetah; the humanoid family has no such reward and keeps its
  // registers: KCC = 0 leaves the immediate loads)
  constexpr int KCC = MULTI ? 0 : 8;  // prefetched entries; rows beyond that finish with immediate loads
  const bool want_cc = rkind == MBD_REW_HALFCHEETAH || rkind == MBD_REW_ANT;  // wave-uniform
  float cc_u[KCC + 1], cc_y[KCC + 1], ccn_u[KCC + 1], ccn_y[KCC + 1];
  auto load_row = [&](int t, float (&ru)[KCC + 1], float (&ry)[KCC + 1]) {
#pragma unroll
    for (int k = 0; k < KCC; ++k) {
      ru[k] = u_row[(size_t)t * Nu + (k < Nu ? k : 0)];
      ry[k] = yb_row[(size_t)t * Nu + (k < Nu ? k : 0)];
    }
  };
#pragma unroll
  for (int k = 0; k < KCC; ++k) { cc_u[k] = cc_y[k] = ccn_u[k] = ccn_y[k] = 0.0f; }
  if (want_cc && KCC > 0) load_row(0, cc_u, cc_y);
  // the parent's POSE for the next substep is fetched as soon as it is final (end of stage 4), so that its
  // LDS round trip overlaps the velocity stages; only the parent's velocities are fetched at the top
  v3 Pp_next = shfl3(p, plane);
  q4 Pr_next = shfl4(r, plane);
  float rew_sum = 0.0f;
  float lp_acc = 0.0f;  // (fused demo log-density: this lane's S_k)
  const float* __restrict__ xref_base = P.xref ? P.xref : (const float*)P.model;

  for (int t = 0; t < H; ++t) {
    if (__builtin_expect(P.dbg_clock != nullptr && t <= 2, 0)) {  // (probes only; wave-uniform, out of line)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if (t == 0) dbg_t1 = now; else if (t == 1) dbg_t2 = now; else dbg_t3 = now;
    }
    // actuator.to_tau: clip to ctrlrange, times gear
    float tau[3], tau_sl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u_rot[k] = cand(u_rot[k], y_rot[k]);
      if constexpr (SLIDES) u_sl[k] = cand(u_sl[k], y_sl[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      tau[k] = fclip(act_rot[k] >= 0 ? u_rot[k] : 0.0f, alo_rot[k], ahi_rot[k]) * gear_rot[k];
      tau_sl[k] = SLIDES ? fclip(act_sl[k] >= 0 ? u_sl[k] : 0.0f, alo_sl[k], ahi_sl[k]) * gear_sl[k] : 0.0f;
    }
    float ctrl_cost = 0.0f;
    if (want_cc) {
#pragma unroll
      for (int k = 0; k < KCC; ++k) {
        const float ua = cand(cc_u[k], cc_y[k]);
        ctrl_cost = k < Nu ? ctrl_cost + ua * ua : ctrl_cost;
      }
      for (int a = KCC; a < Nu; ++a) {
        const float ua = cand(u_row[(size_t)t * Nu + a], yb_row[(size_t)t * Nu + a]);
        ctrl_cost = ctrl_cost + ua * ua;
      }
    }
    // issue the next control step's action loads now (clamped index: no branch) and keep them in flight
    // across the n_frames substeps; they are consumed at the top of the next iteration
    load_actions(t + 1 < H ? t + 1 : t, un_rot, un_sl, yn_rot, yn_sl);
    if (want_cc && KCC > 0) load_row(t + 1 < H ? t + 1 : t, ccn_u, ccn_y);
    // (fused demo log-density) this control step's reference position of the lane's tracked link, in flight across the
    // substeps.  UNCONDITIONAL loads (under `if (P.lp)` the merge of the loaded value with the other side's zero put the full
    // wait for the loads at the top of every control step: +14 us per rollout): the address is valid whatever the launch
    // wants — xref is [K][50][3]; launches without one read the model's bytes — and a launch without lp ignores the sum.
    v3 xr = mk3(0, 0, 0);
    if constexpr (RK == MBD_REW_HUMANOIDTRACK) {
      const float* c = xref_base + ((size_t)(track_k >= 0 ? track_k : 0) * 50 + (t < 50 ? t : 49)) * 3;
      xr = mk3(c[0], c[1], c[2]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // link-frame origin before the step (rewards that look at the incoming state / finite differences)
    const v3 o0 = sub(p, rot(com, r));
    const v3 v0 = sub(v, cross(w, rot(com, r)));

    // One substep; the loop below runs it two at a time (its back edge is a TAKEN branch: 30-60 cycles of
    // instruction-buffer refill for a lone wavefront, see mbd_planar.h).
    // The renormalisations' rare exact side CAN be speculated away (qnormalize_qm, mbd_math.h: the substeps run with QM = 1, the
    // control step is re-run from its saved start with QM = 2 when some lane left the series' range) — what gave the planar
    // kernels +4 ... 6.5 % (mbd_planar.h).  Here it LOSES 2.3 % (metric 1790 -> 1748 steps/s, humanoidtrack2048 2331 -> 2293,
    // same box, profiles/r06_spec3d_ab.txt): without the two branches this compiler emits 867 instead of 846 instructions per
    // humanoid substep (+10 v_mov, +10 unpacked v_fma, two more s_waitcnt; scheduling barriers where the branches were do not
    // bring them back), and the compare-to-branch latency it removes was evidently hidden already.  Off; -DMBD_3D_SPECULATE
    // builds it (instantiations with n_frames compiled in: the run-time loop of the others crashes this compiler's register
    // allocator under the iterative scheduler), bit-identical (148 GPU tests).
#ifdef MBD_3D_SPECULATE
    constexpr int QM_FAST = NFR > 0 ? 1 : 0;
#else
    constexpr int QM_FAST = 0;
#endif
    float q_worst = 0.0f;
    const v3 s_p = p, s_v = v, s_w = w;  // the control step's start
    const q4 s_r = r;
    auto substep_qm = [&](auto qm_tag) __attribute__((always_inline)) {
      constexpr int QM = decltype(qm_tag)::value;
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
        v3 ax0 = f.Xp, ax2 = f.Zc;  // gimbal axes of dofs 0 and 2 (the middle one: f.ax1)
        if constexpr (SPEC) {
          if (sp_ext) {  // (per lane: joints with two or three hinge dofs)
            float e0, e1, e2; v3 x0, x1, x2;
            euler_extrinsic(f.aprot, f.acrot, e0, e1, e2, x0, x1, x2);
            const bool ext = nr_eff >= 2;
            f.ang0 = ext ? e0 : f.ang0; f.ang1 = ext ? e1 : f.ang1; f.ang2 = ext ? e2 : f.ang2;
            ax0 = sel3(ext, x0, ax0); f.ax1 = sel3(ext, x1, f.ax1); ax2 = sel3(ext, x2, ax2);
          }
        }
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
        torque(0, ax0, f.ang0);
        if (multi) {
          torque(1, f.ax1, f.ang1);
          torque(2, ax2, f.ang2);
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
          } else if constexpr (D0 == 1 && D1 == -2 && D2 == -4 && D3 == -6) {  // (ant: one block for the four slots)
            dpp_acc6x4_ant(sv, sw, fp_v, fp_w, rm[0], rm[1], rm[2], rm[3]);
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
      const v3 av = lo3(acc);
      v3 aw = hi3(acc);
      if constexpr (SPEC && !ISO) {
        if (sp_gyro) aw = add(aw, gyro_accel(g_ib, g_I, r, w));  // (pose and velocity the substep started with)
      }
      if (!have_damped) { damped_vx = vel_fac * v.x; damped_vy = vel_fac * v.y; }
      v = mk3(ffma(av.x + grav.x, dt, damped_vx), ffma(av.y + grav.y, dt, damped_vy),
              ffma(av.z + grav.z, dt, vel_fac * v.z));
      w = mk3(ffma(aw.x, dt, ang_fac * w.x), ffma(aw.y, dt, ang_fac * w.y), ffma(aw.z, dt, ang_fac * w.z));
      const v3 p_prev = p;
      const q4 r_prev = r;
      p = mk3(ffma(v.x, dt, p.x), ffma(v.y, dt, p.y), ffma(v.z, dt, p.z));
      r = qrotvec_qm<QM>(r, scale(w, dt), q_worst);
      if constexpr (QM == 1) __builtin_amdgcn_sched_barrier(0);  // (the scheduling regions the branch used to delimit)
      // ---- (3) joints.position_update (Jacobi) ------------------------------------------------------
      if constexpr (DPP && D0 == 1 && D1 == -2 && D2 == -4 && D3 == -6) {
        dpp_fetch7x4_ant(p, r, pm[0], pm[1], pm[2], pm[3], Pp, Pr);
      } else if constexpr (DPP) {
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
        v3 ax0 = f.Xp, ax2 = f.Zc;
        v3 alignA = f.Xp, alignB = f.Yc;  // two hinge dofs: Yc stays perpendicular to Xp (extrinsic: Yp to Xc)
        if constexpr (SPEC) {
          if (sp_ext) {
            float e0, e1, e2; v3 x0, x1, x2;
            euler_extrinsic(f.aprot, f.acrot, e0, e1, e2, x0, x1, x2);
            const bool ext = nr_eff >= 2;
            f.ang0 = ext ? e0 : f.ang0; f.ang1 = ext ? e1 : f.ang1; f.ang2 = ext ? e2 : f.ang2;
            ax0 = sel3(ext, x0, ax0); f.ax1 = sel3(ext, x1, f.ax1); ax2 = sel3(ext, x2, ax2);
            alignA = sel3(ext, qaxes(f.aprot).Y, alignA); alignB = sel3(ext, f.Xc, alignB);
          }
        }
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
        v3 A = sel3(nr == 1, f.Xc, alignA);
        v3 Bv = sel3(nr == 1, f.Xp, alignB);
        const float dxy = dot(alignA, alignB);
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
          c0 = ang_prepare<ISO, AXI>(scale(ax0, -viol_of(0, f.ang0)), ip, ic, W2);
          if (multi) {
            c1 = ang_prepare<ISO, AXI>(scale(f.ax1, -viol_of(1, f.ang1)), ip, ic, W2);
            c2_ = ang_prepare<ISO, AXI>(scale(ax2, -viol_of(2, f.ang2)), ip, ic, W2);
          }
        };
        f2 q_ta, q01;  // (translation, alignment) and (limit 0, limit 1) quotients
        float q2;
        if constexpr (ISO) {
          E = axpy(-viol_of(0, f.ang0), ax0, E);
          if (multi) {
            E = axpy(-viol_of(1, f.ang1), f.ax1, E);
            E = axpy(-viol_of(2, f.ang2), ax2, E);
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
          } else if constexpr (D0 == 1 && D1 == -2 && D2 == -4 && D3 == -6) {
            dpp_acc6x4_ant(sp, sth, dp_p, dp_th, rm[0], rm[1], rm[2], rm[3]);
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
#ifdef MBD_PROBE_NO_CONTACT
        // (TIMING PROBE, variant builds only: the substep of a candidate that never touches — WRONG physics, the upper bound of
        // what a contact early-out could save)
        constexpr bool kContactCode = false;
#else
        constexpr bool kContactCode = true;
#endif
        if constexpr (!kContactCode) {
#pragma unroll
          for (int j = 0; j < MAXCOL; ++j) { con_pos[j] = mk3(0, 0, 0); con_dlam[j] = 0.0f; con_act[j] = false; }
        } else
        if constexpr (MAXCOL == 2 || HELP) {
          // both colliders of the link as one packed pair (the solve is Jacobi: each sees the pose of the stage's
          // start); their corrections are then added in collider order, exactly like the loop below
          // (HELP: a helper lane's two slots are colliders of the OWNER link: they ride on the owner's pose, fetched here —
          // every other lane fetches its own)
          v3 hp = p, hpp = p_prev;
          q4 hr = r, hrp = r_prev;
          if constexpr (HELP) {
            const int hs = help_src;
            hp = shfl3(p, hs); hr = shfl4(r, hs); hpp = shfl3(p_prev, hs); hrp = shfl4(r_prev, hs);
            shfl_join();
          }
          const q4x2 R2 = bcast4(hr);
          const v3x2 cpos = pack3(col_pos[0], col_pos[1]);
          const f2 rad = mk2(col_rad[0], col_rad[1]);
          const v3x2 off = rot2(cpos, R2);
          const v3x2 ctr = add2(bcast3(hp), off);
          const f2 pen = rad - ctr.z;
          const bool act0 = col_has[0] && pen.x > 0.0f, act1 = col_has[1] && pen.y > 0.0f;
          const f2 h = fma2(mk2(-0.5f, -0.5f), pen, rad);
          const v3x2 pos = v3x2{ctr.x, ctr.y, ctr.z - h};
          const v3x2 rc = v3x2{off.x, off.y, off.z - h};
          const v3x2 cn = v3x2{rc.y, -rc.x, mk2(0.0f, 0.0f)};
          const v3x2 icn = iinv_z0_s2<ISO, AXI>(ic, Wc, cn);
          const f2 wn = mk2(ic.inv_mass, ic.inv_mass) + dot_az0_2(cn, icn);
          const v3x2 rl = add2(cpos, irot_z2(-h, R2));
          const v3x2 pprev = add2(bcast3(hpp), rot2(rl, bcast4(hrp)));
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
          if constexpr (HELP) {
            // the owner collects colliders 2, 3 (helper 0, its slots 0, 1) and 4 (helper 1, slot 0): impulse, rotation,
            // contact point, multiplier, flag — and continues the sum in collider order with the loop's own operations;
            // every other lane reads itself and adds nothing (its flags are cleared), a helper's own sum is dropped
            const bool owner = help_owner, helper = help_helper;
            const float f0 = act0 ? 1.0f : 0.0f, f1 = act1 ? 1.0f : 0.0f;
            // (the helpers sit 7 and 8 lanes above their owner in the humanoid layout — the host checked — so the way back
            // is a DPP row shift per value: one VALU slot, no LDS round trip; what the shifts deliver to the other lanes
            // is never used: their flags are cleared)
            auto up7 = [](v3 a) { return v3{dpp_from<7>(a.x), dpp_from<7>(a.y), dpp_from<7>(a.z)}; };
            auto up8 = [](v3 a) { return v3{dpp_from<8>(a.x), dpp_from<8>(a.y), dpp_from<8>(a.z)}; };
            cd_p = sel3(helper, mk3(0, 0, 0), cd_p);
            cd_th = sel3(helper, mk3(0, 0, 0), cd_th);
#pragma unroll
            for (int j = 2; j < MAXCOL; ++j) { con_pos[j] = mk3(0, 0, 0); con_dlam[j] = 0.0f; con_act[j] = false; }
            // (SKIP6: the collection — 39 row shifts and the continued sums — sits behind "some helper lane of the
            // wavefront has a contact": without one every flag it would deliver is clear and it adds nothing)
            if (!SKIP6 || __builtin_expect(__builtin_amdgcn_ballot_w64(helper && (act0 || act1)) != 0ull, 0)) {
              const v3 g_P[3] = {up7(P0), up7(P1), up8(P0)};
              const v3 g_th[3] = {up7(dth0), up7(dth1), up8(dth0)};
              const v3 g_pos[3] = {up7(lo3(pos)), up7(hi3(pos)), up8(lo3(pos))};
              const float g_dl[3] = {dpp_from<7>(dlam.x), dpp_from<7>(dlam.y), dpp_from<8>(dlam.x)};
              const float g_f[3] = {dpp_from<7>(f0), dpp_from<7>(f1), dpp_from<8>(f0)};
#pragma unroll
              for (int j = 2; j < MAXCOL; ++j) {
                const bool a = owner && g_f[j - 2] != 0.0f;
                cd_p = sel3(a, axpy(ic.inv_mass, g_P[j - 2], cd_p), cd_p);
                cd_th = sel3(a, add(cd_th, g_th[j - 2]), cd_th);
                con_pos[j] = g_pos[j - 2]; con_dlam[j] = g_dl[j - 2]; con_act[j] = a;
              }
            }
            con_act[0] = act0 && !helper; con_act[1] = act1 && !helper;  // (a helper has no velocities of its own to correct)
          }
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
        if constexpr (SPEC_AVG && MAXCOL > 1) {
          int n_act = 0;
#pragma unroll
          for (int j = 0; j < MAXCOL; ++j) n_act += con_act[j] ? 1 : 0;
          if (sp_avg) {  // the average over the link's active contacts (two or more; one: untouched)
            const float inv_n = 1.0f / (float)(n_act > 1 ? n_act : 1);  // (exactly 1 for a single contact: no select needed)
            cd_p = scale(cd_p, inv_n);
            cd_th = scale(cd_th, inv_n);
          }
        }
        p = add(p, cd_p);  // zero corrections on links without colliders
        r = qrotvec_qm<QM>(r, cd_th, q_worst);
        if constexpr (QM == 1) __builtin_amdgcn_sched_barrier(0);
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
      // (SKIP6: every slot behind the first sits behind a ballot and a branch that is not taken while it does not touch;
      // its body is out of line)
      // Jacobi per link: every contact of the link sees the velocities stage (5) left, the changes are added in collider
      // order (SPEC, contact6_gauss_seidel: each sees what the link's previous contacts left)
      const v3 v6 = v, w6 = w;
      auto slot6 = [&](int j) __attribute__((always_inline)) {
          v3 rc = sub(con_pos[j], p);
          const v3 see_v = sp_gs ? v : v6, see_w = sp_gs ? w : w6;
          v3 vpt = add(see_v, cross(see_w, rc));
          // restitution needs the pre-solve normal velocity only when elasticity != 0 (wave-uniform); with
          // e = 0 the term max(-e*vn_prev, 0) is exactly 0
          float vn_prev = 0.0f;
          if (elast != 0.0f) vn_prev = add(v_old, cross(w_old, rc)).z;
          float vn = vpt.z;
          v3 vt = mk3(vpt.x, vpt.y, 0.0f);
          float vtn = sqrt_floor(ffma(vt.x, vt.x, vt.y * vt.y));
          float inv = rcp_exact(vtn + 1e-10f);  // (a finite speed plus 1e-10)
          v3 dir = mk3(vt.x * inv, vt.y * inv, 0.0f);
          v3 cn = crossz(rc), cdv = cross_bz0(rc, dir);
          v3 icn = iinv_z0<ISO, AXI>(ic, Wc, cn), icd = iinv<ISO, AXI>(ic, Wc, cdv);
          float wn = ic.inv_mass + dot_az0(cn, icn), wt = ic.inv_mass + dot(cdv, icd);
          float rest = -elast * vn_prev;
          float dvn = (sp_rmin ? fmin_(rest, 0.0f) : fmax_(rest, 0.0f)) - vn;  // (the floor's normal is +z: an approaching contact has vn_prev < 0)
          float jt_max = (mu * con_dlam[j]) * inv_dt;
          float dvt = fmin_(sp_fvel ? jt_max : jt_max * wt, vtn);
          const f2 q_nt = div2_sp_(mk2(dvn, dvt), mk2(wn, wt));
          float jn = q_nt.x, jt = -q_nt.y;
          v3 Pimp = mk3(dir.x * jt, dir.y * jt, jn);
          v3 nv = axpy(ic.inv_mass, Pimp, v);
          v3 nw = add(w, iinv<ISO, AXI>(ic, Wc, cross(rc, Pimp)));
          v = sel3(con_act[j], nv, v);
          w = sel3(con_act[j], nw, w);
      };
#ifndef MBD_PROBE_NO_CONTACT
#pragma unroll
      for (int j = 0; j < MAXCOL; ++j) {
        // (one test PER slot: with one test for all of them humanoidstandup's default plan ran 1161 instead of 1284 steps/s —
        // its torso usually rests on one or two of its spheres, and the others' slots stay skipped)
        if (SKIP6 && j > 0 && __builtin_expect(__builtin_amdgcn_ballot_w64(con_act[j]) == 0ull, 1)) continue;
        slot6(j);
      }
#endif
      if constexpr (SPEC_AVG && MAXCOL > 1) {
        if (!sp_gs && sp_avg) {  // the average of the link's velocity changes: v6 + (v - v6) / n
          int n_act = 0;
#pragma unroll
          for (int j = 0; j < MAXCOL; ++j) n_act += con_act[j] ? 1 : 0;
          const float inv_n = 1.0f / (float)(n_act > 1 ? n_act : 1);
          const v3 av_ = v3{ffma(v.x - v6.x, inv_n, v6.x), ffma(v.y - v6.y, inv_n, v6.y), ffma(v.z - v6.z, inv_n, v6.z)};
          const v3 aw_ = v3{ffma(w.x - w6.x, inv_n, w6.x), ffma(w.y - w6.y, inv_n, w6.y), ffma(w.z - w6.z, inv_n, w6.z)};
          v = sel3(n_act >= 2, av_, v);
          w = sel3(n_act >= 2, aw_, w);
        }
      }
    };
    auto substep = [&]() __attribute__((always_inline)) { substep_qm(std::integral_constant<int, QM_FAST>{}); };
    {
      // code placement: the loop starts where the fewest of its 8-byte instructions straddle a 32-byte fetch boundary
      // (tools/tune_phase.py; the built-in humanoid instantiations)
      phase_pad<(LPS == 16 && ISO && !SLIDES && MULTI && D0 == 1 && D1 == -4 && D2 == -6 && D3 == 0) ? mbd_pad_3d(MAXCOL, RK) : 0>();
      // (all 7 substeps of the humanoid in line: -1.9 % on the metric config — 39 KB of loop body)
      if constexpr (NFR > 1) {
        for (int it = 0; it < 2; ++it) repeat_n<NFR / 2>(substep);
        if constexpr (NFR % 2 != 0) substep();
      } else {
        int fr = 0;
        for (; fr + 1 < nfr; fr += 2) { substep(); substep(); }
        if (fr < nfr) substep();
      }
    }  // substeps
    if constexpr (QM_FAST == 1) {
      // (wave-uniform test; a NaN compares false, like the branch it replaces)
      if (__builtin_expect(__builtin_amdgcn_fcmpf(q_worst, 0.05f, 2 /* ogt */) != 0ull, 0)) {
        p = s_p; r = s_r; v = s_v; w = s_w;
        Pp_next = shfl3(p, plane);
        Pr_next = shfl4(r, plane);
        for (int fr = 0; fr < nfr; ++fr) substep_qm(std::integral_constant<int, 2>{});
      }
    }

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
      cart_cos = shfl(cs, base + R.lane_of1_rel);          // cos of link 1's hinge angle, fetched by the root lane
    }
    {  // (every lane evaluates it — a handful of instructions — and the root lane's value is the one kept: under
       // `if (root_lane)` the block sat out of line behind two taken branches per control step)
      float rew;
      if (rkind == MBD_REW_HUMANOIDRUN) {
        rew = o1.x * 1.0f - fclip(fabs_(o1.z - 1.3f), -1.0f, 1.0f) * 1.0f - fabs_(o1.y) * 0.1f;
      } else if (rkind == MBD_REW_HOPPER) {
        rew = o1.x - fclip(fabs_(o1.z - rp0), -1.0f, 1.0f) * rp1;
      } else if (rkind == MBD_REW_HALFCHEETAH) {
        rew = rp0 * ((o1.x - o0.x) / dt_ctrl) - rp1 * ctrl_cost;
      } else if (rkind == MBD_REW_ANT) {
        // reward_params[5] != 0: terminate_when_unhealthy (the stock setting) — the healthy term is unconditional
        float healthy = (Mg->reward_params[5] != 0.0f || (o1.z >= Mg->reward_params[2] && o1.z <= Mg->reward_params[3]))
                            ? Mg->reward_params[4] : 0.0f;
        rew = (rp0 * ((o1.x - o0.x) / dt_ctrl) + healthy) - rp1 * ctrl_cost;
      } else if (rkind == MBD_REW_CARTPOLE) {
        rew = cart_cos - fabs_(cart_vs);
      } else if (rkind == MBD_REW_HUMANOIDSTANDUP) {
        rew = 1.5f - fclip(fabs_(o1.z - 1.3f), -2.0f, 1.0f) - fabs_(o1.x) * 0.1f - fabs_(o1.y) * 0.1f;
      } else {
        rew = 1.0f + (-fabs_(v0.x - 1.6f) - fabs_(o0.z - 1.3f) - fabs_(o0.y) * 0.1f);
      }
      rew_sum = rew_sum + rew;
      if (__builtin_expect(root_lane && b_ok && P.rewss != nullptr, 1)) P.rewss[(size_t)b * H + t] = rew;
    }
    // (block placement: the expected side of a branch is the fall-through one, and a lone wavefront pays 30-60 cycles
    // for a TAKEN branch — the reward row is wanted by every plan; positions only by API callers since round 6: the tracking
    // reward's plans get their demo log-density from the accumulation below)
    if (__builtin_expect(P.xpos != nullptr && track_k >= 0 && b_ok, 0)) {
      float* o = P.xpos + (((size_t)b * H + t) * K + track_k) * 3;
      o[0] = o1.x; o[1] = o1.y; o[2] = o1.z;
    }
    if constexpr (RK == MBD_REW_HUMANOIDTRACK) {  // (every lane evaluates it; the tracked links' lanes are read at the end)
      const float ex = o1.x - xr.x, ey = o1.y - xr.y, ez = o1.z - xr.z;
      float d = fsqrt(ex * ex + ey * ey + ez * ez);
      d = fclip(d, 0.0f, 0.5f);
      const float sd = d / 0.5f;
      lp_acc = lp_acc + sd * sd;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { u_rot[k] = un_rot[k]; u_sl[k] = un_sl[k]; y_rot[k] = yn_rot[k]; y_sl[k] = yn_sl[k]; }
#pragma unroll
    for (int k = 0; k < KCC; ++k) { cc_u[k] = ccn_u[k]; cc_y[k] = ccn_y[k]; }
  }  // control steps
  if (P.dbg_clock && lane == 0) {
    P.dbg_clock[wave_id * 6 + 3] = dbg_t1;
    P.dbg_clock[wave_id * 6 + 4] = dbg_t2;
    P.dbg_clock[wave_id * 6 + 5] = dbg_t3;
    P.dbg_clock[wave_id * 6 + 0] = dbg_t0;
    P.dbg_clock[wave_id * 6 + 1] = __builtin_amdgcn_s_memtime();
    P.dbg_clock[wave_id * 6 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |
                                      ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
  }
  if (root_lane && b_ok && P.rews) P.rews[b] = rew_sum / (float)H;
  if constexpr (RK == MBD_REW_HUMANOIDTRACK) {
    if (P.lp != nullptr) {  // (wave-uniform) S_0 + S_1 + ... in link order, each fetched from its lane: the sum of one value
      float acc = 0.0f;     // and fifteen zeros over the candidate's lanes is that value, whatever the order
      for (int k = 0; k < K; ++k) {
        float sk = track_k == k ? lp_acc : 0.0f;
#pragma unroll
        for (int off = LPS / 2; off >= 1; off >>= 1) sk = sk + __shfl_xor(sk, off, 64);
        acc = k == 0 ? sk : acc + sk;
      }
      if (root_lane && b_ok) P.lp[b] = 0.0f - acc / (float)(K * H);
    }
  }
  if (P.state_final && link_ok && b_ok) {
    float* o = P.state_final + ((size_t)b * L + l) * MBD_LINK_STATE;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = r.w; o[4] = r.x; o[5] = r.y; o[6] = r.z;
    o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = w.x; o[11] = w.y; o[12] = w.z;
  }
}


}  // namespace mbd
