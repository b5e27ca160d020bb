// mbd_planar.h — the rollout kernel for PLANAR models (mbd_model_t.flags & MBD_FLAG_PLANAR: hopper, walker2d,
// halfcheetah, cartpole): the same six stages of the positional rigid-body step as rollout_kernel (mbd_kernels.h), on
// the in-plane coordinates only — position (x, z), orientation as the half-angle pair (w, y) of the unit quaternion
// (w, 0, y, 0), velocity (x, z), angular velocity about y: 7 floats per link instead of 13, a rotation is a 2x2
// product with (cos, sin) built once per stage, the inertia about y is a constant, the hinge angle comes from the
// relative half-angle pair, an angular correction is linear in its (scalar) error.  ONE LINK PER LANE, LPS lanes per
// candidate, parent <-> child traffic by DPP row shifts when the tree fits a family (every built-in model) and by
// ds_bpermute otherwise, state in registers for the whole rollout — the layout of rollout_kernel.
//
// This is a SPECIFICATION of its own for these models (the 3-D float arithmetic is not exactly planar: the joint
// frames are quarter turns with irrational components, a rollout leaks 1e-5..1e-3 out of the plane), selected by a
// flag of the MODEL and shared with the CPU checker, which the tests hold it to bit for bit.
#pragma once

#include <type_traits>

#include "mbd_kernels.h"

namespace mbd {

struct PCs { float c, s; };  // cos / sin of the FULL angle about +y
__device__ __forceinline__ PCs pl_cs(float w, float y) {
  const float y2 = y + y;
  return PCs{ffma(-y2, y, 1.0f), y2 * w};
}
__device__ __forceinline__ void pl_rot(PCs a, float x, float z, float& ox, float& oz) {
  ox = ffma(a.s, z, a.c * x);
  oz = ffma(-a.s, x, a.c * z);
}
__device__ __forceinline__ float pl_cross(float rx, float rz, float fx, float fz) { return ffma(rz, fx, -(rx * fz)); }
__device__ __forceinline__ void pl_rel(float Pw, float Py, float Cw, float Cy, float& wr, float& yr) {
  wr = ffma(Pw, Cw, Py * Cy);
  yr = ffma(Pw, Cy, -(Py * Cw));
}
__device__ __forceinline__ float pl_angle(float wr, float yr) {
  const float sn = (wr + wr) * yr;
  const float cn = ffma(-(yr + yr), yr, 1.0f);
  return angle_unit(sn, cn);
}
// the same on (parent, child) or (collider 0, collider 1) pairs: one v_pk_* issue slot for both, identical roundings
struct PCs2 { f2 c, s; };
__device__ __forceinline__ PCs2 pl_cs2(f2 w, f2 y) {
  const f2 y2 = y + y;
  return PCs2{fma2(-y2, y, mk2(1.0f, 1.0f)), y2 * w};
}
__device__ __forceinline__ void pl_rot2(PCs2 a, f2 x, f2 z, f2& ox, f2& oz) {
  ox = fma2(a.s, z, a.c * x);
  oz = fma2(-a.s, x, a.c * z);
}
__device__ __forceinline__ f2 pl_cross2(f2 rx, f2 rz, f2 fx, f2 fz) { return fma2(rz, fx, -(rx * fz)); }
__device__ __forceinline__ f2 bc2(float a) { return mk2(a, a); }

// QM: how the renormalisation's rare exact side (|n2 - 1| > 0.05: a link turning by more than 0.45 rad in ONE substep) is handled.
//   0  a branch where it happens (v_cmp -> s_and_saveexec -> s_cbranch: ~40 cycles of compare-to-branch latency for a lone
//      wavefront per SIMD even when never taken, tools/probes/probe_branch.hip)
//   1  SPECULATIVE (the early-out instantiations): the series is used unconditionally and the largest |n2 - 1| seen is kept in
//      `worst` (one v_max); the caller tests it once per CONTROL step and, when it ever exceeded the bound, re-runs that control
//      step from its saved start with QM = 2
//   2  both sides computed, the exact one selected where it applies — the values of QM = 0, branch-free
template <bool NORMALIZE, int QM = 0>
__device__ __forceinline__ void pl_qupdate(float& w, float& y, float dth, float& worst) {
  const float h = 0.5f * dth;
  float nw = ffma(-h, y, w), ny = ffma(h, w, y);
  if constexpr (NORMALIZE) {
    const float n2 = ffma(nw, nw, ny * ny);
    const float e = n2 - 1.0f;
    float inv = ffma(ffma(ffma(ffma(0.2734375f, e, -0.3125f), e, 0.375f), e, -0.5f), e, 1.0f);
    if constexpr (QM == 1) {
      worst = fmax_(worst, fabs_(e));
    } else if constexpr (QM == 2) {
      const float exact = 1.0f / fsqrt(n2);
      inv = fabs_(e) > 0.05f ? exact : inv;
    } else {
      if (__builtin_expect(fabs_(e) > 0.05f, 0)) inv = 1.0f / fsqrt(n2);
    }
    nw = nw * inv; ny = ny * inv;
  }
  w = nw; y = ny;
}

// LPS lanes per candidate; D0, D1: DPP layout (lane(parent) = lane(s-th child) + Ds; D0 = 0: ds_bpermute exchange,
// up to kMaxChildren children); MAXCOL sphere colliders per link (0: the contact stages are compiled out; 2: the stages as one
// packed pair; 4 — round 6, models whose links carry three or four spheres: the halfcheetah under collide_all_capsules, its
// torso and head capsules on one link — as two packed pairs; SPEC and a tuned Gauss-Seidel stage (6): collider by collider)
// FL: the model's wave-uniform switches as COMPILE-TIME constants — bit 0: some hinge has a joint spring (any_stiff),
// bit 1: some slide dof has a finite range (slide_limits), bit 2: elasticity != 0 — or -1: read them at run time.  As
// run-time flags each is a taken forward branch per substep for the models that lack the feature (every built-in one
// lacks two or three), and a lone wavefront pays for a taken branch with a refill of its instruction buffer.
// RK: the model's reward kind as a compile-time constant (-1: run time), like rollout_kernel's.
// NFR: the model's n_frames (even) as a compile-time constant, or 0: read at run time.  The substep loop then runs
// twice over NFR / 2 substeps in line instead of NFR / 4 times over four plus a remainder loop: every iteration saved is
// a taken branch (hopper, n_frames = 20: 7 -> 2 per control step, +2.7 % at N = 512; the in-line body stays under
// ~30 KB — the humanoid's seven substeps in line, 39 KB, lost 1.9 %).
// SPEC: the model's specification switches that exist in the plane (mbd_model_flags: contact_avg, contact6_gauss_seidel,
// friction_vel_bound, restitution_min — DESIGN.md §9) are read at run time and honoured as the checker's planar
// restatement states them; one general instantiation is built with it (models carrying such a bit run there), every other one
// compiles the default specification in.
// EO (round 6): the launch puts P.cpw < 64 / LPS candidates on a wavefront (launches that would leave SIMDs idle anyway:
// hopper512's 32 wavefronts become 512, one candidate each) and the substep takes a WAVE-UNIFORM EARLY-OUT around the
// contact code: once the penetrations of stage (4) are known and no lane has a sphere below the plane, the rest of stage
// (4) and all of stage (6) are exact no-ops (every effect of theirs sits behind a select on `active`), so the wavefront
// branches around them — a single hopper candidate is airborne in 87-90 % of its substeps, a halfcheetah in 43-57 %,
// which 16 (8) candidates sharing a wavefront never are together.  The lanes of the groups beyond P.cpw REPEAT the
// wavefront's candidates (group g runs candidate g mod cpw, stores nothing), so the predicate is that of the cpw
// candidates alone.  Results are bit-identical for every cpw (the skipped path keeps the stage's two additions of +0 and
// its quaternion renormalisation).
template <int LPS, int MAXCOL, int D0 = 0, int D1 = 0, int FL = -1, int RK = -1, int NFR = 0, bool SPEC = false, bool EO = false>
__global__ __launch_bounds__(256) void rollout_planar_kernel(RolloutParams P) {
  static_assert(NFR % 2 == 0, "NFR: two iterations of NFR / 2");
  static_assert(!SPEC || (D0 == 0 && FL < 0), "SPEC: the general shuffle-exchange instantiation");
  static_assert(!EO || (MAXCOL == 2 && !SPEC), "EO: the packed two-collider contact stages");
  static_assert(MAXCOL == 0 || MAXCOL == 2 || MAXCOL == 4, "MAXCOL: colliders per link come in packed pairs");
  static_assert(!EO || (MBD_TUNED_SPEC & MBD_FLAG_CONTACT6_GAUSS_SEIDEL) == 0, "EO: stage (6) as a packed pair (Jacobi)");
  constexpr bool DPP = D0 != 0;
  // the renormalisations' rare exact side is SPECULATED away (pl_qupdate QM = 1: no compare-to-branch latency in the substep;
  // a control step in which it would have been taken is re-run).  MBD_PLANAR_NO_SPECULATE: the branches of rounds 1-5 (A/B).
#ifdef MBD_PLANAR_NO_SPECULATE
  constexpr bool SPECULATE = EO;
#else
  constexpr bool SPECULATE = true;
#endif
  constexpr int NSLOT = DPP ? (D1 != 0 ? 2 : 1) : kMaxChildren;
  rollout_progress(P);
  const int rblock = rollout_block(P);  // (noise workgroups and the idle ones of a pinned launch are done here: mbd_kernels.h)
  if (rblock < 0) return;
  const mbd_model_t* __restrict__ M = P.model;
  const int lane = threadIdx.x & 63;
  const int base = lane & ~(LPS - 1);
  const int l_lane = lane & (LPS - 1);
  const int L = M->n_links;
  const int l_link = DPP ? (int)P.lane_tab[l_lane] : l_lane;
  const bool link_ok = l_link >= 0 && l_link < L;
  const int l = link_ok ? l_link : 0;
  auto lane_of = [&](int link) { return base + (DPP ? (int)P.lane_tab[16 + link] : link); };
  const bool root_lane = link_ok && l == 0;
  constexpr int SPW = 64 / LPS;
  const int wave_id = rblock * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // candidates per wavefront: SPW, or (EO) the launch's choice P.cpw, a power of two below it
  const int cpw = EO ? P.cpw : SPW;
  const int grp = lane / LPS;
  const int b_first = wave_id * cpw;
  const int b_raw = b_first + (EO ? (grp & (cpw - 1)) : grp);
  const bool b_ok = b_raw < P.B && grp < cpw;
  // (lanes without a candidate of their own repeat one of the wavefront's: the tail of the launch, the groups beyond cpw)
  const int b = b_raw < P.B ? b_raw : (EO && b_first < P.B ? b_first : P.B - 1);
  const int H = P.H, Nu = M->n_act, nfr = NFR > 0 ? NFR : M->n_frames;

  // ---- per-lane model constants (padding lanes: everything that scales a contribution is zero) ------------
  const int parent = M->parent[l];
  const bool world_parent = parent < 0;
  const int plane = (link_ok && parent >= 0) ? lane_of(parent) : lane;
  const int nr = link_ok ? M->n_rot[l] : 0, ns = link_ok ? M->n_slide[l] : 0;
  const float im_c = link_ok ? M->inv_mass[l] : 0.0f, iy_c = link_ok ? M->inv_inertia[l][1] : 0.0f;
  const float im_p = (link_ok && !world_parent) ? M->inv_mass[parent] : 0.0f;
  const float iy_p = (link_ok && !world_parent) ? M->inv_inertia[parent][1] : 0.0f;
  const float invm_sum = im_p + im_c;
  const float apx = M->ap_pos[l][0], apz = M->ap_pos[l][2], acx = M->ac_pos[l][0], acz = M->ac_pos[l][2];
  const float sg = (M->ap_rot[l][0] * M->ap_rot[l][3] < 0.0f) ? -1.0f : 1.0f;
  const float wpar = world_parent ? 1.0f : 0.0f;  // the world's orientation (1, 0): added to the fetched zero
  // (parent, child) pairs of the per-joint constants
  const f2 anc_x = mk2(apx, acx), anc_z = mk2(apz, acz);
  const f2 im2 = mk2(-im_p, im_c), iy2 = mk2(iy_p, iy_c), iy2s = mk2(-iy_p, iy_c);
  float sx[2], sz[2], sl_lo[2], sl_hi[2], sl_damp[2];
  {
    const q4 aprot = q4{M->ap_rot[l][0], M->ap_rot[l][1], M->ap_rot[l][2], M->ap_rot[l][3]};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool has = link_ok && j < ns;
      const v3 s = rot(mk3(M->slide_axis[l][j][0], M->slide_axis[l][j][1], M->slide_axis[l][j][2]), aprot);
      sx[j] = has ? s.x : 0.0f; sz[j] = has ? s.z : 0.0f;
      sl_lo[j] = has ? M->slide_lo[l][j] : -1e9f; sl_hi[j] = has ? M->slide_hi[l][j] : 1e9f;
      sl_damp[j] = M->slide_damp[l][j];
    }
  }
  const float lim_lo = M->rot_lo[l][0], lim_hi = M->rot_hi[l][0];
  const float stiff = M->rot_stiff[l][0], damp = M->rot_damp[l][0];
  const float ang_damp = link_ok ? M->ang_damp[l] : 0.0f, vel_damp = link_ok ? M->vel_damp[l] : 0.0f;
  const float js_pos = link_ok ? M->joint_scale_pos : 0.0f, js_ang = M->joint_scale_ang;
  float kc = 0.0f, kp = 0.0f;
  if (link_ok) { kc = (iy_c / (iy_p + iy_c)) * js_ang; kp = (iy_p / (iy_p + iy_c)) * js_ang; }
  int act_rot = -1, act_sl[2] = {-1, -1};
  float gear_rot = 0.0f, alo_rot = 0.0f, ahi_rot = 0.0f, gear_sl[2] = {0.0f, 0.0f}, alo_sl[2] = {0.0f, 0.0f}, ahi_sl[2] = {0.0f, 0.0f};
  for (int a = 0; a < Nu; ++a) {
    if (M->act_link[a] != l || !link_ok) continue;
    const int s = M->act_slot[a];
    if (s == 0) { act_rot = a; gear_rot = M->act_gear[a]; alo_rot = M->act_lo[a]; ahi_rot = M->act_hi[a]; }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (s == 3 + j) { act_sl[j] = a; gear_sl[j] = M->act_gear[a]; alo_sl[j] = M->act_lo[a]; ahi_sl[j] = M->act_hi[a]; }
  }
  // children: DPP masks (rm[s]: this link has an s-th child at lane - Ds; pm[s]: this link is the s-th child of its
  // parent at lane + Ds) or source lanes of the shuffle exchange (a missing child: own lane, masked)
  float rm[2] = {0.0f, 0.0f}, pm[2] = {0.0f, 0.0f};
  int child_src[NSLOT];
  float child_m[NSLOT];
  {
    int nc = 0;
#pragma unroll
    for (int c = 0; c < NSLOT; ++c) { child_src[c] = lane; child_m[c] = 0.0f; }
    for (int c = l + 1; c < L; ++c)
      if (M->parent[c] == l && link_ok) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j)
          if (j == nc) { child_src[j] = lane_of(c); child_m[j] = 1.0f; }
        ++nc;
      }
    if constexpr (DPP) {
      int myslot = -1;
      if (link_ok && parent >= 0) {
        myslot = 0;
        for (int c = 0; c < l; ++c) myslot += M->parent[c] == parent ? 1 : 0;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) { rm[k] = (k < NSLOT && child_m[k < NSLOT ? k : 0] != 0.0f) ? 1.0f : 0.0f; pm[k] = myslot == k ? 1.0f : 0.0f; }
    }
  }
  float colx[MAXCOL > 0 ? MAXCOL : 1], colz[MAXCOL > 0 ? MAXCOL : 1], col_rad[MAXCOL > 0 ? MAXCOL : 1];
  bool col_has[MAXCOL > 0 ? MAXCOL : 1];
  if constexpr (MAXCOL > 0) {
    int nc = 0;
#pragma unroll
    for (int j = 0; j < MAXCOL; ++j) { col_has[j] = false; col_rad[j] = 0.0f; colx[j] = colz[j] = 0.0f; }
    for (int k = 0; k < M->n_col; ++k)
      if (M->col_link[k] == l && link_ok) {
#pragma unroll
        for (int j = 0; j < MAXCOL; ++j)
          if (j == nc) { col_has[j] = true; col_rad[j] = M->col_radius[k]; colx[j] = M->col_pos[k][0]; colz[j] = M->col_pos[k][2]; }
        ++nc;
      }
  }
  // (EO) the radii with -inf where the lane has no such collider: its "penetration" in the early-out test is then -inf
  const f2 radb2 = mk2((MAXCOL > 0 && col_has[0]) ? col_rad[0] : -__builtin_inff(),
                       (MAXCOL > 1 && col_has[MAXCOL > 1 ? 1 : 0]) ? col_rad[MAXCOL > 1 ? 1 : 0] : -__builtin_inff());
  const float comx = M->com[l][0], comz = M->com[l][2];
  const float dt = M->dt, inv_dt = 1.0f / M->dt, vel_fac = M->vel_fac, ang_fac = M->ang_fac;
  const float two_inv_dt = 2.0f * inv_dt;
  const float coll_scale = M->collide_scale, mu = M->friction, elast = M->elasticity;
  const float gx = link_ok ? M->gravity[0] : 0.0f, gz = link_ok ? M->gravity[2] : 0.0f;
  const int rkind = RK >= 0 ? RK : M->reward_kind;
  const float rp0 = M->reward_params[0], rp1 = M->reward_params[1];
  const float dt_ctrl = M->dt * (float)nfr;
  const int spec = SPEC ? (M->flags & MBD_SPEC_FLAGS) : MBD_TUNED_SPEC;  // (wave-uniform; a constant in the tuned instantiations: the tests below fold away)
  constexpr bool SPEC_AVG = SPEC || (MBD_TUNED_SPEC & MBD_FLAG_CONTACT_AVG) != 0;
  constexpr bool TUNED_GS = !SPEC && (MBD_TUNED_SPEC & MBD_FLAG_CONTACT6_GAUSS_SEIDEL) != 0;  // (stage (6) cannot be a packed pair then)
  const bool sp_avg = (spec & MBD_FLAG_CONTACT_AVG) != 0, sp_gs = (spec & MBD_FLAG_CONTACT6_GAUSS_SEIDEL) != 0;
  const bool sp_fvel = (spec & MBD_FLAG_FRICTION_VEL_BOUND) != 0, sp_rmin = (spec & MBD_FLAG_RESTITUTION_MIN) != 0;

  // ---- exchange -----------------------------------------------------------------------------------------
  auto from_parent = [&](float v) -> float {  // the value of v in the parent's lane (0 for a world parent)
    if constexpr (DPP) {
      float o = dpp_from<D0>(v) * pm[0];
      if constexpr (D1 != 0) o = ffma(dpp_from<D1>(v), pm[1], o);
      return o;
    } else {
      const float o = shfl(v, plane);
      return world_parent ? 0.0f : o;
    }
  };
  auto add_children = [&](float own, float contrib) -> float {  // own + child 0 + child 1 + ..., in child order
    if constexpr (DPP) {
      float o = ffma(dpp_from<-D0>(contrib), rm[0], own);
      if constexpr (D1 != 0) o = ffma(dpp_from<-D1>(contrib), rm[1], o);
      return o;
    } else {
      float o = own;
#pragma unroll
      for (int c = 0; c < NSLOT; ++c) {
        if (c < P.max_children) o = ffma(shfl(contrib, child_src[c]), child_m[c], o);
      }
      return o;
    }
  };

  // ---- state -------------------------------------------------------------------------------------------
  const int pl = plan_of(P, b);
  const float* s0 = P.state0 + (size_t)pl * P.plan_state_stride + l * MBD_LINK_STATE;
  float px = s0[0], pz = s0[2], qw = s0[3], qy = s0[5], vx = s0[7], vz = s0[9], om = s0[11];
  if (!link_ok) { px = pz = 0.0f; qw = 1.0f; qy = 0.0f; vx = vz = om = 0.0f; }

  const float* u_row = P.us + (size_t)b * H * Nu;
  // lazy candidates (wave-uniform; RolloutParams): u_row holds normals, the action is clip(eps * sigma + Ybar_i, -1, 1)
  // (branch-free, like rollout_kernel: unconditional Ybar loads — from P.us itself, ignored, when not lazy — and selects)
  const bool lazy = P.ybar != nullptr;
  const float* __restrict__ yb_row = lazy ? P.ybar + (size_t)pl * P.plan_ybar_stride : P.us;
  const float sigma = P.sigma;
  auto cand = [&](float e, float yb) {
    const float c = fclip(e * sigma + yb, -1.0f, 1.0f);
    return lazy ? c : e;
  };
  auto load_u = [&](int t, int a) { return u_row[(size_t)t * Nu + (a >= 0 ? a : 0)]; };
  auto load_y = [&](int t, int a) { return yb_row[(size_t)t * Nu + (a >= 0 ? a : 0)]; };
  float u_rot = load_u(0, act_rot), u_sl0 = load_u(0, act_sl[0]), u_sl1 = load_u(0, act_sl[1]);
  float y_rot = load_y(0, act_rot), y_sl0 = load_y(0, act_sl[0]), y_sl1 = load_y(0, act_sl[1]);
  // control cost (halfcheetah): the whole action row of the next control step travels with the other prefetched
  // actions (mbd_kernels.h: fetched where it is used, its Nu dependent round trips were 10 % of the rollout)
  constexpr int KCC = 8;
  const bool want_cc = rkind == MBD_REW_HALFCHEETAH;  // wave-uniform
  float cc_u[KCC], cc_y[KCC], ccn_u[KCC], ccn_y[KCC];
  auto load_row = [&](int t, float (&ru)[KCC], float (&ry)[KCC]) {
#pragma unroll
    for (int k = 0; k < KCC; ++k) {
      ru[k] = u_row[(size_t)t * Nu + (k < Nu ? k : 0)];
      ry[k] = yb_row[(size_t)t * Nu + (k < Nu ? k : 0)];
    }
  };
#pragma unroll
  for (int k = 0; k < KCC; ++k) { cc_u[k] = cc_y[k] = ccn_u[k] = ccn_y[k] = 0.0f; }
  if (want_cc) load_row(0, cc_u, cc_y);
  float rew_sum = 0.0f;

  for (int t = 0; t < H; ++t) {
    u_rot = cand(u_rot, y_rot); u_sl0 = cand(u_sl0, y_sl0); u_sl1 = cand(u_sl1, y_sl1);
    const float tau0 = fclip(act_rot >= 0 ? u_rot : 0.0f, alo_rot, ahi_rot) * gear_rot;
    float tau_sl[2];
    tau_sl[0] = fclip(act_sl[0] >= 0 ? u_sl0 : 0.0f, alo_sl[0], ahi_sl[0]) * gear_sl[0];
    tau_sl[1] = fclip(act_sl[1] >= 0 ? u_sl1 : 0.0f, alo_sl[1], ahi_sl[1]) * gear_sl[1];
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
    const int tn = t + 1 < H ? t + 1 : t;  // the next control step's actions, in flight across the substeps
    const float un_rot = load_u(tn, act_rot), un_sl0 = load_u(tn, act_sl[0]), un_sl1 = load_u(tn, act_sl[1]);
    const float yn_rot = load_y(tn, act_rot), yn_sl0 = load_y(tn, act_sl[0]), yn_sl1 = load_y(tn, act_sl[1]);
    if (want_cc) load_row(tn, ccn_u, ccn_y);
    __builtin_amdgcn_sched_barrier(0);
    float o0x, o0z;
    {
      float tx, tz;
      pl_rot(pl_cs(qw, qy), comx, comz, tx, tz);
      o0x = px - tx; o0z = pz - tz;
    }
    (void)o0z;

    // One substep; the loop below runs it four at a time: its back edge is a TAKEN branch, which costs a lone wavefront
    // 30-60 cycles of instruction-buffer refill — 3 % of a 350-instruction substep (n_frames is 20 / 16 / 4 for the
    // built-in planar models: the remainder loop never runs for them).
    float q_worst = 0.0f;  // the largest |n2 - 1| a renormalisation of this control step saw (pl_qupdate, QM = 1)
    auto substep_qm = [&](auto qm_tag) __attribute__((always_inline)) {
      constexpr int QM = decltype(qm_tag)::value;
      // ---- (1) joints.acceleration_update ----------------------------------------------------------------
      float Ppx = from_parent(px), Ppz = from_parent(pz), Pw = from_parent(qw) + wpar, Py = from_parent(qy);
      const float Pvx = from_parent(vx), Pvz = from_parent(vz), Pom = from_parent(om);
      float fcvx, fcvz, fcw, fpvx, fpvz, fpw;
      {
        // (parent, child) pairs: low half the parent side, high half the child side
        const PCs2 c2 = pl_cs2(mk2(Pw, qw), mk2(Py, qy));
        f2 rx, rz;
        pl_rot2(c2, anc_x, anc_z, rx, rz);  // the lever arms (rp, rc)
        const f2 om2 = mk2(Pom, om);
        const f2 vax = fma2(om2, rz, mk2(Pvx, vx)), vaz = fma2(-om2, rx, mk2(Pvz, vz));  // anchor velocities (vp, vc)
        float rvx = vax.y - vax.x, rvz = vaz.y - vaz.x;
        const float rw = om - Pom;
        float fk = ffma(-damp, sg * rw, tau0);
        if (FL >= 0 ? (FL & 1) != 0 : P.any_stiff != 0) {  // (wave-uniform) the hinge angle only feeds the joint spring: -0 * ang is an exact zero
          float wr, yr;
          pl_rel(Pw, Py, qw, qy, wr, yr);
          const float ang = sg * pl_angle(wr, yr);
          fk = ffma(-stiff, ang, fk);
        }
        fk = nr >= 1 ? fk : 0.0f;
        const float Ty = ffma(-ang_damp, rw, fk * sg);
        float Fx = 0.0f, Fz = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float vs = ffma(rvx, sx[j], rvz * sz[j]);
          const float fs = j < ns ? ffma(-sl_damp[j], vs, tau_sl[j]) : 0.0f;
          Fx = ffma(fs, sx[j], Fx); Fz = ffma(fs, sz[j], Fz);
          rvx = ffma(-vs, sx[j], rvx); rvz = ffma(-vs, sz[j], rvz);
        }
        Fx = ffma(-vel_damp, rvx, Fx); Fz = ffma(-vel_damp, rvz, Fz);
        const f2 linx = bc2(Fx) * im2, linz = bc2(Fz) * im2;                   // (-F/m_p, F/m_c)
        const f2 angw = (bc2(Ty) + pl_cross2(rx, rz, bc2(Fx), bc2(Fz))) * iy2;  // ((T + rp x F)/I_p, (T + rc x F)/I_c)
        fpvx = linx.x; fcvx = linx.y; fpvz = linz.x; fcvz = linz.y;
        fpw = -angw.x; fcw = angw.y;
      }
      // ---- (2) integrator.integrate_xdd ---------------------------------------------------------------------
      {
        const float ax = add_children(fcvx, fpvx), az = add_children(fcvz, fpvz), aw = add_children(fcw, fpw);
        vx = ffma(ax + gx, dt, vel_fac * vx);
        vz = ffma(az + gz, dt, vel_fac * vz);
        om = ffma(aw, dt, ang_fac * om);
      }
      const float pxp = px, pzp = pz, qwp = qw, qyp = qy;
      px = ffma(vx, dt, px);
      pz = ffma(vz, dt, pz);
      pl_qupdate<true, QM>(qw, qy, om * dt, q_worst);
      // ---- (3) joints.position_update (Jacobi) ---------------------------------------------------------------
      Ppx = from_parent(px); Ppz = from_parent(pz); Pw = from_parent(qw) + wpar; Py = from_parent(qy);
      float dcx, dcz, dcth, dpx, dpz, dpth;
      {
        const PCs2 c2p = pl_cs2(mk2(Pw, qw), mk2(Py, qy));
        f2 rx, rz;
        pl_rot2(c2p, anc_x, anc_z, rx, rz);
        const f2 awx = mk2(Ppx, px) + rx, awz = mk2(Ppz, pz) + rz;  // world anchors (ap, ac)
        float dx = awx.x - awx.y, dz = awz.x - awz.y;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float cf = -ffma(dx, sx[j], dz * sz[j]);
          dx = ffma(cf, sx[j], dx); dz = ffma(cf, sz[j], dz);
        }
        const float c2 = ffma(dx, dx, dz * dz);
        const f2 cr = pl_cross2(rx, rz, bc2(dx), bc2(dz));
        const f2 wq2 = cr * (iy2 * cr);
        const float den = ffma(invm_sum, c2, wq2.x + wq2.y) + 1e-20f;
        const float g = div_pos_(c2, den) * js_pos;
        const float Px = dx * g, Pz = dz * g;
        f2 lx2 = bc2(Px) * im2, lz2 = bc2(Pz) * im2;                      // (dp_p, dc_p)
        const f2 t2 = pl_cross2(rx, rz, bc2(Px), bc2(Pz)) * iy2;
        f2 th2 = mk2(-t2.x, t2.y);                                         // (dp_th, dc_th)
        if (FL >= 0 ? (FL & 2) != 0 : P.slide_limits != 0) {  // (wave-uniform)
          const float ex = awx.y - awx.x, ez = awz.y - awz.x;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float qs = ffma(ex, sx[j], ez * sz[j]);
            const float viol = qs - fclip(qs, sl_lo[j], sl_hi[j]);
            const float lx = sx[j] * (-viol), lz = sz[j] * (-viol);
            const float l2 = ffma(lx, lx, lz * lz);
            const f2 lcr = pl_cross2(rx, rz, bc2(lx), bc2(lz));
            const f2 lw = lcr * (iy2 * lcr);
            const float dens = ffma(invm_sum, l2, lw.x + lw.y);
            const float gs = div_pos_(l2, dens + 1e-20f) * js_pos;
            const float Sx = lx * gs, Sz = lz * gs;
            lx2 = fma2(bc2(Sx), im2, lx2); lz2 = fma2(bc2(Sz), im2, lz2);
            th2 = fma2(pl_cross2(rx, rz, bc2(Sx), bc2(Sz)), iy2s, th2);
          }
        }
        dpx = lx2.x; dcx = lx2.y; dpz = lz2.x; dcz = lz2.y;
        dpth = th2.x; dcth = th2.y;
        float wr, yr;
        pl_rel(Pw, Py, qw, qy, wr, yr);
        const float ang = sg * pl_angle(wr, yr);
        const float viol = ang - fclip(ang, lim_lo, lim_hi);
        const float E_hinge = (-viol) * sg;
        const float E_weld = __builtin_copysignf(2.0f, wr) * (-yr);
        const float E = nr < 1 ? E_weld : E_hinge;
        dcth = ffma(kc, E, dcth);
        dpth = ffma(-kp, E, dpth);
      }
      {
        const float ax = add_children(dcx, dpx), az = add_children(dcz, dpz), ath = add_children(dcth, dpth);
        px = px + ax; pz = pz + az;
        pl_qupdate<false>(qw, qy, ath, q_worst);  // renormalised at the end of stage (4)
      }
      // ---- (4) sphere-plane contacts + collisions.resolve_position -----------------------------------------
      float cposx[MAXCOL > 0 ? MAXCOL : 1], cposz[MAXCOL > 0 ? MAXCOL : 1], cdlam[MAXCOL > 0 ? MAXCOL : 1];
      bool cact[MAXCOL > 0 ? MAXCOL : 1];
      float vz_old, om_old;
      constexpr int J1 = MAXCOL > 1 ? 1 : 0;  // (the second collider's slot, where there is one)
      // ---- (5) integrator.project_xd ------------------------------------------------------------------------
      auto project_xd = [&]() __attribute__((always_inline)) {
        vz_old = vz; om_old = om;
        vx = (px - pxp) * inv_dt;
        vz = (pz - pzp) * inv_dt;
        const float dqw = ffma(qw, qwp, qy * qyp);
        const float dqy = ffma(qy, qwp, -(qw * qyp));
        om = dqy * __builtin_copysignf(two_inv_dt, dqw);
      };
      // ---- (6) collisions.resolve_velocity (Jacobi per link), both colliders of the link as one packed pair ----
      // (Jacobi makes them independent, like stage (4): every contact of the link computes its impulse from the velocities
      // stage (5) left; the changes are added in collider order)
      float vx6 = 0.0f, vz6 = 0.0f, om6 = 0.0f;  // what stage (5) left: every pair of the link computes from these
      auto resolve_velocity_pair = [&](auto pair_tag) __attribute__((always_inline)) {
        constexpr int Q = decltype(pair_tag)::value, C0 = MAXCOL > 1 ? 2 * Q : 0, C1 = MAXCOL > 1 ? 2 * Q + 1 : 0;
        if constexpr (Q == 0) { vx6 = vx; vz6 = vz; om6 = om; }
        const f2 rcx = mk2(cposx[C0], cposx[C1]) - bc2(px), rcz = mk2(cposz[C0], cposz[C1]) - bc2(pz);
        const f2 vptx = fma2(bc2(om6), rcz, bc2(vx6)), vptz = fma2(bc2(-om6), rcx, bc2(vz6));
        f2 vn_prev = bc2(0.0f);
        if (FL >= 0 ? (FL & 4) != 0 : elast != 0.0f) vn_prev = fma2(bc2(-om_old), rcx, bc2(vz_old));  // (wave-uniform; with e = 0 the term is exactly 0)
        const f2 vtn = __builtin_elementwise_abs(vptx);
        const f2 icn = rcx * bc2(iy_c);
        const f2 wn = fma2(icn, rcx, bc2(im_c));
        const f2 wt = fma2(rcz, rcz * bc2(iy_c), bc2(im_c));
        const f2 rest = bc2(-elast) * vn_prev;
        const f2 dvn = (sp_rmin ? mk2(fmin_(rest.x, 0.0f), fmin_(rest.y, 0.0f)) : mk2(fmax_(rest.x, 0.0f), fmax_(rest.y, 0.0f))) - vptz;
        const f2 jt_max = (bc2(mu) * mk2(cdlam[C0], cdlam[C1])) * bc2(inv_dt);
        const f2 jw = sp_fvel ? jt_max : jt_max * wt;
        const f2 dvt = mk2(fmin_(jw.x, vtn.x), fmin_(jw.y, vtn.y));
        f2 q_n, q_t;
        div2x2_sp_(dvn, wn, dvt, wt, q_n, q_t);
        const f2 Pix = mk2(-__builtin_copysignf(q_t.x, vptx.x), -__builtin_copysignf(q_t.y, vptx.y)), Piz = q_n;  // friction opposes the slip
        const f2 dom = pl_cross2(rcx, rcz, Pix, Piz) * bc2(iy_c);
        const float nvx0 = ffma(im_c, Pix.x, vx), nvz0 = ffma(im_c, Piz.x, vz), nom0 = om + dom.x;
        vx = cact[C0] ? nvx0 : vx; vz = cact[C0] ? nvz0 : vz; om = cact[C0] ? nom0 : om;
        float nvx1 = ffma(im_c, Pix.y, vx), nvz1 = ffma(im_c, Piz.y, vz), nom1 = om + dom.y;
        // (EO: the three stay SELECTS — left alone the compiler sinks them under an EXEC mask, and one divergent region anywhere
        // makes it linearise the early-out's uniform if / else through flag registers: a second branch on the common path)
        if constexpr (EO) asm volatile("" : "+v"(nvx1), "+v"(nvz1), "+v"(nom1));
        vx = cact[C1] ? nvx1 : vx; vz = cact[C1] ? nvz1 : vz; om = cact[C1] ? nom1 : om;
        if constexpr (SPEC_AVG && MAXCOL == 2) {
          if (sp_avg) {  // both touch: the average of the link's two velocity changes, v6 + (v - v6) / 2
            const bool both = cact[C0] && cact[C1];
            const f2 v62 = mk2(vx6, vz6);
            const f2 a2 = fma2(mk2(vx, vz) - v62, bc2(0.5f), v62);  // (the linear pair packed: same roundings per component)
            vx = both ? a2.x : vx;
            vz = both ? a2.y : vz;
            om = both ? ffma(om - om6, 0.5f, om6) : om;
          }
        }
      };
      {
        float cdx = 0.0f, cdz = 0.0f, cdth = 0.0f;
        if constexpr (MAXCOL == 2 || MAXCOL == 4) {
          // the link's colliders as packed PAIRS (0, 1) and, MAXCOL = 4, (2, 3) (the solve is Jacobi: each sees the pose of the
          // stage's start); their corrections are then added in collider order
          constexpr int NP = MAXCOL / 2;
          const PCs a = pl_cs(qw, qy);
          f2 cx2[NP], cz2[NP], rad2[NP], offx[NP], offz[NP], ctrx[NP], ctrz[NP], pen[NP];
          bool act0[NP], act1[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            cx2[q] = mk2(colx[2 * q], colx[2 * q + 1]); cz2[q] = mk2(colz[2 * q], colz[2 * q + 1]);
            rad2[q] = mk2(col_rad[2 * q], col_rad[2 * q + 1]);
            offx[q] = fma2(bc2(a.s), cz2[q], bc2(a.c) * cx2[q]); offz[q] = fma2(bc2(-a.s), cx2[q], bc2(a.c) * cz2[q]);
            ctrx[q] = bc2(px) + offx[q]; ctrz[q] = bc2(pz) + offz[q];
            pen[q] = rad2[q] - ctrz[q];
            act0[q] = col_has[2 * q] && pen[q].x > 0.0f; act1[q] = col_has[2 * q + 1] && pen[q].y > 0.0f;
          }
          // the rest of the stage: everything it changes sits behind act0 / act1
          auto resolve_position_pair = [&](auto pair_tag) __attribute__((always_inline)) {
            constexpr int Q = decltype(pair_tag)::value;
            const PCs ap = pl_cs(qwp, qyp);
            const f2 h = fma2(bc2(-0.5f), pen[Q], rad2[Q]);
            const f2 posx = ctrx[Q], posz = ctrz[Q] - h;
            const f2 rcx = offx[Q], rcz = offz[Q] - h;
            const f2 icn = rcx * bc2(iy_c);
            const f2 wn = fma2(icn, rcx, bc2(im_c));
            const f2 d = -h;
            const f2 rlx = fma2(bc2(-a.s), d, cx2[Q]), rlz = fma2(bc2(a.c), d, cz2[Q]);
            const f2 pprevx = bc2(pxp) + fma2(bc2(ap.s), rlz, bc2(ap.c) * rlx);
            const f2 ddx = posx - pprevx;
            // static friction: the tangent is the x axis, |d|^2 / (d.W d) of the 3-D form is 1 / (im + rcz^2 iy)
            const f2 wt = fma2(rcz, rcz * bc2(iy_c), bc2(im_c));
            f2 q_n, q_g;
            div2x2_(pen[Q], wn, bc2(1.0f), wt, q_n, q_g);
            const f2 dlam = q_n * bc2(coll_scale), sx = q_g * ddx;
            const f2 lim = bc2(mu) * dlam;
            const f2 lhs = sx * sx, rhs = lim * lim;
            const f2 Pix = mk2(lhs.x < rhs.x ? -sx.x : 0.0f, lhs.y < rhs.y ? -sx.y : 0.0f), Piz = dlam;
            const f2 dth = pl_cross2(rcx, rcz, Pix, Piz) * bc2(iy_c);
            cdx = act0[Q] ? ffma(im_c, Pix.x, cdx) : cdx;
            cdz = act0[Q] ? ffma(im_c, Piz.x, cdz) : cdz;
            cdth = act0[Q] ? cdth + dth.x : cdth;
            cdx = act1[Q] ? ffma(im_c, Pix.y, cdx) : cdx;
            cdz = act1[Q] ? ffma(im_c, Piz.y, cdz) : cdz;
            cdth = act1[Q] ? cdth + dth.y : cdth;
            if constexpr (!SPEC && MAXCOL == 2 && (MBD_TUNED_SPEC & MBD_FLAG_CONTACT_AVG) != 0) {  // both touch: half the summed correction
              // (SPEC, and four colliders: averaged below over the link's count.  One select of the factor — times 1 is exact —
              // and a packed product: 3 instructions where three selected products took 6)
              // (not in the early-out instantiations: packing the corrections there costs their common path three moves)
              const float half = (act0[Q] && act1[Q]) ? 0.5f : 1.0f;
              if constexpr (EO) {
                cdx = cdx * half; cdz = cdz * half; cdth = cdth * half;
              } else {
                const f2 c2 = mk2(cdx, cdz) * bc2(half);
                cdx = c2.x; cdz = c2.y; cdth = cdth * half;
              }
            }
            cposx[2 * Q] = posx.x; cposx[2 * Q + 1] = posx.y; cposz[2 * Q] = posz.x; cposz[2 * Q + 1] = posz.y;
            cdlam[2 * Q] = dlam.x; cdlam[2 * Q + 1] = dlam.y; cact[2 * Q] = act0[Q]; cact[2 * Q + 1] = act1[Q];
          };
          if constexpr (EO) {
            // WAVE-UNIFORM early-out: no sphere of any lane is below the plane.  What remains of stages (4) - (6) then:
            // the corrections stay (+0, +0, 0) — added and renormalised as always — stage (5), and a stage (6) whose every
            // select keeps the old value.
            // The test costs a lone wavefront its compare-to-branch latency (~40 cycles: tools/probes/probe_branch.hip), so
            // (a) it is ONE compare into vcc — the larger of the two penetrations, with -inf radii on the lanes that lack a
            // collider: no scalar mask arithmetic — and (b) the common side's pose update is computed between the compare
            // and the branch, where it is free (the side that has a contact discards it).
            const f2 penb = radb2 - ctrz[0];
            const bool touching = __builtin_amdgcn_fcmpf(fmax_(penb.x, penb.y), 0.0f, 2 /* ogt */) != 0ull;
            __builtin_amdgcn_sched_barrier(0);
            float fpx = px + cdx, fpz = pz + cdz, fqw = qw, fqy = qy;
            pl_qupdate<true, QM>(fqw, fqy, cdth, q_worst);
            asm volatile("" : "+v"(fpx), "+v"(fpz), "+v"(fqw), "+v"(fqy));  // (computed HERE: not sunk behind the branch)
            __builtin_amdgcn_sched_barrier(0);
            if (__builtin_expect(!touching, 1)) {
              px = fpx; pz = fpz; qw = fqw; qy = fqy;
              project_xd();
            } else {
              resolve_position_pair(std::integral_constant<int, 0>{});
              px = px + cdx; pz = pz + cdz;
              pl_qupdate<true, QM>(qw, qy, cdth, q_worst);
              project_xd();
              resolve_velocity_pair(std::integral_constant<int, 0>{});
            }
          } else {
            resolve_position_pair(std::integral_constant<int, 0>{});
            if constexpr (MAXCOL == 4) resolve_position_pair(std::integral_constant<int, 1>{});
          }
        } else         if constexpr (MAXCOL > 0) {
          const PCs a = pl_cs(qw, qy), ap = pl_cs(qwp, qyp);
#pragma unroll
          for (int j = 0; j < MAXCOL; ++j) {
            float offx, offz;
            pl_rot(a, colx[j], colz[j], offx, offz);
            const float ctrx = px + offx, ctrz = pz + offz;
            const float pen = col_rad[j] - ctrz;
            const bool active = col_has[j] && pen > 0.0f;
            const float h = ffma(-0.5f, pen, col_rad[j]);
            const float posx = ctrx, posz = ctrz - h;
            const float rcx = offx, rcz = offz - h;
            const float icn = rcx * iy_c;
            const float wn = ffma(icn, rcx, im_c);
            const float d = -h;
            const float rlx = ffma(-a.s, d, colx[j]), rlz = ffma(a.c, d, colz[j]);
            const float pprevx = pxp + ffma(ap.s, rlz, ap.c * rlx);
            const float ddx = posx - pprevx;
            const float wt = ffma(rcz, rcz * iy_c, im_c);
            const f2 q_ng = div2_pos_(mk2(pen, 1.0f), mk2(wn, wt));
            const float dlam = q_ng.x * coll_scale, sx = q_ng.y * ddx;
            const float lim = mu * dlam;
            const float Pix = (sx * sx < lim * lim) ? -sx : 0.0f, Piz = dlam;
            const float dth = pl_cross(rcx, rcz, Pix, Piz) * iy_c;
            cdx = active ? ffma(im_c, Pix, cdx) : cdx;
            cdz = active ? ffma(im_c, Piz, cdz) : cdz;
            cdth = active ? cdth + dth : cdth;
            cposx[j] = posx; cposz[j] = posz; cdlam[j] = dlam; cact[j] = active;
          }
        }
        if constexpr (SPEC_AVG && MAXCOL > 0 && (SPEC || MAXCOL != 2)) {  // (tuned, two colliders: averaged inside the pair above)
          if (sp_avg) {  // the average over the link's active contacts (two or more; one: untouched)
            int n_act = 0;
#pragma unroll
            for (int j = 0; j < MAXCOL; ++j) n_act += cact[j] ? 1 : 0;
            const float inv_n = 1.0f / (float)(n_act > 1 ? n_act : 1);
            cdx = cdx * inv_n; cdz = cdz * inv_n; cdth = cdth * inv_n;  // (inv_n is exactly 1 for a single contact)
          }
        }
        if constexpr (!EO) {
          px = px + cdx; pz = pz + cdz;
          pl_qupdate<true, QM>(qw, qy, cdth, q_worst);
        }
      }
      if constexpr (!EO) project_xd();
      // ---- (6) collisions.resolve_velocity (Jacobi per link) ---------------------------------------------------
      // every contact of the link computes its impulse from the velocities stage (5) left; the changes are added in collider
      // order (SPEC, contact6_gauss_seidel: one after the other, each from the running values)
      if constexpr (EO) {
        // (done above, on the path that had a contact)
      } else if constexpr ((MAXCOL == 2 || MAXCOL == 4) && !SPEC && !TUNED_GS) {
        resolve_velocity_pair(std::integral_constant<int, 0>{});
        if constexpr (MAXCOL == 4) {
          resolve_velocity_pair(std::integral_constant<int, 1>{});
          if constexpr (SPEC_AVG) {
            if (sp_avg) {  // the average of the link's velocity changes over its active contacts: v6 + (v - v6) / n
              int n_act = 0;
#pragma unroll
              for (int j = 0; j < MAXCOL; ++j) n_act += cact[j] ? 1 : 0;
              const float inv_n = 1.0f / (float)(n_act > 1 ? n_act : 1);
              vx = n_act >= 2 ? ffma(vx - vx6, inv_n, vx6) : vx;
              vz = n_act >= 2 ? ffma(vz - vz6, inv_n, vz6) : vz;
              om = n_act >= 2 ? ffma(om - om6, inv_n, om6) : om;
            }
          }
        }
      } else if constexpr (MAXCOL > 0) {
        vx6 = vx; vz6 = vz; om6 = om;  // what every contact of the link sees
#pragma unroll
        for (int j = 0; j < MAXCOL; ++j) {
          const float rcx = cposx[j] - px, rcz = cposz[j] - pz;
          const float svx = sp_gs ? vx : vx6, svz = sp_gs ? vz : vz6, som = sp_gs ? om : om6;
          const float vptx = ffma(som, rcz, svx), vptz = ffma(-som, rcx, svz);
          float vn_prev = 0.0f;
          if (FL >= 0 ? (FL & 4) != 0 : elast != 0.0f) vn_prev = ffma(-om_old, rcx, vz_old);  // (wave-uniform; with e = 0 the term is exactly 0)
          // (in the plane the slip direction is the sign of vptx: no normalising division, lever arm rcz)
          const float vtn = fabs_(vptx);
          const float icn = rcx * iy_c;
          const float wn = ffma(icn, rcx, im_c);
          const float wt = ffma(rcz, rcz * iy_c, im_c);
          const float rest = -elast * vn_prev;
          const float dvn = (sp_rmin ? fmin_(rest, 0.0f) : fmax_(rest, 0.0f)) - vptz;
          const float jt_max = (mu * cdlam[j]) * inv_dt;
          const float dvt = fmin_(sp_fvel ? jt_max : jt_max * wt, vtn);
          const f2 q_nt = div2_sp_(mk2(dvn, dvt), mk2(wn, wt));
          const float Pix = -__builtin_copysignf(q_nt.y, vptx), Piz = q_nt.x;  // friction opposes the slip
          const float nvx = ffma(im_c, Pix, vx), nvz = ffma(im_c, Piz, vz);
          const float nom = om + pl_cross(rcx, rcz, Pix, Piz) * iy_c;
          vx = cact[j] ? nvx : vx; vz = cact[j] ? nvz : vz; om = cact[j] ? nom : om;
        }
        if constexpr (SPEC_AVG) {
          if (!sp_gs && sp_avg) {  // the average of the link's velocity changes: v6 + (v - v6) / n
            int n_act = 0;
#pragma unroll
            for (int j = 0; j < MAXCOL; ++j) n_act += cact[j] ? 1 : 0;
            const float inv_n = 1.0f / (float)(n_act > 1 ? n_act : 1);
            vx = n_act >= 2 ? ffma(vx - vx6, inv_n, vx6) : vx;
            vz = n_act >= 2 ? ffma(vz - vz6, inv_n, vz6) : vz;
            om = n_act >= 2 ? ffma(om - om6, inv_n, om6) : om;
          }
        }
      }
    };
    auto substep = [&]() __attribute__((always_inline)) { substep_qm(std::integral_constant<int, SPECULATE ? 1 : 0>{}); };
    // the control step's start, for the rare re-run with the exact renormalisation
    const float s_px = px, s_pz = pz, s_qw = qw, s_qy = qy, s_vx = vx, s_vz = vz, s_om = om;
    {
      phase_pad<mbd_pad_planar(LPS, MAXCOL, D0, D1, FL, RK, NFR)>();  // (code placement: tools/tune_phase.py)
      int fr = 0;
      if constexpr (NFR > 0) {  // two iterations of NFR / 2 substeps in line
        for (int it = 0; it < 2; ++it) repeat_n<NFR / 2>(substep);
      } else {
        for (; fr + 3 < nfr; fr += 4) { substep(); substep(); substep(); substep(); }
        for (; fr < nfr; ++fr) substep();
      }
    }
    if constexpr (SPECULATE) {
      // some renormalisation of this control step left the series' range (wave-uniform test; NaN compares false, like the
      // branch it replaces): the control step again from its start, every renormalisation with its exact side selected
      if (__builtin_expect(__builtin_amdgcn_fcmpf(q_worst, 0.05f, 2 /* ogt */) != 0ull, 0)) {
        px = s_px; pz = s_pz; qw = s_qw; qy = s_qy; vx = s_vx; vz = s_vz; om = s_om;
        for (int fr = 0; fr < nfr; ++fr) substep_qm(std::integral_constant<int, 2>{});
      }
    }

    // ---- reward ------------------------------------------------------------------------------------------------
    float o1x, o1z;
    {
      float tx, tz;
      pl_rot(pl_cs(qw, qy), comx, comz, tx, tz);
      o1x = px - tx; o1z = pz - tz;
    }
    float cart_cos = 0.0f, cart_vs = 0.0f;
    if (rkind == MBD_REW_CARTPOLE) {  // cartpole.py:45: cos(q[1]) - |qd[0]| (wave-uniform branch)
      const float Pw = from_parent(qw) + wpar, Py = from_parent(qy);
      float wr, yr, sn, cs;
      pl_rel(Pw, Py, qw, qy, wr, yr);
      sincos_(sg * pl_angle(wr, yr), &sn, &cs);     // link 1's hinge angle (meaningful on link 1's lane)
      float rx, rz;
      pl_rot(pl_cs(qw, qy), acx, acz, rx, rz);
      const float vcx = ffma(om, rz, vx), vcz = ffma(-om, rx, vz);
      cart_vs = ffma(vcx, sx[0], vcz * sz[0]);     // own slide-0 velocity: used on link 0's lane
      cart_cos = shfl(cs, lane_of(1));
    }
    {  // (every lane evaluates it, the root lane's is kept; the store is the expected side: rollout_kernel's note)
      float rew;
      if (rkind == MBD_REW_HOPPER) {
        rew = o1x - fclip(fabs_(o1z - rp0), -1.0f, 1.0f) * rp1;
      } else if (rkind == MBD_REW_HALFCHEETAH) {
        rew = rp0 * ((o1x - o0x) / dt_ctrl) - rp1 * ctrl_cost;
      } else {
        rew = cart_cos - fabs_(cart_vs);
      }
      rew_sum = rew_sum + rew;
      if (__builtin_expect(root_lane && b_ok && P.rewss != nullptr, 1)) P.rewss[(size_t)b * H + t] = rew;
    }
    u_rot = un_rot; u_sl0 = un_sl0; u_sl1 = un_sl1;
    y_rot = yn_rot; y_sl0 = yn_sl0; y_sl1 = yn_sl1;
#pragma unroll
    for (int k = 0; k < KCC; ++k) { cc_u[k] = ccn_u[k]; cc_y[k] = ccn_y[k]; }
  }  // control steps
  if (root_lane && b_ok && P.rews) P.rews[b] = rew_sum / (float)H;
  if (P.state_final && link_ok && b_ok) {
    float* o = P.state_final + ((size_t)b * L + l) * MBD_LINK_STATE;
    o[0] = px; o[1] = 0.0f; o[2] = pz; o[3] = qw; o[4] = 0.0f; o[5] = qy; o[6] = 0.0f;
    o[7] = vx; o[8] = 0.0f; o[9] = vz; o[10] = 0.0f; o[11] = om; o[12] = 0.0f;
  }
}

}  // namespace mbd
