/*
 * mbd_oracle_physics.c — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into or imported by the
 * product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * CPU restatement of what the reference calls through `PipelineEnv.pipeline_step(..)` with
 * backend="positional" (call sites: mbd/envs/humanoidrun.py:17,29,36; hopper.py:18,30,40;
 * humanoidtrack.py:46,54,65) plus the env wrappers' rewards (humanoidrun.py:46-51, hopper.py:57-65,
 * humanoidtrack.py:87-96) and `rollout_us` (mbd/utils.py:14-20).
 *
 * PARITY UNPINNED.  The arithmetic lives in Brax (third-party; not vendored, not pinned and not even
 * declared by the reference's setup.py:8-24; absent from this container, as are jax and mujoco), so
 * this file restates the PUBLISHED algorithm the positional backend implements — maximal-coordinate
 * extended position-based dynamics, Müller et al. 2020, "Detailed Rigid Body Simulation with Extended
 * Position Based Dynamics" — in the stage order of brax/positional/pipeline.py::step as recalled in
 * SURVEY.md App. C:
 *     actuator.to_tau -> joints.acceleration_update (+gravity) -> integrator.integrate_xdd ->
 *     joints.position_update -> geometry.contact + collisions.resolve_position ->
 *     integrator.project_xd -> collisions.resolve_velocity
 * It is the SPECIFICATION the HIP kernels are tested against (teacher-forced, per diffusion step);
 * it is not bit-compatible with Brax and no claim of that is made.  tools/dump_golden.py produces
 * real Brax vectors when run under a jax+brax install; tests consume them if present.
 * What IS pinned to the reference's code (round 4): everything AROUND the physics — the wrappers' reward expressions, which
 * state each reads, resets, humanoidtrack's lag and step counter, n_frames per wrapper, rollout_us — by executing the
 * reference's wrappers and planner with THIS file under brax's PipelineEnv (tools/make_ref_golden.py run_brax,
 * tests/golden/ref_run_*.npz, tests/test_ref_golden.py).
 *
 * NUMERICAL CONTRACT: all arithmetic goes through oracle/spec_math.h (explicit fma order, exact div /
 * sqrt, polynomial atan2) and the per-link code is written in the same masked, lane-uniform form the
 * HIP kernel uses (a link = a lane; parent data fetched, children's contributions added in increasing
 * child index), so the kernel reproduces these results BIT FOR BIT.  See DESIGN.md §Numerics.
 *
 * Build:  -DORC_REAL=float (default, matches the reference's f32) or -DORC_REAL=double (used by the
 * tests to measure how much f32 round-off is amplified over a rollout).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mbd_hip.h"
#include "spec_math.h"

#define ORC_API __attribute__((visibility("default")))

typedef struct { real p[3]; real r[4]; } xf_t;
typedef struct { real v[3]; real w[3]; } mo_t;

typedef struct {
  real inv_mass;
  real ib[6];    /* body-frame inverse inertia xx yy zz xy xz yz */
  real W[6];     /* world-frame inverse inertia R Ib R^T (xx yy zz xy xz yz), see inert_refresh() */
  int iso;       /* model-wide: inverse inertia is ib[0] * identity */
  int world;     /* the static world: everything is zero */
  int axi;       /* model-wide: every tensor is axisymmetric about a link axis, see model_axisym() */
  real axa, axc; /* axi: Ib = axa Id + (axc - axa) u u^T */
  real u[3];     /* axi: the symmetry axis in the link frame */
  real Z[3];     /* axi: the world-frame symmetry axis R u, see inert_refresh() */
} inert_t;

/* Axisymmetric models (every body-frame inverse inertia is diagonal with two equal entries: capsules — hopper,
 * walker2d): Ib = a Id + (c - a) u u^T, R Ib R^T = a Id + (c - a) U U^T with U = R u.  For these the contract is that
 * closed form — only U is rebuilt per stage (sp_rot of the constant u) and the tensor is applied as
 * a v + (c - a)(U.v) U — not the general R Ib R^T product (same mathematics, different roundings; the kernels' AXI
 * instantiations follow it). */
static int model_axisym(const mbd_model_t* m) {
  if (m->iso_inertia) return 0;
  for (int l = 0; l < m->n_links; ++l) {
    const float* ib = m->inv_inertia[l];
    if (ib[3] != 0.0f || ib[4] != 0.0f || ib[5] != 0.0f) return 0;
    if (ib[0] != ib[1] && ib[1] != ib[2] && ib[0] != ib[2]) return 0;
  }
  return 1;
}
static void axi_setup(inert_t* in) { /* (a, c, u) from diag(xx, yy, zz): the odd one out is the axis */
  const real xx = in->ib[0], yy = in->ib[1], zz = in->ib[2];
  const int ez = xx == yy, ex = !ez && yy == zz;
  in->axa = ez ? xx : (ex ? yy : xx);
  in->axc = ez ? zz : (ex ? xx : yy);
  in->u[0] = ex ? R(1) : R(0); in->u[1] = (!ez && !ex) ? R(1) : R(0); in->u[2] = ez ? R(1) : R(0);
}

/* World-frame inverse inertia of a link at orientation r: W = R Ib R^T with R = (X Y Z) the axes of r,
 * T = R Ib first, then the six unique entries of T R^T (com.inv_inertia in Brax).  Re-evaluated at the
 * head of every stage that applies it — (1), (3), (4), (6) — from the orientation current there; inside
 * a stage the Jacobi solves all see that same tensor.  Isotropic models never call it. */
static void inert_refresh(inert_t* in, const real r[4]) {
  if (in->iso || in->world) return;
  if (in->axi) { sp_rot(in->u, r, in->Z); return; }
  real X[3], Y[3], Z[3], T[3][3];
  sp_qaxes(r, X, Y, Z);
  const real xx = in->ib[0], yy = in->ib[1], zz = in->ib[2], xy = in->ib[3], xz = in->ib[4], yz = in->ib[5];
  for (int i = 0; i < 3; ++i) {
    T[i][0] = sp_fma(Z[i], xz, sp_fma(Y[i], xy, X[i] * xx));
    T[i][1] = sp_fma(Z[i], yz, sp_fma(Y[i], yy, X[i] * xy));
    T[i][2] = sp_fma(Z[i], zz, sp_fma(Y[i], yz, X[i] * xz));
  }
#define ORC_W(i, j) sp_fma(T[i][2], Z[j], sp_fma(T[i][1], Y[j], T[i][0] * X[j]))
  in->W[0] = ORC_W(0, 0); in->W[1] = ORC_W(1, 1); in->W[2] = ORC_W(2, 2);
  in->W[3] = ORC_W(0, 1); in->W[4] = ORC_W(0, 2); in->W[5] = ORC_W(1, 2);
#undef ORC_W
}

/* world-frame inverse inertia applied to v: W v  (iso: ib0 * v) */
static inline void iinv_apply(const inert_t* in, const real v[3], real o[3]) {
  if (in->world) { sp_set3(o, 0, 0, 0); return; }
  if (in->iso) { sp_scale3(v, in->ib[0], o); return; }
  if (in->axi) {
    const real kd = (in->axc - in->axa) * sp_dot3(in->Z, v);
    real m0 = sp_fma(kd, in->Z[0], in->axa * v[0]), m1 = sp_fma(kd, in->Z[1], in->axa * v[1]);
    real m2 = sp_fma(kd, in->Z[2], in->axa * v[2]);
    sp_set3(o, m0, m1, m2);
    return;
  }
  const real* W = in->W;
  real m[3];
  m[0] = sp_fma(W[4], v[2], sp_fma(W[3], v[1], W[0] * v[0]));
  m[1] = sp_fma(W[5], v[2], sp_fma(W[1], v[1], W[3] * v[0]));
  m[2] = sp_fma(W[2], v[2], sp_fma(W[5], v[1], W[4] * v[0]));
  sp_copy3(m, o);
}

/* joint frames of link l given parent pose P and child pose C (kinematics.world_to_joint) */
typedef struct {
  real ap[3], ac[3]; /* anchor world positions on parent / child                                   */
  real rp[3], rc[3]; /* lever arms: the anchor offsets rotated into the world (ap = P.p + rp, ...)  */
  real aprot[4], acrot[4];
  real Xp[3], Yp[3], Zp[3], Xc[3], Yc[3], Zc[3];
  real ang[3];       /* joint-frame Euler angles x, y', z''                                        */
  real ax[3][3];     /* gimbal axes in the world frame: Xp, line of nodes, Zc                      */
} jf_t;

static void joint_frames(const mbd_model_t* m, int l, const xf_t* P, const xf_t* C, jf_t* f) {
  real appos[3] = {m->ap_pos[l][0], m->ap_pos[l][1], m->ap_pos[l][2]};
  real acpos[3] = {m->ac_pos[l][0], m->ac_pos[l][1], m->ac_pos[l][2]};
  real aprot[4] = {m->ap_rot[l][0], m->ap_rot[l][1], m->ap_rot[l][2], m->ap_rot[l][3]};
  real acrot[4] = {m->ac_rot[l][0], m->ac_rot[l][1], m->ac_rot[l][2], m->ac_rot[l][3]};
  sp_rot(appos, P->r, f->rp); sp_add3(P->p, f->rp, f->ap);
  sp_rot(acpos, C->r, f->rc); sp_add3(C->p, f->rc, f->ac);
  sp_qmul(P->r, aprot, f->aprot);
  sp_qmul(C->r, acrot, f->acrot);
  sp_qaxes(f->aprot, f->Xp, f->Yp, f->Zp);
  sp_qaxes(f->acrot, f->Xc, f->Yc, f->Zc);
  /* R_rel = Rx(a) Ry(b) Rz(c), columns = child axes in the parent joint frame.  sin b = Zc.Xp; both
   * (sin a, cos a) ~ (-Zc.Yp, Zc.Zp) and (sin c, cos c) ~ (-Yc.Xp, Xc.Xp) have length cos b, so one
   * reciprocal normalises them and the line of nodes Zc x Xp */
  real sb = sp_clip(sp_dot3(f->Zc, f->Xp), R(-1), R(1));
  real cb2 = sp_fma(-sb, sb, R(1));
  real cb = sp_sqrt_floor(cb2);
  real inv = sp_div(R(1), cb + R(1e-10));
  f->ang[0] = sp_angle_unit(-sp_dot3(f->Zc, f->Yp) * inv, sp_dot3(f->Zc, f->Zp) * inv);
  f->ang[1] = sp_angle_unit(sb, cb);
  f->ang[2] = sp_angle_unit(-sp_dot3(f->Yc, f->Xp) * inv, sp_dot3(f->Xc, f->Xp) * inv);
  sp_copy3(f->Xp, f->ax[0]);
  sp_copy3(f->Zc, f->ax[2]);
  real n[3];
  sp_cross3(f->Zc, f->Xp, n);
  sp_scale3(n, inv, f->ax[1]);
  if ((m->flags & MBD_FLAG_EULER_EXTRINSIC) && m->n_rot[l] >= 2) {
    /* R_rel = Rz(c) Ry(b) Rx(a) (rotations about the FIXED joint-frame axes).  Its transpose is Rx(-a) Ry(-b) Rz(-c): the
     * decomposition above with the roles of parent and child frames exchanged yields (-a, -b, -c), and the relative
     * angular velocity of the child is a' Xc + b' (Zp x Xc)/cos b + c' Zp — the same expressions with p <-> c. */
    real sb2 = sp_clip(sp_dot3(f->Zp, f->Xc), R(-1), R(1));
    real cb2e = sp_fma(-sb2, sb2, R(1));
    real cbe = sp_sqrt_floor(cb2e);
    real inve = sp_div(R(1), cbe + R(1e-10));
    f->ang[0] = -sp_angle_unit(-sp_dot3(f->Zp, f->Yc) * inve, sp_dot3(f->Zp, f->Zc) * inve);
    f->ang[1] = -sp_angle_unit(sb2, cbe);
    f->ang[2] = -sp_angle_unit(-sp_dot3(f->Yp, f->Xc) * inve, sp_dot3(f->Xp, f->Xc) * inve);
    sp_copy3(f->Xc, f->ax[0]);
    sp_copy3(f->Zp, f->ax[2]);
    sp_cross3(f->Zp, f->Xc, n);
    sp_scale3(n, inve, f->ax[1]);
  }
}

/* body-frame inertia tensor (xx yy zz xy xz yz) from the model's inverse: adjugate / determinant of the symmetric 3x3
 * (MBD_FLAG_GYROSCOPIC; the kernels evaluate the same expressions once per env) */
static void inertia_from_inverse(const real b[6], real I[6]) {
  const real xx = b[0], yy = b[1], zz = b[2], xy = b[3], xz = b[4], yz = b[5];
  const real c0 = yy * zz - yz * yz, c1 = xz * yz - xy * zz, c2 = xy * yz - xz * yy;
  const real det = (xx * c0 + xy * c1) + xz * c2;
  const real id = R(1) / det;
  I[0] = c0 * id; I[1] = (xx * zz - xz * xz) * id; I[2] = (xx * yy - xy * xy) * id;
  I[3] = c1 * id; I[4] = c2 * id; I[5] = (xy * xz - xx * yz) * id;
}
/* the gyroscopic angular acceleration -I^-1 (w x I w), evaluated in the body frame of orientation r and rotated back */
static void gyro_accel(const real ib[6], const real r[4], const real w[3], real out[3]) {
  real I[6], wb[3], Lb[3], g[3], ab[3];
  inertia_from_inverse(ib, I);
  sp_irot(w, r, wb);
  Lb[0] = sp_fma(I[4], wb[2], sp_fma(I[3], wb[1], I[0] * wb[0]));
  Lb[1] = sp_fma(I[5], wb[2], sp_fma(I[1], wb[1], I[3] * wb[0]));
  Lb[2] = sp_fma(I[2], wb[2], sp_fma(I[5], wb[1], I[4] * wb[0]));
  sp_cross3(wb, Lb, g);
  ab[0] = -sp_fma(ib[4], g[2], sp_fma(ib[3], g[1], ib[0] * g[0]));
  ab[1] = -sp_fma(ib[5], g[2], sp_fma(ib[1], g[1], ib[3] * g[0]));
  ab[2] = -sp_fma(ib[2], g[2], sp_fma(ib[5], g[1], ib[4] * g[0]));
  sp_rot(ab, r, out);
}

/* one angular positional correction: rotate child by +e, parent by -e, split by angular inverse
 * masses. e: rotation-vector error (world). With n = e/|e|, w = n.I^-1 n and lambda = |e|/(wp+wc) the
 * correction lambda * I^-1 n equals I^-1 e * |e|^2 / (e.I_p^-1 e + e.I_c^-1 e): one division, no
 * square root. accumulates into dth_c / dth_p. */
static void ang_correct(const real e[3], const inert_t* ip, const inert_t* ic, real scale, real dth_p[3],
                        real dth_c[3]) {
  real inp[3], inc[3];
  iinv_apply(ip, e, inp);
  iinv_apply(ic, e, inc);
  real den = sp_dot3(e, inp) + sp_dot3(e, inc);
  real g = sp_div_pos(sp_dot3(e, e), den + R(1e-20)) * scale;
  sp_axpy3(g, inc, dth_c);
  sp_axpy3(-g, inp, dth_p);
}

/* the contact normal is the +z axis of the floor plane: specialised vector helpers (same roundings
 * as the generic ones with n = (0,0,1), minus the multiplications by zero) */
static inline void crossz(const real a[3], real o[3]) { o[0] = a[1]; o[1] = -a[0]; o[2] = R(0); }
/* the same for vectors whose z component is an exact zero (contact normals' moment arms, tangential slip):
 * value-identical to the generic helpers, minus the products with that zero */
static inline real dot_az0(const real a[3], const real b[3]) { return sp_fma(a[0], b[0], a[1] * b[1]); }
static inline void cross_bz0(const real a[3], const real b[3], real o[3]) {
  real x = -(a[2] * b[1]), y = a[2] * b[0], z = sp_fma(a[0], b[1], -(a[1] * b[0]));
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void iinv_apply_z0(const inert_t* in, const real v[3], real o[3]) {
  if (in->world) { sp_set3(o, 0, 0, 0); return; }
  if (in->iso) { sp_set3(o, v[0] * in->ib[0], v[1] * in->ib[0], R(0)); return; }
  if (in->axi) {
    const real kd = (in->axc - in->axa) * sp_fma(in->Z[0], v[0], in->Z[1] * v[1]);
    real m0 = sp_fma(kd, in->Z[0], in->axa * v[0]), m1 = sp_fma(kd, in->Z[1], in->axa * v[1]);
    sp_set3(o, m0, m1, kd * in->Z[2]);
    return;
  }
  const real* W = in->W;
  real m0 = sp_fma(W[3], v[1], W[0] * v[0]), m1 = sp_fma(W[1], v[1], W[3] * v[0]), m2 = sp_fma(W[5], v[1], W[4] * v[0]);
  sp_set3(o, m0, m1, m2);
}

typedef struct {
  int active;
  real pos[3]; /* world contact point */
  real dlam;   /* normal lambda of the positional solve */
} contact_t;

#include "mbd_oracle_planar.h"

static void to_planar(int L, const xf_t* x, const mo_t* xd, pxf_t* px, pmo_t* pd) {
  for (int l = 0; l < L; ++l) {
    px[l].px = x[l].p[0]; px[l].pz = x[l].p[2]; px[l].qw = x[l].r[0]; px[l].qy = x[l].r[2];
    pd[l].vx = xd[l].v[0]; pd[l].vz = xd[l].v[2]; pd[l].om = xd[l].w[1];
  }
}
static void pl_origin3(const mbd_model_t* m, int l, const xf_t* x, real o[3]) { /* x.pos of a planar model's link */
  real tx, tz;
  pl_rot(pl_cs(x[l].r[0], x[l].r[2]), R(m->com[l][0]), R(m->com[l][2]), &tx, &tz);
  o[0] = x[l].p[0] - tx; o[1] = R(0); o[2] = x[l].p[2] - tz;
}
static void from_planar(int L, const pxf_t* px, const pmo_t* pd, xf_t* x, mo_t* xd) {
  for (int l = 0; l < L; ++l) {
    x[l].p[0] = px[l].px; x[l].p[1] = R(0); x[l].p[2] = px[l].pz;
    x[l].r[0] = px[l].qw; x[l].r[1] = R(0); x[l].r[2] = px[l].qy; x[l].r[3] = R(0);
    xd[l].v[0] = pd[l].vx; xd[l].v[1] = R(0); xd[l].v[2] = pd[l].vz;
    xd[l].w[0] = R(0); xd[l].w[1] = pd[l].om; xd[l].w[2] = R(0);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* one physics substep (brax/positional/pipeline.py::step)                                           */
/* ------------------------------------------------------------------------------------------------ */
/* stage dump (tools/compare_golden.py): after each of the six stages the link records [L][13] =
 * (p, r, v, w) as they stand there; stage (1) puts the joint accelerations (before gravity) in the v / w slots. */
static float* g_stage_dump = 0;
#ifdef _OPENMP
#pragma omp threadprivate(g_stage_dump)
#endif
static void dump_stage(int stage, int L, const xf_t* x, const mo_t* xd) {
  if (!g_stage_dump) return;
  float* o = g_stage_dump + (size_t)stage * L * MBD_LINK_STATE;
  for (int l = 0; l < L; ++l) {
    float* a = o + l * MBD_LINK_STATE;
    for (int i = 0; i < 3; ++i) { a[i] = (float)x[l].p[i]; a[7 + i] = (float)xd[l].v[i]; a[10 + i] = (float)xd[l].w[i]; }
    for (int i = 0; i < 4; ++i) a[3 + i] = (float)x[l].r[i];
  }
}

static void substep(const mbd_model_t* m, xf_t* x, mo_t* xd, const real* tau_rot /* [L][3] */,
                    const real* tau_slide /* [L][3] */) {
  const int L = m->n_links;
  if (m->flags & MBD_FLAG_PLANAR) { /* planar models: the planar restatement (mbd_oracle_planar.h) */
    pxf_t px[MBD_MAX_LINKS]; pmo_t pd[MBD_MAX_LINKS];
    to_planar(L, x, xd, px, pd);
    substep_planar(m, px, pd, tau_rot, tau_slide, g_stage_dump);
    from_planar(L, px, pd, x, xd);
    return;
  }
  const real dt = R(m->dt);
  const real inv_dt = R(1) / dt;
  static const xf_t WORLD_X = {{0, 0, 0}, {1, 0, 0, 0}};
  static const mo_t WORLD_XD = {{0, 0, 0}, {0, 0, 0}};
  inert_t in[MBD_MAX_LINKS + 1];
  const int axi = model_axisym(m);
  for (int l = 0; l < L; ++l) {
    in[l].inv_mass = R(m->inv_mass[l]);
    for (int k = 0; k < 6; ++k) in[l].ib[k] = R(m->inv_inertia[l][k]);
    in[l].iso = m->iso_inertia;
    in[l].axi = axi;
    if (axi) axi_setup(&in[l]);
    in[l].world = 0;
    inert_refresh(&in[l], x[l].r); /* stage (1) */
  }
  inert_t* world_in = &in[MBD_MAX_LINKS];
  memset(world_in, 0, sizeof(*world_in));
  world_in->world = 1;

  /* ---- (1) joints.acceleration_update: actuator torque, joint spring/damping, constraint damping */
  real fc_v[MBD_MAX_LINKS][3], fc_w[MBD_MAX_LINKS][3]; /* acceleration of the child from its own joint */
  real fp_v[MBD_MAX_LINKS][3], fp_w[MBD_MAX_LINKS][3]; /* acceleration of the parent from joint l     */
  memset(fc_v, 0, sizeof(fc_v)); memset(fc_w, 0, sizeof(fc_w));
  memset(fp_v, 0, sizeof(fp_v)); memset(fp_w, 0, sizeof(fp_w));
  for (int l = 0; l < L; ++l) {
    if (m->n_rot[l] < 0) continue; /* free joint: no joint forces */
    const int p = m->parent[l];
    const xf_t* P = p >= 0 ? &x[p] : &WORLD_X;
    const mo_t* Pd = p >= 0 ? &xd[p] : &WORLD_XD;
    const inert_t* ip = p >= 0 ? &in[p] : world_in;
    jf_t f;
    joint_frames(m, l, P, &x[l], &f);
    real rc[3], rp[3], t[3], vc[3], vp[3], rel_v[3], rel_w[3];
    sp_copy3(f.rc, rc); sp_copy3(f.rp, rp); /* the lever arms themselves, not (anchor - position) */
    sp_cross3(xd[l].w, rc, t); sp_add3(xd[l].v, t, vc);
    sp_cross3(Pd->w, rp, t); sp_add3(Pd->v, t, vp);
    sp_sub3(vc, vp, rel_v); sp_sub3(xd[l].w, Pd->w, rel_w);
    real T[3] = {0, 0, 0}, F[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
      real qdk = sp_dot3(rel_w, f.ax[k]);
      real fk = sp_fma(-R(m->rot_stiff[l][k]), f.ang[k], sp_fma(-R(m->rot_damp[l][k]), qdk, tau_rot[l * 3 + k]));
      if (k >= m->n_rot[l]) fk = R(0);
      sp_axpy3(fk, f.ax[k], T);
    }
    for (int k = 0; k < m->n_slide[l]; ++k) {
      real s[3], sa[3] = {m->slide_axis[l][k][0], m->slide_axis[l][k][1], m->slide_axis[l][k][2]};
      sp_rot(sa, f.aprot, s);
      real vs = sp_dot3(rel_v, s);
      sp_axpy3(sp_fma(-R(m->slide_damp[l][k]), vs, tau_slide[l * 3 + k]), s, F); /* motor + MJCF joint damping */
      sp_axpy3(-vs, s, rel_v); /* free direction: no constraint damping */
    }
    sp_axpy3(-R(m->ang_damp[l]), rel_w, T);
    sp_axpy3(-R(m->vel_damp[l]), rel_v, F);
    /* child: +F at anchor, +T ; parent: -F at anchor, -T */
    real mom[3], tot[3];
    sp_scale3(F, in[l].inv_mass, fc_v[l]);
    sp_cross3(rc, F, mom); sp_add3(T, mom, tot); iinv_apply(&in[l], tot, fc_w[l]);
    sp_scale3(F, -ip->inv_mass, fp_v[l]);
    sp_cross3(rp, F, mom); sp_add3(T, mom, tot); iinv_apply(ip, tot, t); sp_scale3(t, R(-1), fp_w[l]);
  }
  /* ---- (2) integrator.integrate_xdd: damp, accelerate (+gravity), integrate pose */
  xf_t x_prev[MBD_MAX_LINKS];
  mo_t acc_dump[MBD_MAX_LINKS];
  memset(x_prev, 0, sizeof(x_prev)); memset(acc_dump, 0, sizeof(acc_dump));
  for (int l = 0; l < L; ++l) {
    real av[3], aw[3];
    sp_copy3(fc_v[l], av); sp_copy3(fc_w[l], aw);
    for (int c = l + 1; c < L; ++c)
      if (m->parent[c] == l) { sp_add3(av, fp_v[c], av); sp_add3(aw, fp_w[c], aw); }
    if ((m->flags & MBD_FLAG_GYROSCOPIC) && !m->iso_inertia) { /* from the pose and velocity the substep started with */
      real ib[6], gy[3];
      for (int k = 0; k < 6; ++k) ib[k] = R(m->inv_inertia[l][k]);
      gyro_accel(ib, x[l].r, xd[l].w, gy);
      sp_add3(aw, gy, aw);
    }
    sp_copy3(av, acc_dump[l].v); sp_copy3(aw, acc_dump[l].w);
    for (int i = 0; i < 3; ++i) {
      xd[l].v[i] = sp_fma(av[i] + R(m->gravity[i]), dt, R(m->vel_fac) * xd[l].v[i]);
      xd[l].w[i] = sp_fma(aw[i], dt, R(m->ang_fac) * xd[l].w[i]);
    }
    x_prev[l] = x[l];
    for (int i = 0; i < 3; ++i) x[l].p[i] = sp_fma(xd[l].v[i], dt, x[l].p[i]);
    real th[3];
    sp_scale3(xd[l].w, dt, th);
    sp_qrotvec(x[l].r, th);
  }
  dump_stage(0, L, x_prev, acc_dump); /* (1): poses before the step, accelerations in the velocity slots */
  dump_stage(1, L, x, xd);            /* (2) */
  /* ---- (3) joints.position_update (Jacobi: every joint sees the same post-integration poses) */
  for (int l = 0; l < L; ++l) inert_refresh(&in[l], x[l].r);
  real dc_p[MBD_MAX_LINKS][3], dc_th[MBD_MAX_LINKS][3], dp_p[MBD_MAX_LINKS][3], dp_th[MBD_MAX_LINKS][3];
  memset(dc_p, 0, sizeof(dc_p)); memset(dc_th, 0, sizeof(dc_th));
  memset(dp_p, 0, sizeof(dp_p)); memset(dp_th, 0, sizeof(dp_th));
  for (int l = 0; l < L; ++l) {
    if (m->n_rot[l] < 0) continue;
    const int p = m->parent[l];
    const xf_t* P = p >= 0 ? &x[p] : &WORLD_X;
    const inert_t* ip = p >= 0 ? &in[p] : world_in;
    const inert_t* ic = &in[l];
    jf_t f;
    joint_frames(m, l, P, &x[l], &f);
    /* translational: pull the child anchor onto the parent anchor (free along slide axes) */
    real d[3], rc[3], rp[3];
    sp_sub3(f.ap, f.ac, d);
    for (int k = 0; k < m->n_slide[l]; ++k) {
      real s[3], sa[3] = {m->slide_axis[l][k][0], m->slide_axis[l][k][1], m->slide_axis[l][k][2]};
      sp_rot(sa, f.aprot, s);
      sp_axpy3(-sp_dot3(d, s), s, d);
    }
    sp_copy3(f.rc, rc); sp_copy3(f.rp, rp); /* the lever arms themselves, not (anchor - position) */
    /* with n = d/|d| and lambda = |d|/(wp+wc): P = lambda n = d * |d|^2 / (|d|^2 (1/m_p + 1/m_c) +
     * (rp x d).I_p^-1 (rp x d) + (rc x d).I_c^-1 (rc x d)) — one division, no square root */
    real c2 = sp_dot3(d, d);
    real cp[3], cc[3], icp[3], icc[3];
    sp_cross3(rp, d, cp); sp_cross3(rc, d, cc);
    iinv_apply(ip, cp, icp); iinv_apply(ic, cc, icc);
    real den = sp_fma(ip->inv_mass + ic->inv_mass, c2, sp_dot3(cp, icp) + sp_dot3(cc, icc));
    real g = sp_div_pos(c2, den + R(1e-20)) * R(m->joint_scale_pos);
    real Pimp[3], mom[3], t[3];
    sp_scale3(d, g, Pimp);
    sp_scale3(Pimp, ic->inv_mass, dc_p[l]);
    sp_cross3(rc, Pimp, mom); iinv_apply(ic, mom, dc_th[l]);
    sp_scale3(Pimp, -ip->inv_mass, dp_p[l]);
    sp_cross3(rp, Pimp, mom); iinv_apply(ip, mom, t); sp_scale3(t, R(-1), dp_th[l]);
    /* slide limits: push the child back along the slide axis by the violation (same one-division form) */
    for (int k = 0; k < m->n_slide[l]; ++k) {
      real sx[3], sa[3] = {m->slide_axis[l][k][0], m->slide_axis[l][k][1], m->slide_axis[l][k][2]};
      sp_rot(sa, f.aprot, sx);
      real apc[3];
      sp_sub3(f.ac, f.ap, apc);
      real qs = sp_dot3(apc, sx);
      real lo = R(m->slide_lo[l][k]), hi = R(m->slide_hi[l][k]);
      real viol = qs < lo ? qs - lo : (qs > hi ? qs - hi : R(0));
      real dl[3];
      sp_scale3(sx, -viol, dl);
      real l2 = sp_dot3(dl, dl);
      real lp[3], lc[3], ilp[3], ilc[3];
      sp_cross3(rp, dl, lp); sp_cross3(rc, dl, lc);
      iinv_apply(ip, lp, ilp); iinv_apply(ic, lc, ilc);
      real dens = sp_fma(ip->inv_mass + ic->inv_mass, l2, sp_dot3(lp, ilp) + sp_dot3(lc, ilc));
      real gs = sp_div_pos(l2, dens + R(1e-20)) * R(m->joint_scale_pos);
      real Ps[3], u[3];
      sp_scale3(dl, gs, Ps);
      sp_scale3(Ps, ic->inv_mass, u); sp_add3(dc_p[l], u, dc_p[l]);
      sp_cross3(rc, Ps, mom); iinv_apply(ic, mom, u); sp_add3(dc_th[l], u, dc_th[l]);
      sp_scale3(Ps, -ip->inv_mass, u); sp_add3(dp_p[l], u, dp_p[l]);
      sp_cross3(rp, Ps, mom); iinv_apply(ip, mom, u); sp_scale3(u, R(-1), u); sp_add3(dp_th[l], u, dp_th[l]);
    }
    /* angular alignment by joint type */
    real e[3] = {0, 0, 0};
    const int nr = m->n_rot[l];
    if (nr == 0) {
      real qc[4] = {f.acrot[0], -f.acrot[1], -f.acrot[2], -f.acrot[3]}, qe[4];
      sp_qmul(f.aprot, qc, qe);
      real s = sp_copysign(R(2), qe[0]);
      sp_set3(e, s * qe[1], s * qe[2], s * qe[3]);
    } else {
      /* (2 dofs: the child's Y stays perpendicular to the parent's X; under MBD_FLAG_EULER_EXTRINSIC — R = Ry Rx — the
       * parent's Y stays perpendicular to the child's X instead) */
      const int ext = (m->flags & MBD_FLAG_EULER_EXTRINSIC) && nr == 2;
      const real* A = nr == 1 ? f.Xc : (ext ? f.Yp : f.Xp);
      const real* B = nr == 1 ? f.Xp : (ext ? f.Xc : f.Yc);
      real sc = nr == 1 ? R(1) : (nr == 2 ? sp_dot3(A, B) : R(0));
      real cr[3];
      sp_cross3(A, B, cr);
      sp_scale3(cr, sc, e);
    }
    if (m->iso_inertia) {
      /* isotropic inverse inertia ib*Id: lambda = |e|/(ib_p + ib_c) and the child's share I_c^-1 n lambda is
       * e * ib_c/(ib_p + ib_c) — linear in e, so alignment and limit errors are summed first and applied once */
      real E[3];
      sp_copy3(e, E);
      for (int k = 0; k < 3; ++k) {
        real a = f.ang[k], lo = R(m->rot_lo[l][k]), hi = R(m->rot_hi[l][k]);
        real viol = a < lo ? a - lo : (a > hi ? a - hi : R(0));
        if (k >= nr) viol = R(0);
        sp_axpy3(-viol, f.ax[k], E);
      }
      const real ibp = ip->world ? R(0) : ip->ib[0], ibc = ic->ib[0];
      const real kc = (ibc / (ibp + ibc)) * R(m->joint_scale_ang), kp = (ibp / (ibp + ibc)) * R(m->joint_scale_ang);
      sp_axpy3(kc, E, dc_th[l]);
      sp_axpy3(-kp, E, dp_th[l]);
    } else {
      ang_correct(e, ip, ic, R(m->joint_scale_ang), dp_th[l], dc_th[l]);
      /* joint limits on the Euler angles */
      for (int k = 0; k < 3; ++k) {
        real a = f.ang[k], lo = R(m->rot_lo[l][k]), hi = R(m->rot_hi[l][k]);
        real viol = a < lo ? a - lo : (a > hi ? a - hi : R(0));
        if (k >= nr) viol = R(0);
        real el[3];
        sp_scale3(f.ax[k], -viol, el);
        ang_correct(el, ip, ic, R(m->joint_scale_ang), dp_th[l], dc_th[l]);
      }
    }
  }
  for (int l = 0; l < L; ++l) {
    real dp[3], dth[3];
    sp_copy3(dc_p[l], dp); sp_copy3(dc_th[l], dth);
    for (int c = l + 1; c < L; ++c)
      if (m->parent[c] == l) { sp_add3(dp, dp_p[c], dp); sp_add3(dth, dp_th[c], dth); }
    sp_add3(x[l].p, dp, x[l].p);
    sp_qrotvec_raw(x[l].r, dth); /* renormalised at the end of stage (4) */
  }
  dump_stage(2, L, x, xd); /* (3) */
  /* ---- (4) geometry.contact (sphere-plane) + collisions.resolve_position */
  for (int l = 0; l < L; ++l) inert_refresh(&in[l], x[l].r); /* (not yet renormalised, like the contact points) */
  contact_t con[MBD_MAX_COL];
  real cd_p[MBD_MAX_LINKS][3], cd_th[MBD_MAX_LINKS][3];
  int has_col[MBD_MAX_LINKS];
  memset(cd_p, 0, sizeof(cd_p)); memset(cd_th, 0, sizeof(cd_th)); memset(has_col, 0, sizeof(has_col));
  const real mu = R(m->friction);
  for (int k = 0; k < m->n_col; ++k) {
    const int l = m->col_link[k];
    has_col[l] = 1;
    real cl[3] = {m->col_pos[k][0], m->col_pos[k][1], m->col_pos[k][2]}, t[3], ctr[3];
    sp_rot(cl, x[l].r, t); sp_add3(x[l].p, t, ctr);
    const real rad = R(m->col_radius[k]);
    real pen = rad - ctr[2];
    con[k].active = pen > R(0);
    con[k].dlam = 0;
    if (!con[k].active) continue;
    /* the contact point sits h below the sphere's centre; its lever arm is the rotated collider offset minus that
     * drop (not pos - p), its link-frame coordinates are the collider offset plus the drop rotated back */
    const real h = sp_fma(R(-0.5), pen, rad);
    sp_set3(con[k].pos, ctr[0], ctr[1], ctr[2] - h);
    real rc[3] = {t[0], t[1], t[2] - h}, cn[3], icn[3];
    crossz(rc, cn); iinv_apply_z0(&in[l], cn, icn);
    real w = in[l].inv_mass + dot_az0(cn, icn);
    real dlam = sp_div_pos(pen, w) * R(m->collide_scale);
    con[k].dlam = dlam;
    real Pimp[3] = {0, 0, dlam}, mom[3];
    /* static friction: undo the tangential motion of the contact point over this substep if the
     * required tangential lambda stays inside the friction cone. With t = dx_t/|dx_t|:
     * lambda_t t = -dx_t * |dx_t|^2 / (|dx_t|^2/m + (rc x dx_t).I^-1 (rc x dx_t)); the cone test
     * |lambda_t| < mu lambda_n is done on squares — one division, no square root */
    real rl[3], pprev[3], dx[3];
    sp_irot_z(-h, x[l].r, rl); sp_add3(cl, rl, rl);
    sp_rot(rl, x_prev[l].r, t); sp_add3(x_prev[l].p, t, pprev);
    sp_sub3(con[k].pos, pprev, dx);
    dx[2] = R(0);
    real ct2 = sp_fma(dx[0], dx[0], dx[1] * dx[1]);
    real cnt[3], icnt[3];
    cross_bz0(rc, dx, cnt); iinv_apply(&in[l], cnt, icnt);
    real dent = sp_fma(in[l].inv_mass, ct2, sp_dot3(cnt, icnt));
    real gt = sp_div_pos(ct2, dent + R(1e-20));
    real lim = mu * dlam;
    if ((ct2 * gt) * gt < lim * lim) { Pimp[0] = (-gt) * dx[0]; Pimp[1] = (-gt) * dx[1]; }
    sp_axpy3(in[l].inv_mass, Pimp, cd_p[l]);
    sp_cross3(rc, Pimp, mom); iinv_apply(&in[l], mom, t); sp_add3(cd_th[l], t, cd_th[l]);
  }
  int n_act[MBD_MAX_LINKS];
  memset(n_act, 0, sizeof(n_act));
  for (int k = 0; k < m->n_col; ++k) n_act[m->col_link[k]] += con[k].active ? 1 : 0;
  for (int l = 0; l < L; ++l) { /* all links: zero corrections where there is no collider */
    if ((m->flags & MBD_FLAG_CONTACT_AVG) && n_act[l] >= 2) { /* the average over the link's active contacts */
      const real inv_n = R(1) / (real)n_act[l];
      sp_scale3(cd_p[l], inv_n, cd_p[l]); sp_scale3(cd_th[l], inv_n, cd_th[l]);
    }
    sp_add3(x[l].p, cd_p[l], x[l].p);
    sp_qrotvec(x[l].r, cd_th[l]);
  }
  dump_stage(3, L, x, xd); /* (4) */
  /* ---- (5) integrator.project_xd: velocities from the position change */
  mo_t xd_prev[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) {
    xd_prev[l] = xd[l];
    for (int i = 0; i < 3; ++i) xd[l].v[i] = (x[l].p[i] - x_prev[l].p[i]) * inv_dt;
    real qc[4] = {x_prev[l].r[0], -x_prev[l].r[1], -x_prev[l].r[2], -x_prev[l].r[3]}, dq[4];
    sp_qmul(x[l].r, qc, dq);
    real s = sp_copysign(R(2), dq[0]) * inv_dt;
    for (int i = 0; i < 3; ++i) xd[l].w[i] = dq[1 + i] * s;
  }
  dump_stage(4, L, x, xd); /* (5) */
  /* ---- (6) collisions.resolve_velocity: restitution + dynamic friction at active contacts */
  for (int l = 0; l < L; ++l) inert_refresh(&in[l], x[l].r);
  /* Jacobi per link (default since round 5): every contact of a link computes its impulse from the velocities stage (5)
   * left and the changes are added in collider order — what a vmap over contacts followed by a per-link segment sum (Brax's
   * code structure: hopper.py:18,40 and envs/__init__.py:30-31 reach it through PipelineEnv) computes.
   * MBD_FLAG_CONTACT6_GAUSS_SEIDEL keeps the sequential form of rounds 1-4. */
  const int jacobi6 = (m->flags & MBD_FLAG_CONTACT6_GAUSS_SEIDEL) == 0;
  mo_t xd6[MBD_MAX_LINKS]; /* the velocities stage (5) left: what every contact of a link sees (Jacobi) */
  for (int l = 0; l < L; ++l) xd6[l] = xd[l];
  for (int k = 0; k < m->n_col; ++k) {
    if (!con[k].active) continue;
    const int l = m->col_link[k];
    const mo_t* see = jacobi6 ? &xd6[l] : &xd[l]; /* (Gauss-Seidel: what the link's previous contacts left) */
    real rc[3], t[3], vpt[3], vprev[3];
    sp_sub3(con[k].pos, x[l].p, rc);
    sp_cross3(see->w, rc, t); sp_add3(see->v, t, vpt);
    sp_cross3(xd_prev[l].w, rc, t); sp_add3(xd_prev[l].v, t, vprev);
    real vn = vpt[2], vn_prev = vprev[2];
    real vt[3] = {vpt[0], vpt[1], R(0)};
    real vtn = sp_sqrt_floor(sp_fma(vt[0], vt[0], vt[1] * vt[1]));
    real inv = sp_div(R(1), vtn + R(1e-10));
    real dir[3], cn[3], icn[3], cdv[3], icd[3];
    sp_set3(dir, vt[0] * inv, vt[1] * inv, R(0));
    crossz(rc, cn); iinv_apply_z0(&in[l], cn, icn);
    cross_bz0(rc, dir, cdv); iinv_apply(&in[l], cdv, icd);
    real wn = in[l].inv_mass + dot_az0(cn, icn), wt = in[l].inv_mass + sp_dot3(cdv, icd);
    real rest = -R(m->elasticity) * vn_prev;
    /* restitution (Mueller et al. 2020, eq. 34, written for a normal that points OUT of the floor, +z): an approaching
     * contact has vn_prev < 0, and the velocity after the solve is max(-e vn_prev, 0).  (Until round 3 this read
     * min(...): with this normal that term is never positive and the elasticity did nothing — found by
     * tests/test_oracle_invariants.py::test_restitution; every built-in model has e = 0, for which both forms give 0.
     * MBD_FLAG_RESTITUTION_MIN keeps the literal min(): which sign convention Brax's normal has is unpinned.) */
    real dvn = ((m->flags & MBD_FLAG_RESTITUTION_MIN) ? sp_min(rest, R(0)) : sp_max(rest, R(0))) - vn;
    real jt_max = (mu * con[k].dlam) * inv_dt; /* friction bound mu * lambda_n / h: an impulse (default) or a velocity */
    real dvt = sp_min((m->flags & MBD_FLAG_FRICTION_VEL_BOUND) ? jt_max : jt_max * wt, vtn);
    real jn = sp_div(dvn, wn), jt = -sp_div_pos(dvt, wt);
    real Pimp[3] = {dir[0] * jt, dir[1] * jt, jn};
    sp_axpy3(in[l].inv_mass, Pimp, xd[l].v); /* (Jacobi: the changes are added in collider order, onto the running value) */
    real mom[3];
    sp_cross3(rc, Pimp, mom); iinv_apply(&in[l], mom, t); sp_add3(xd[l].w, t, xd[l].w);
  }
  if (jacobi6 && (m->flags & MBD_FLAG_CONTACT_AVG))
    for (int l = 0; l < L; ++l)
      if (n_act[l] >= 2) { /* the average of the link's velocity changes: v6 + (v - v6) / n */
        const real inv_n = R(1) / (real)n_act[l];
        for (int i = 0; i < 3; ++i) {
          xd[l].v[i] = sp_fma(xd[l].v[i] - xd6[l].v[i], inv_n, xd6[l].v[i]);
          xd[l].w[i] = sp_fma(xd[l].w[i] - xd6[l].w[i], inv_n, xd6[l].w[i]);
        }
      }
  dump_stage(5, L, x, xd); /* (6) */
}

/* ---- state <-> float buffers -------------------------------------------------------------------- */
static void load_state(const float* s, int L, xf_t* x, mo_t* xd) {
  for (int l = 0; l < L; ++l) {
    const float* a = s + l * MBD_LINK_STATE;
    for (int i = 0; i < 3; ++i) { x[l].p[i] = a[i]; xd[l].v[i] = a[7 + i]; xd[l].w[i] = a[10 + i]; }
    for (int i = 0; i < 4; ++i) x[l].r[i] = a[3 + i];
  }
}
static void store_state(float* s, int L, const xf_t* x, const mo_t* xd) {
  for (int l = 0; l < L; ++l) {
    float* a = s + l * MBD_LINK_STATE;
    for (int i = 0; i < 3; ++i) { a[i] = (float)x[l].p[i]; a[7 + i] = (float)xd[l].v[i]; a[10 + i] = (float)xd[l].w[i]; }
    for (int i = 0; i < 4; ++i) a[3 + i] = (float)x[l].r[i];
  }
}
/* world position of the link-frame origin: x.pos = x_i.pos - rotate(com, rot)  (com.to_world) */
static void link_origin(const mbd_model_t* m, int l, const xf_t* x, real o[3]) {
  real c[3] = {m->com[l][0], m->com[l][1], m->com[l][2]}, t[3];
  sp_rot(c, x[l].r, t);
  sp_sub3(x[l].p, t, o);
}
static void link_origin_vel(const mbd_model_t* m, int l, const xf_t* x, const mo_t* xd, real o[3]) {
  real c[3] = {m->com[l][0], m->com[l][1], m->com[l][2]}, t[3], u[3];
  sp_rot(c, x[l].r, t);
  sp_cross3(xd[l].w, t, u);
  sp_sub3(xd[l].v, u, o);
}

/* the env wrappers' _get_reward as functions of the root link's frame origin before (o0, with its velocity v0) and after
 * (o1) the control step — every reward kind but cartpole's, which reads joint coordinates (env_step).  Exported as
 * orc_reward so that tests can hold these expressions to the reference's own _get_reward code, executed
 * (tools/make_ref_golden.py, tests/test_ref_golden.py). */
static real reward_origin(const mbd_model_t* m, const real o0[3], const real v0[3], const real o1[3], const float* action) {
  switch (m->reward_kind) {
    case MBD_REW_HUMANOIDRUN: /* humanoidrun.py:46-51 */
      return o1[0] * R(1) - sp_clip(sp_abs(o1[2] - R(1.3)), R(-1), R(1)) * R(1) - sp_abs(o1[1]) * R(0.1);
    case MBD_REW_HOPPER: /* hopper.py:57-65 (z0 = 1.0) / walker2d.py:57-62 (z0 = 1.1) */
      return o1[0] - sp_clip(sp_abs(o1[2] - R(m->reward_params[0])), R(-1), R(1)) * R(m->reward_params[1]);
    case MBD_REW_HALFCHEETAH: { /* brax half_cheetah: forward velocity - 0.1*|a|^2 */
      real ctrl = 0;
      for (int a = 0; a < m->n_act; ++a) ctrl += R(action[a]) * R(action[a]);
      real dtc = R(m->dt) * (real)m->n_frames;
      return R(m->reward_params[0]) * ((o1[0] - o0[0]) / dtc) - R(m->reward_params[1]) * ctrl;
    }
    case MBD_REW_HUMANOIDTRACK: /* humanoidtrack.py:87-96: from the INCOMING state */
      return R(1) + (-sp_abs(v0[0] - R(1.6)) - sp_abs(o0[2] - R(1.3)) - sp_abs(o0[1]) * R(0.1));
    case MBD_REW_ANT: { /* brax ant: forward velocity + healthy bonus - ctrl cost */
      real ctrl = 0;
      for (int a = 0; a < m->n_act; ++a) ctrl += R(action[a]) * R(action[a]);
      real dtc = R(m->dt) * (real)m->n_frames;
      /* reward_params[5] != 0: terminate_when_unhealthy (brax's default) makes the healthy term unconditional */
      real healthy = (m->reward_params[5] != 0.0f || (o1[2] >= R(m->reward_params[2]) && o1[2] <= R(m->reward_params[3])))
                         ? R(m->reward_params[4]) : R(0);
      return (R(m->reward_params[0]) * ((o1[0] - o0[0]) / dtc) + healthy) - R(m->reward_params[1]) * ctrl;
    }
    case MBD_REW_HUMANOIDSTANDUP: /* humanoidstandup.py:50-56 */
      return R(1.5) - sp_clip(sp_abs(o1[2] - R(1.3)), R(-2), R(1)) - sp_abs(o1[0]) * R(0.1) - sp_abs(o1[1]) * R(0.1);
    default: return 0;
  }
}

/* env.step for one environment: n_frames substeps with the action held, then the reward
 * (PipelineEnv.pipeline_step + the env wrapper's _get_reward). returns the reward. */
static real env_step(const mbd_model_t* m, xf_t* x, mo_t* xd, const float* action) {
  const int L = m->n_links;
  real tau_rot[MBD_MAX_LINKS * 3], tau_slide[MBD_MAX_LINKS * 3];
  memset(tau_rot, 0, sizeof(tau_rot)); memset(tau_slide, 0, sizeof(tau_slide));
  /* actuator.to_tau: clip to ctrlrange, times gear, scatter */
  for (int a = 0; a < m->n_act; ++a) {
    real u = sp_clip(R(action[a]), R(m->act_lo[a]), R(m->act_hi[a])) * R(m->act_gear[a]);
    int l = m->act_link[a], s = m->act_slot[a];
    if (s < 3) tau_rot[l * 3 + s] += u; else tau_slide[l * 3 + s - 3] += u;
  }
  real o0[3], v0[3];
  const int planar = (m->flags & MBD_FLAG_PLANAR) != 0;
  if (planar) { pl_origin3(m, 0, x, o0); v0[0] = v0[1] = v0[2] = R(0); }
  else { link_origin(m, 0, x, o0); link_origin_vel(m, 0, x, xd, v0); }
  for (int f = 0; f < m->n_frames; ++f) substep(m, x, xd, tau_rot, tau_slide);
  real o1[3];
  if (planar) pl_origin3(m, 0, x, o1); else link_origin(m, 0, x, o1);
  (void)L;
  switch (m->reward_kind) {
    case MBD_REW_CARTPOLE: { /* cartpole.py:45: cos(q[1]) - |qd[0]|: hinge angle of link 1, slide velocity of link 0 */
      if (planar) {
        plink_t K[MBD_MAX_LINKS];
        pl_setup(m, K);
        real wr, yr, sn, cs, r0x, r0z;
        pl_rel(x[0].r[0], x[0].r[2], x[1].r[0], x[1].r[2], &wr, &yr);
        sp_sincos(K[1].sg * pl_angle(wr, yr), &sn, &cs);
        pl_rot(pl_cs(x[0].r[0], x[0].r[2]), K[0].acx, K[0].acz, &r0x, &r0z);
        const real vcx = sp_fma(xd[0].w[1], r0z, xd[0].v[0]), vcz = sp_fma(-xd[0].w[1], r0x, xd[0].v[2]);
        return cs - sp_abs(sp_fma(vcx, K[0].sx[0], vcz * K[0].sz[0]));
      }
      static const xf_t WORLD_X = {{0, 0, 0}, {1, 0, 0, 0}};
      jf_t f0, f1;
      joint_frames(m, 0, &WORLD_X, &x[0], &f0);
      joint_frames(m, 1, &x[0], &x[1], &f1);
      real sx[3], sa[3] = {m->slide_axis[0][0][0], m->slide_axis[0][0][1], m->slide_axis[0][0][2]}, rc0[3], t[3], vc[3];
      sp_rot(sa, f0.aprot, sx);
      sp_sub3(f0.ac, x[0].p, rc0);
      sp_cross3(xd[0].w, rc0, t); sp_add3(xd[0].v, t, vc); /* the parent is the static world */
      real sn, cs;
      sp_sincos(f1.ang[0], &sn, &cs);
      return cs - sp_abs(sp_dot3(vc, sx));
    }
    default: return reward_origin(m, o0, v0, o1, action);
  }
}

ORC_API float orc_reward(const mbd_model_t* m, const float* o0, const float* v0, const float* o1, const float* action) {
  real a[3] = {R(o0[0]), R(o0[1]), R(o0[2])}, b[3] = {R(v0[0]), R(v0[1]), R(v0[2])}, c[3] = {R(o1[0]), R(o1[1]), R(o1[2])};
  return (float)reward_origin(m, a, b, c, action);
}

ORC_API float orc_env_step(const mbd_model_t* m, const float* state_in, const float* action,
                           float* state_out) {
  xf_t x[MBD_MAX_LINKS]; mo_t xd[MBD_MAX_LINKS];
  load_state(state_in, m->n_links, x, xd);
  real r = env_step(m, x, xd, action);
  store_state(state_out, m->n_links, x, xd);
  return (float)r;
}

/* a single physics substep with explicit generalized forces — used by the invariants tests */
ORC_API void orc_substep(const mbd_model_t* m, const float* state_in, const float* action,
                         float* state_out) {
  xf_t x[MBD_MAX_LINKS]; mo_t xd[MBD_MAX_LINKS];
  load_state(state_in, m->n_links, x, xd);
  real tau_rot[MBD_MAX_LINKS * 3], tau_slide[MBD_MAX_LINKS * 3];
  memset(tau_rot, 0, sizeof(tau_rot)); memset(tau_slide, 0, sizeof(tau_slide));
  for (int a = 0; a < m->n_act; ++a) {
    real u = sp_clip(R(action[a]), R(m->act_lo[a]), R(m->act_hi[a])) * R(m->act_gear[a]);
    int l = m->act_link[a], s = m->act_slot[a];
    if (s < 3) tau_rot[l * 3 + s] += u; else tau_slide[l * 3 + s - 3] += u;
  }
  substep(m, x, xd, tau_rot, tau_slide);
  store_state(state_out, m->n_links, x, xd);
}

/* one substep with the state after each of the six stages: stages [6][L][13] (see dump_stage) */
ORC_API void orc_substep_stages(const mbd_model_t* m, const float* state_in, const float* action, float* state_out,
                                float* stages) {
  g_stage_dump = stages;
  orc_substep(m, state_in, action, state_out);
  g_stage_dump = 0;
}

/* jax.vmap(rollout_us, in_axes=(None, 0)) (mbd_planner.py:109; utils.py:14-20): the reward and the
 * tracked link positions AFTER every control step. Single-threaded (cores = 1) unless built with
 * -fopenmp, in which case candidates are distributed over threads. */
ORC_API void orc_rollout(const mbd_model_t* m, const float* state0, const float* us /* [B][H][Nu] */,
                         int B, int H, float* rewss /* [B][H] */, float* xpos /* [B][H][K][3] or NULL */,
                         float* state_final /* [B][state] or NULL */) {
  const int L = m->n_links, Nu = m->n_act, K = m->n_track;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int b = 0; b < B; ++b) {
    xf_t x[MBD_MAX_LINKS]; mo_t xd[MBD_MAX_LINKS];
    load_state(state0, L, x, xd);
    for (int t = 0; t < H; ++t) {
      real r = env_step(m, x, xd, &us[((size_t)b * H + t) * Nu]);
      rewss[(size_t)b * H + t] = (float)r;
      if (xpos)
        for (int k = 0; k < K; ++k) {
          real o[3];
          link_origin(m, m->track_link[k], x, o);
          for (int i = 0; i < 3; ++i) xpos[(((size_t)b * H + t) * K + k) * 3 + i] = (float)o[i];
        }
    }
    if (state_final) store_state(state_final + (size_t)b * L * MBD_LINK_STATE, L, x, xd);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* reset: kinematics.forward + com.from_world  (pipeline_init; humanoidrun.py:29, hopper.py:30)      */
/* q [n_q], qd [n_qd] -> state [L][13]                                                               */
/* ------------------------------------------------------------------------------------------------ */
ORC_API void orc_forward(const mbd_model_t* m, const float* q, const float* qd, float* state) {
  const int L = m->n_links;
  xf_t X[MBD_MAX_LINKS];   /* link frames */
  real V[MBD_MAX_LINKS][3], W[MBD_MAX_LINKS][3]; /* velocity of the link origin, angular velocity */
  xf_t x[MBD_MAX_LINKS]; mo_t xd[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) {
    const int p = m->parent[l];
    xf_t P = {{0, 0, 0}, {1, 0, 0, 0}};
    real Pv[3] = {0, 0, 0}, Pw[3] = {0, 0, 0};
    if (p >= 0) { P = X[p]; memcpy(Pv, V[p], sizeof(Pv)); memcpy(Pw, W[p], sizeof(Pw)); }
    const float* ql = q + m->q_idx[l];
    const float* qdl = qd + m->qd_idx[l];
    if (m->n_rot[l] < 0) { /* free joint: q = pos, quat ; qd = vel, ang */
      for (int i = 0; i < 3; ++i) { X[l].p[i] = ql[i]; V[l][i] = qdl[i]; W[l][i] = qdl[3 + i]; }
      for (int i = 0; i < 4; ++i) X[l].r[i] = ql[3 + i];
      if (!(m->flags & MBD_FLAG_RESET_QUAT_RAW)) sp_qnormalize(X[l].r); /* (brax may leave it un-normalised) */
    } else {
      real jpos[3] = {0, 0, 0}, jrot[4] = {1, 0, 0, 0}, sv[3] = {0, 0, 0}, wrel[3] = {0, 0, 0};
      const int ns = m->n_slide[l], nr = m->n_rot[l];
      for (int k = 0; k < ns; ++k) {
        real a[3] = {m->slide_axis_body[l][k][0], m->slide_axis_body[l][k][1], m->slide_axis_body[l][k][2]};
        sp_axpy3(R(ql[k]), a, jpos);
        sp_axpy3(R(qdl[k]), a, sv);
      }
      const int ext = (m->flags & MBD_FLAG_EULER_EXTRINSIC) && nr >= 2;
      for (int k = 0; k < nr; ++k) {
        real a[3] = {m->rot_axis[l][k][0], m->rot_axis[l][k][1], m->rot_axis[l][k][2]}, ac[3];
        real h = R(0.5) * R(ql[ns + k]);
        real s, c;
        sp_sincos(h, &s, &c);
        real qk[4] = {c, s * a[0], s * a[1], s * a[2]}, t[4];
        if (ext) { /* about the FIXED axes: R <- R_k R, and what was accumulated so far turns with R_k */
          sp_rot(wrel, qk, ac); sp_copy3(ac, wrel);
          sp_axpy3(R(qdl[ns + k]), a, wrel);
          sp_qmul(qk, jrot, t);
        } else {
          sp_rot(a, jrot, ac);
          sp_axpy3(R(qdl[ns + k]), ac, wrel);
          sp_qmul(jrot, qk, t);
        }
        memcpy(jrot, t, sizeof(t));
      }
      real jp[3] = {m->joint_pos[l][0], m->joint_pos[l][1], m->joint_pos[l][2]}, t[3];
      sp_rot(jp, jrot, t);
      for (int i = 0; i < 3; ++i) jpos[i] += jp[i] - t[i];
      real lp[3] = {m->link_pos[l][0], m->link_pos[l][1], m->link_pos[l][2]};
      real lr[4] = {m->link_rot[l][0], m->link_rot[l][1], m->link_rot[l][2], m->link_rot[l][3]};
      real lpos[3], lrot[4];
      sp_rot(jpos, lr, t); sp_add3(lp, t, lpos);
      sp_qmul(lr, jrot, lrot);
      sp_rot(lpos, P.r, t); sp_add3(P.p, t, X[l].p);
      sp_qmul(P.r, lrot, X[l].r);
      real u[3], A[3], rA[3], vA[3];
      sp_rot(wrel, lr, t); sp_rot(t, P.r, u); sp_add3(Pw, u, W[l]);
      sp_rot(jp, X[l].r, rA); sp_add3(X[l].p, rA, A);
      sp_sub3(A, P.p, t); sp_cross3(Pw, t, u); sp_add3(Pv, u, vA);
      sp_rot(sv, lr, t); sp_rot(t, P.r, u); sp_add3(vA, u, vA);
      sp_cross3(W[l], rA, u); sp_sub3(vA, u, V[l]);
    }
    /* com.from_world */
    real c[3] = {m->com[l][0], m->com[l][1], m->com[l][2]}, rc[3], u[3];
    sp_rot(c, X[l].r, rc);
    sp_add3(X[l].p, rc, x[l].p);
    memcpy(x[l].r, X[l].r, sizeof(x[l].r));
    sp_cross3(W[l], rc, u); sp_add3(V[l], u, xd[l].v);
    memcpy(xd[l].w, W[l], sizeof(xd[l].w));
  }
  store_state(state, L, x, xd);
}

/* world positions of all link origins [L][3] (x.pos) and joint angles, for tests and viewers */
ORC_API void orc_link_positions(const mbd_model_t* m, const float* state, float* xpos) {
  xf_t x[MBD_MAX_LINKS]; mo_t xd[MBD_MAX_LINKS];
  load_state(state, m->n_links, x, xd);
  for (int l = 0; l < m->n_links; ++l) {
    real o[3];
    link_origin(m, l, x, o);
    for (int i = 0; i < 3; ++i) xpos[l * 3 + i] = (float)o[i];
  }
}
ORC_API void orc_joint_angles(const mbd_model_t* m, const float* state, float* ang /* [L][3] */) {
  static const xf_t WORLD_X = {{0, 0, 0}, {1, 0, 0, 0}};
  xf_t x[MBD_MAX_LINKS]; mo_t xd[MBD_MAX_LINKS];
  load_state(state, m->n_links, x, xd);
  for (int l = 0; l < m->n_links; ++l) {
    for (int i = 0; i < 3; ++i) ang[l * 3 + i] = 0;
    if (m->n_rot[l] < 0) continue;
    jf_t f;
    joint_frames(m, l, m->parent[l] >= 0 ? &x[m->parent[l]] : &WORLD_X, &x[l], &f);
    for (int i = 0; i < 3; ++i) ang[l * 3 + i] = (float)f.ang[i];
  }
}
ORC_API int orc_real_bytes(void) { return (int)sizeof(real); }
ORC_API int orc_model_bytes(void) { return (int)sizeof(mbd_model_t); }
