/*
 * mbd_oracle_planar.h — TEST INFRASTRUCTURE ONLY (CPU oracle), included by mbd_oracle_physics.c.
 *
 * The same six stages of brax/positional/pipeline.py::step (see mbd_oracle_physics.c) for PLANAR models
 * (mbd_model_t.flags & MBD_FLAG_PLANAR: every hinge turns about the world y axis, every slide moves in the x-z
 * plane, every offset lies in that plane — hopper, walker2d, halfcheetah, cartpole), restated on the in-plane
 * coordinates only: position (x, z), orientation as the half-angle pair (w, y) of the unit quaternion
 * (w, 0, y, 0), velocity (x, z), angular velocity about y.  It is what the 3-D arithmetic computes when the
 * out-of-plane components are exact zeros — which they are NOT in the 3-D float arithmetic (the joint frames carry
 * ap_rot = (0.7071…, 0, 0, ±0.7071…), so a rollout leaks 1e-5…1e-3 out of the plane; tests/test_oracle_physics.py
 * measures how far the two restatements drift apart) — so this is a SPECIFICATION of its own for these models, shared
 * with the kernels' planar instantiation, selected by a flag of the MODEL.  Differences that are not just dropped
 * zeros: the hinge angle comes from the relative half-angle pair instead of direction cosines, angular corrections
 * take the linear form a scalar inertia allows (like the isotropic 3-D path), the tangential contact speed is |v_t|.
 *
 * A link = a lane: parent data fetched, children's contributions added in increasing child index.
 */
#ifndef ORC_PLANAR_H
#define ORC_PLANAR_H

typedef struct { real px, pz, qw, qy; } pxf_t;   /* pose   */
typedef struct { real vx, vz, om; } pmo_t;       /* motion */
typedef struct { real c, s; } pcs_t;             /* cos / sin of the FULL angle about +y */

static inline pcs_t pl_cs(real w, real y) {
  real y2 = y + y;
  pcs_t o;
  o.c = sp_fma(-y2, y, R(1));
  o.s = y2 * w;
  return o;
}
/* R(theta) (x, z): x' = c x + s z, z' = c z - s x */
static inline void pl_rot(pcs_t a, real x, real z, real* ox, real* oz) {
  *ox = sp_fma(a.s, z, a.c * x);
  *oz = sp_fma(-a.s, x, a.c * z);
}
/* (r x F)_y = r.z F.x - r.x F.z */
static inline real pl_cross(real rx, real rz, real fx, real fz) { return sp_fma(rz, fx, -(rx * fz)); }
/* relative half-angle pair conj(P) (x) C and the joint angle about +y it encodes */
static inline void pl_rel(real Pw, real Py, real Cw, real Cy, real* wr, real* yr) {
  *wr = sp_fma(Pw, Cw, Py * Cy);
  *yr = sp_fma(Pw, Cy, -(Py * Cw));
}
static inline real pl_angle(real wr, real yr) {
  real sn = (wr + wr) * yr;
  real cn = sp_fma(-(yr + yr), yr, R(1));
  return sp_angle_unit(sn, cn);
}
static inline void pl_qupdate(real* w, real* y, real dth, int normalize) {
  real h = R(0.5) * dth;
  real nw = sp_fma(-h, *y, *w), ny = sp_fma(h, *w, *y);
  if (normalize) {
    real n2 = sp_fma(nw, nw, ny * ny);
    real e = n2 - R(1);
    real inv = sp_fma(sp_fma(sp_fma(sp_fma(R(0.2734375), e, R(-0.3125)), e, R(0.375)), e, R(-0.5)), e, R(1));
    if (sp_abs(e) > R(0.05)) inv = R(1) / sp_sqrt(n2);
    nw = nw * inv; ny = ny * inv;
  }
  *w = nw; *y = ny;
}

typedef struct {
  real im, iy;           /* inverse mass, inverse inertia about y */
  real apx, apz, acx, acz;
  real sg;               /* the hinge turns about sg * (+y) */
  real sx[2], sz[2];     /* world-frame slide axes (zero when the joint has no such slot) */
} plink_t;

static void pl_setup(const mbd_model_t* m, plink_t* k) {
  for (int l = 0; l < m->n_links; ++l) {
    k[l].im = R(m->inv_mass[l]);
    k[l].iy = R(m->inv_inertia[l][1]);
    k[l].apx = R(m->ap_pos[l][0]); k[l].apz = R(m->ap_pos[l][2]);
    k[l].acx = R(m->ac_pos[l][0]); k[l].acz = R(m->ac_pos[l][2]);
    k[l].sg = (m->ap_rot[l][0] * m->ap_rot[l][3] < 0.0f) ? R(-1) : R(1);
    real aprot[4] = {m->ap_rot[l][0], m->ap_rot[l][1], m->ap_rot[l][2], m->ap_rot[l][3]};
    for (int j = 0; j < 2; ++j) {
      real sa[3] = {m->slide_axis[l][j][0], m->slide_axis[l][j][1], m->slide_axis[l][j][2]}, s[3] = {0, 0, 0};
      if (j < m->n_slide[l]) sp_rot(sa, aprot, s); /* the parent of a slide joint is the world: a constant */
      k[l].sx[j] = s[0]; k[l].sz[j] = s[2];
    }
  }
}

static void substep_planar(const mbd_model_t* m, pxf_t* x, pmo_t* xd, const real* tau_rot, const real* tau_slide,
                           float* stage_dump) {
  const int L = m->n_links;
  const real dt = R(m->dt), inv_dt = R(1) / dt, two_inv_dt = R(2) * inv_dt;
  static const pxf_t WORLD_X = {0, 0, 1, 0};
  static const pmo_t WORLD_XD = {0, 0, 0};
  plink_t K[MBD_MAX_LINKS];
  pl_setup(m, K);
  const real js_pos = R(m->joint_scale_pos), js_ang = R(m->joint_scale_ang);
  int slide_limits = 0;
  for (int l = 0; l < L; ++l)
    for (int j = 0; j < m->n_slide[l]; ++j)
      if (m->slide_lo[l][j] > -1e8f || m->slide_hi[l][j] < 1e8f) slide_limits = 1;
#define PL_DUMP(stage)                                                                              \
  if (stage_dump) {                                                                                 \
    float* o_ = stage_dump + (size_t)(stage) * L * MBD_LINK_STATE;                                  \
    for (int l_ = 0; l_ < L; ++l_) {                                                                \
      float* a_ = o_ + l_ * MBD_LINK_STATE;                                                         \
      memset(a_, 0, sizeof(float) * MBD_LINK_STATE);                                                \
      a_[0] = (float)x[l_].px; a_[2] = (float)x[l_].pz; a_[3] = (float)x[l_].qw; a_[5] = (float)x[l_].qy; \
      a_[7] = (float)xd[l_].vx; a_[9] = (float)xd[l_].vz; a_[11] = (float)xd[l_].om;                \
    }                                                                                               \
  }

  /* ---- (1) joints.acceleration_update ------------------------------------------------------------- */
  real fcvx[MBD_MAX_LINKS], fcvz[MBD_MAX_LINKS], fcw[MBD_MAX_LINKS];
  real fpvx[MBD_MAX_LINKS], fpvz[MBD_MAX_LINKS], fpw[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) {
    const int p = m->parent[l];
    const pxf_t* P = p >= 0 ? &x[p] : &WORLD_X;
    const pmo_t* Pd = p >= 0 ? &xd[p] : &WORLD_XD;
    const real im_p = p >= 0 ? K[p].im : R(0), iy_p = p >= 0 ? K[p].iy : R(0);
    const pcs_t cP = pl_cs(P->qw, P->qy), cC = pl_cs(x[l].qw, x[l].qy);
    real rpx, rpz, rcx, rcz;
    pl_rot(cP, K[l].apx, K[l].apz, &rpx, &rpz);
    pl_rot(cC, K[l].acx, K[l].acz, &rcx, &rcz);
    const real vpx = sp_fma(Pd->om, rpz, Pd->vx), vpz = sp_fma(-Pd->om, rpx, Pd->vz);
    const real vcx = sp_fma(xd[l].om, rcz, xd[l].vx), vcz = sp_fma(-xd[l].om, rcx, xd[l].vz);
    real rvx = vcx - vpx, rvz = vcz - vpz;
    const real rw = xd[l].om - Pd->om;
    real wr, yr;
    pl_rel(P->qw, P->qy, x[l].qw, x[l].qy, &wr, &yr);
    const real ang = K[l].sg * pl_angle(wr, yr);
    const real qd = K[l].sg * rw;
    real fk = sp_fma(-R(m->rot_stiff[l][0]), ang, sp_fma(-R(m->rot_damp[l][0]), qd, tau_rot[l * 3]));
    if (m->n_rot[l] < 1) fk = R(0);
    const real Ty = sp_fma(-R(m->ang_damp[l]), rw, fk * K[l].sg);
    real Fx = R(0), Fz = R(0);
    for (int j = 0; j < 2; ++j) {
      const real sx = K[l].sx[j], sz = K[l].sz[j];
      const real vs = sp_fma(rvx, sx, rvz * sz);
      const real fs = j < m->n_slide[l] ? sp_fma(-R(m->slide_damp[l][j]), vs, tau_slide[l * 3 + j]) : R(0);
      Fx = sp_fma(fs, sx, Fx); Fz = sp_fma(fs, sz, Fz);
      rvx = sp_fma(-vs, sx, rvx); rvz = sp_fma(-vs, sz, rvz);
    }
    Fx = sp_fma(-R(m->vel_damp[l]), rvx, Fx); Fz = sp_fma(-R(m->vel_damp[l]), rvz, Fz);
    fcvx[l] = Fx * K[l].im; fcvz[l] = Fz * K[l].im;
    fcw[l] = (Ty + pl_cross(rcx, rcz, Fx, Fz)) * K[l].iy;
    fpvx[l] = Fx * (-im_p); fpvz[l] = Fz * (-im_p);
    fpw[l] = -((Ty + pl_cross(rpx, rpz, Fx, Fz)) * iy_p);
  }
  /* ---- (2) integrator.integrate_xdd ----------------------------------------------------------------- */
  pxf_t x_prev[MBD_MAX_LINKS];
  pmo_t acc[MBD_MAX_LINKS];
  memset(x_prev, 0, sizeof(x_prev)); memset(acc, 0, sizeof(acc));
  for (int l = 0; l < L; ++l) {
    real ax = fcvx[l], az = fcvz[l], aw = fcw[l];
    for (int c = l + 1; c < L; ++c)
      if (m->parent[c] == l) { ax = ax + fpvx[c]; az = az + fpvz[c]; aw = aw + fpw[c]; }
    acc[l].vx = ax; acc[l].vz = az; acc[l].om = aw;
    xd[l].vx = sp_fma(ax + R(m->gravity[0]), dt, R(m->vel_fac) * xd[l].vx);
    xd[l].vz = sp_fma(az + R(m->gravity[2]), dt, R(m->vel_fac) * xd[l].vz);
    xd[l].om = sp_fma(aw, dt, R(m->ang_fac) * xd[l].om);
    x_prev[l] = x[l];
    x[l].px = sp_fma(xd[l].vx, dt, x[l].px);
    x[l].pz = sp_fma(xd[l].vz, dt, x[l].pz);
    pl_qupdate(&x[l].qw, &x[l].qy, xd[l].om * dt, 1);
  }
  if (stage_dump) { /* (1): poses before the step, accelerations in the velocity slots */
    float* o_ = stage_dump;
    for (int l = 0; l < L; ++l) {
      float* a = o_ + l * MBD_LINK_STATE;
      memset(a, 0, sizeof(float) * MBD_LINK_STATE);
      a[0] = (float)x_prev[l].px; a[2] = (float)x_prev[l].pz; a[3] = (float)x_prev[l].qw; a[5] = (float)x_prev[l].qy;
      a[7] = (float)acc[l].vx; a[9] = (float)acc[l].vz; a[11] = (float)acc[l].om;
    }
  }
  PL_DUMP(1);
  /* ---- (3) joints.position_update (Jacobi) ---------------------------------------------------------- */
  real dcx[MBD_MAX_LINKS], dcz[MBD_MAX_LINKS], dcth[MBD_MAX_LINKS], dpx[MBD_MAX_LINKS], dpz[MBD_MAX_LINKS], dpth[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) {
    const int p = m->parent[l];
    const pxf_t* P = p >= 0 ? &x[p] : &WORLD_X;
    const real im_p = p >= 0 ? K[p].im : R(0), iy_p = p >= 0 ? K[p].iy : R(0);
    const real im_c = K[l].im, iy_c = K[l].iy, invm_sum = im_p + im_c;
    const pcs_t cP = pl_cs(P->qw, P->qy), cC = pl_cs(x[l].qw, x[l].qy);
    real rpx, rpz, rcx, rcz;
    pl_rot(cP, K[l].apx, K[l].apz, &rpx, &rpz);
    pl_rot(cC, K[l].acx, K[l].acz, &rcx, &rcz);
    const real apx = P->px + rpx, apz = P->pz + rpz, acx = x[l].px + rcx, acz = x[l].pz + rcz;
    real dx = apx - acx, dz = apz - acz;
    for (int j = 0; j < 2; ++j) {
      const real sx = K[l].sx[j], sz = K[l].sz[j];
      const real cf = -sp_fma(dx, sx, dz * sz);
      dx = sp_fma(cf, sx, dx); dz = sp_fma(cf, sz, dz);
    }
    const real c2 = sp_fma(dx, dx, dz * dz);
    const real crp = pl_cross(rpx, rpz, dx, dz), crc = pl_cross(rcx, rcz, dx, dz);
    const real wq = crp * (iy_p * crp) + crc * (iy_c * crc);
    const real den = sp_fma(invm_sum, c2, wq) + R(1e-20);
    const real g = sp_div_pos(c2, den) * js_pos;
    const real Px = dx * g, Pz = dz * g;
    dcx[l] = Px * im_c; dcz[l] = Pz * im_c;
    dpx[l] = Px * (-im_p); dpz[l] = Pz * (-im_p);
    dcth[l] = pl_cross(rcx, rcz, Px, Pz) * iy_c;
    dpth[l] = -(pl_cross(rpx, rpz, Px, Pz) * iy_p);
    if (slide_limits) {
      const real ex = acx - apx, ez = acz - apz;
      for (int j = 0; j < 2; ++j) {
        const real sx = K[l].sx[j], sz = K[l].sz[j];
        const real qs = sp_fma(ex, sx, ez * sz);
        const real lo = j < m->n_slide[l] ? R(m->slide_lo[l][j]) : R(-1e9), hi = j < m->n_slide[l] ? R(m->slide_hi[l][j]) : R(1e9);
        const real viol = qs - sp_clip(qs, lo, hi);
        const real lx = sx * (-viol), lz = sz * (-viol);
        const real l2 = sp_fma(lx, lx, lz * lz);
        const real lp = pl_cross(rpx, rpz, lx, lz), lc = pl_cross(rcx, rcz, lx, lz);
        const real dens = sp_fma(invm_sum, l2, lp * (iy_p * lp) + lc * (iy_c * lc));
        const real gs = sp_div_pos(l2, dens + R(1e-20)) * js_pos;
        const real Sx = lx * gs, Sz = lz * gs;
        dcx[l] = sp_fma(Sx, im_c, dcx[l]); dcz[l] = sp_fma(Sz, im_c, dcz[l]);
        dpx[l] = sp_fma(Sx, -im_p, dpx[l]); dpz[l] = sp_fma(Sz, -im_p, dpz[l]);
        dcth[l] = sp_fma(pl_cross(rcx, rcz, Sx, Sz), iy_c, dcth[l]);
        dpth[l] = sp_fma(pl_cross(rpx, rpz, Sx, Sz), -iy_p, dpth[l]);
      }
    }
    /* angular: orientation lock of hinge-less joints, Euler-angle limit of hinges — one scalar error about y,
     * split by the (scalar) angular inverse masses: linear in the error, no division */
    real wr, yr;
    pl_rel(P->qw, P->qy, x[l].qw, x[l].qy, &wr, &yr);
    real E;
    if (m->n_rot[l] < 1) {
      E = sp_copysign(R(2), wr) * (-yr);
    } else {
      const real ang = K[l].sg * pl_angle(wr, yr);
      const real viol = ang - sp_clip(ang, R(m->rot_lo[l][0]), R(m->rot_hi[l][0]));
      E = (-viol) * K[l].sg;
    }
    const real kc = (iy_c / (iy_p + iy_c)) * js_ang, kp = (iy_p / (iy_p + iy_c)) * js_ang;
    dcth[l] = sp_fma(kc, E, dcth[l]);
    dpth[l] = sp_fma(-kp, E, dpth[l]);
  }
  for (int l = 0; l < L; ++l) {
    real ax = dcx[l], az = dcz[l], ath = dcth[l];
    for (int c = l + 1; c < L; ++c)
      if (m->parent[c] == l) { ax = ax + dpx[c]; az = az + dpz[c]; ath = ath + dpth[c]; }
    x[l].px = x[l].px + ax; x[l].pz = x[l].pz + az;
    pl_qupdate(&x[l].qw, &x[l].qy, ath, 0); /* renormalised at the end of stage (4) */
  }
  PL_DUMP(2);
  /* ---- (4) sphere-plane contacts + collisions.resolve_position -------------------------------------- */
  int act[MBD_MAX_COL];
  real cposx[MBD_MAX_COL], cposz[MBD_MAX_COL], cdlam[MBD_MAX_COL];
  real cdx[MBD_MAX_LINKS], cdz[MBD_MAX_LINKS], cdth[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) cdx[l] = cdz[l] = cdth[l] = R(0);
  const real mu = R(m->friction);
  for (int k = 0; k < m->n_col; ++k) {
    const int l = m->col_link[k];
    const real cx = R(m->col_pos[k][0]), cz = R(m->col_pos[k][2]), rad = R(m->col_radius[k]);
    const pcs_t a = pl_cs(x[l].qw, x[l].qy), ap = pl_cs(x_prev[l].qw, x_prev[l].qy);
    real offx, offz;
    pl_rot(a, cx, cz, &offx, &offz);
    const real ctrx = x[l].px + offx, ctrz = x[l].pz + offz;
    const real pen = rad - ctrz;
    act[k] = pen > R(0);
    const real h = sp_fma(R(-0.5), pen, rad);
    cposx[k] = ctrx; cposz[k] = ctrz - h;
    const real rcx = offx, rcz = offz - h;
    const real icn = rcx * K[l].iy;
    const real wn = sp_fma(icn, rcx, K[l].im);
    const real d = -h;
    const real rlx = sp_fma(-a.s, d, cx), rlz = sp_fma(a.c, d, cz); /* collider offset + the drop rotated back */
    const real pprevx = x_prev[l].px + sp_fma(ap.s, rlz, ap.c * rlx);
    const real ddx = cposx[k] - pprevx;
    /* static friction: the tangent is the x axis, so |d|^2 / (d.W d) of the 3-D form is 1 / (im + rcz^2 iy) */
    const real wt = sp_fma(rcz, rcz * K[l].iy, K[l].im);
    const real dlam = sp_div_pos(pen, wn) * R(m->collide_scale);
    const real sx = sp_div_pos(R(1), wt) * ddx;
    cdlam[k] = dlam;
    const real lim = mu * dlam;
    const real Pix = (sx * sx < lim * lim) ? -sx : R(0), Piz = dlam;
    const real dth = pl_cross(rcx, rcz, Pix, Piz) * K[l].iy;
    if (act[k]) { /* (onto exact zeros for the first active collider of a link) */
      cdx[l] = sp_fma(K[l].im, Pix, cdx[l]); cdz[l] = sp_fma(K[l].im, Piz, cdz[l]); cdth[l] = cdth[l] + dth;
    }
  }
  int n_act[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) n_act[l] = 0;
  for (int k = 0; k < m->n_col; ++k) n_act[m->col_link[k]] += act[k] ? 1 : 0;
  for (int l = 0; l < L; ++l) {
    if ((m->flags & MBD_FLAG_CONTACT_AVG) && n_act[l] >= 2) { /* the average over the link's active contacts */
      const real inv_n = R(1) / (real)n_act[l];
      cdx[l] = cdx[l] * inv_n; cdz[l] = cdz[l] * inv_n; cdth[l] = cdth[l] * inv_n;
    }
    x[l].px = x[l].px + cdx[l]; x[l].pz = x[l].pz + cdz[l];
    pl_qupdate(&x[l].qw, &x[l].qy, cdth[l], 1);
  }
  PL_DUMP(3);
  /* ---- (5) integrator.project_xd ---------------------------------------------------------------------- */
  pmo_t xd_old[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) {
    xd_old[l] = xd[l];
    xd[l].vx = (x[l].px - x_prev[l].px) * inv_dt;
    xd[l].vz = (x[l].pz - x_prev[l].pz) * inv_dt;
    const real dqw = sp_fma(x[l].qw, x_prev[l].qw, x[l].qy * x_prev[l].qy);
    const real dqy = sp_fma(x[l].qy, x_prev[l].qw, -(x[l].qw * x_prev[l].qy));
    xd[l].om = dqy * sp_copysign(two_inv_dt, dqw);
  }
  PL_DUMP(4);
  /* ---- (6) collisions.resolve_velocity (Jacobi per link: every contact of a link from the velocities stage (5) left,
   * the changes added in collider order; MBD_FLAG_CONTACT6_GAUSS_SEIDEL: one after the other, the form of rounds 1-4) - */
  const int jacobi6 = (m->flags & MBD_FLAG_CONTACT6_GAUSS_SEIDEL) == 0;
  pmo_t xd6[MBD_MAX_LINKS];
  for (int l = 0; l < L; ++l) xd6[l] = xd[l];
  for (int k = 0; k < m->n_col; ++k) {
    if (!act[k]) continue;
    const int l = m->col_link[k];
    const pmo_t* see = jacobi6 ? &xd6[l] : &xd[l];
    const real rcx = cposx[k] - x[l].px, rcz = cposz[k] - x[l].pz;
    const real vptx = sp_fma(see->om, rcz, see->vx), vptz = sp_fma(-see->om, rcx, see->vz);
    const real vn_prev = sp_fma(-xd_old[l].om, rcx, xd_old[l].vz);
    /* in the plane the slip direction is a SIGN (the 3-D form's vt / (|vt| + 1e-10) is +-1 up to 1e-10 / |vt|):
     * no division, the tangential lever arm is rcz itself */
    const real vtn = sp_abs(vptx);
    const real icn = rcx * K[l].iy;
    const real wn = sp_fma(icn, rcx, K[l].im);
    const real wt = sp_fma(rcz, rcz * K[l].iy, K[l].im);
    const real rest = -R(m->elasticity) * vn_prev;
    const real dvn = ((m->flags & MBD_FLAG_RESTITUTION_MIN) ? sp_min(rest, R(0)) : sp_max(rest, R(0))) - vptz;
    const real jt_max = (mu * cdlam[k]) * inv_dt;
    const real dvt = sp_min((m->flags & MBD_FLAG_FRICTION_VEL_BOUND) ? jt_max : jt_max * wt, vtn);
    const real jn = sp_div(dvn, wn);
    const real Pix = -sp_copysign(sp_div_pos(dvt, wt), vptx), Piz = jn; /* friction opposes the slip */
    xd[l].vx = sp_fma(K[l].im, Pix, xd[l].vx);
    xd[l].vz = sp_fma(K[l].im, Piz, xd[l].vz);
    xd[l].om = xd[l].om + pl_cross(rcx, rcz, Pix, Piz) * K[l].iy;
  }
  if (jacobi6 && (m->flags & MBD_FLAG_CONTACT_AVG))
    for (int l = 0; l < L; ++l)
      if (n_act[l] >= 2) { /* the average of the link's velocity changes: v6 + (v - v6) / n */
        const real inv_n = R(1) / (real)n_act[l];
        xd[l].vx = sp_fma(xd[l].vx - xd6[l].vx, inv_n, xd6[l].vx);
        xd[l].vz = sp_fma(xd[l].vz - xd6[l].vz, inv_n, xd6[l].vz);
        xd[l].om = sp_fma(xd[l].om - xd6[l].om, inv_n, xd6[l].om);
      }
  PL_DUMP(5);
#undef PL_DUMP
}

#endif /* ORC_PLANAR_H */
