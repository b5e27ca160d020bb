// probe_quad.hip — would a COMPONENT-PER-LANE layout (4 lanes per link, one candidate per wavefront, 1024 wavefronts at
// N=1024 instead of 256) cut the rollout kernel's issue slots per substep?  (round-2 verdict item 3; kill criterion:
// less than 1.3x fewer issue slots.)
//
// Times the largest block of a substep — joint_frames, evaluated twice per substep, a third of the kernel — in both
// layouts under the same numerical contract (explicit fma order, exact division / square root sequences, the same
// polynomial angle kernel), one wavefront per SIMD on every SIMD:
//   S  the product's layout: a link per lane, (parent, child) pairs in v_pk_* registers   (mbd_kernels.h as shipped)
//   Q  a component per lane: x, y, z, w of a link in the 4 lanes of a DPP quad; cross products, quaternion products
//      and frame assembly through quad_perm DPP (v_mul_f32_dpp / v_fmac_f32_dpp / v_mov_b32_dpp), the three Euler
//      angles of a joint evaluated in lanes x, y, z by ONE polynomial evaluation
// and checks that Q reproduces S's anchors and Euler angles.  Prints ns per evaluation and the ratio.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../model-based-diffusion_amd/csrc/mbd_kernels.h"

using namespace mbd;

// ---------------------------------------------------------------------------------------------------------------
// Q: quad-layout primitives.  A "qv" is one float per lane; lanes 4s..4s+3 of slot s hold (x, y, z, w).
// ---------------------------------------------------------------------------------------------------------------
#define QPERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL>
__device__ __forceinline__ float qp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
constexpr int P_YZX = QPERM(1, 2, 0, 3), P_ZXY = QPERM(2, 0, 1, 3);
constexpr int P_B0 = QPERM(0, 0, 0, 0), P_B1 = QPERM(1, 1, 1, 1), P_B2 = QPERM(2, 2, 2, 2), P_B3 = QPERM(3, 3, 3, 3);
#define DPPM " row_mask:0xf bank_mask:0xf"
// cross(a, b), rounded exactly like mbd::cross: lane x holds fma(a.y, b.z, -(a.z b.y)) etc.  3 VALU.
__device__ __forceinline__ float cross_q(float a, float b) {
  float t, r;
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %0, -%2, %3 quad_perm:[1,2,0,3]" DPPM "\n\t"
      "v_fmac_f32_dpp %0, %3, %2 quad_perm:[1,2,0,3]" DPPM "\n\t"
      "s_nop 1\n\t"
      "v_mov_b32_dpp %1, %0 quad_perm:[1,2,0,3]" DPPM
      : "=&v"(t), "=&v"(r)
      : "v"(a), "v"(b));
  return r;
}
// two independent cross products interleaved: the second hides the first one's DPP hazard
__device__ __forceinline__ void cross2_q(float a1, float b1, float a2, float b2, float& r1, float& r2) {
  float t1, t2;
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %0, -%4, %5 quad_perm:[1,2,0,3]" DPPM "\n\t"
      "v_mul_f32_dpp %1, -%6, %7 quad_perm:[1,2,0,3]" DPPM "\n\t"
      "v_fmac_f32_dpp %0, %5, %4 quad_perm:[1,2,0,3]" DPPM "\n\t"
      "v_fmac_f32_dpp %1, %7, %6 quad_perm:[1,2,0,3]" DPPM "\n\t"
      "s_nop 0\n\t"
      "v_mov_b32_dpp %2, %0 quad_perm:[1,2,0,3]" DPPM "\n\t"
      "v_mov_b32_dpp %3, %1 quad_perm:[1,2,0,3]" DPPM
      : "=&v"(t1), "=&v"(t2), "=&v"(r1), "=&v"(r2)
      : "v"(a1), "v"(b1), "v"(a2), "v"(b2));
}
// rot(v, q) for a (parent, child) pair at once: t = 2 (u x v); v + w t + u x t   (mbd::rot's roundings)
__device__ __forceinline__ void rot2_q(float v1, float q1, float v2, float q2, float& o1, float& o2) {
  float t1, t2, c1, c2;
  cross2_q(q1, v1, q2, v2, t1, t2);
  t1 = t1 + t1;
  t2 = t2 + t2;
  cross2_q(q1, t1, q2, t2, c1, c2);
  o1 = ffma(qp<P_B3>(q1), t1, v1) + c1;
  o2 = ffma(qp<P_B3>(q2), t2, v2) + c2;
}
// a (x) b with b a per-link CONSTANT given as its four sign-permuted images (host-prepared):
//   b0 = (bx, by, bz, bw)  bX = (bw, -bz, by, -bx)  bY = (bz, bw, -bx, -by)  bZ = (-by, bx, bw, -bz)
// rounded exactly like mbd::qmul (aw b first, then the x, y, z terms).  4 VALU.
struct QConst { float b0, bX, bY, bZ; };
__device__ __forceinline__ float qmulc_q(float a, const QConst& b) {
  float s = qp<P_B3>(a) * b.b0;
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0]" DPPM "\n\t"
      "v_fmac_f32_dpp %0, %1, %3 quad_perm:[1,1,1,1]" DPPM "\n\t"
      "v_fmac_f32_dpp %0, %1, %4 quad_perm:[2,2,2,2]" DPPM
      : "+v"(s)
      : "v"(a), "v"(b.bX), "v"(b.bY), "v"(b.bZ));
  return s;
}
// columns X, Y, Z of the rotation matrix of q (mbd::qaxes' roundings): 17 VALU
__device__ __forceinline__ void qaxes_q(float q, bool l0, bool l1, float& X, float& Y, float& Z) {
  const float q2 = q + q;
  const float sq = q * q2;                       // (xx, yy, zz, ww)
  const float cyc = q * qp<P_YZX>(q2);           // (x y2, y z2, z x2) = (xy, yz, xz)
  const float pw = qp<P_B3>(q) * q2;             // (wx, wy, wz)
  const float S = qp<P_YZX>(sq) + qp<P_ZXY>(sq); // (yy+zz, zz+xx, xx+yy)
  const float D = 1.0f - S;                      // (X.x, Y.y, Z.z)
  const float pwr = qp<P_ZXY>(pw);               // (wz, wx, wy)
  const float Up = cyc + pwr;                    // (xy+wz, yz+wx, xz+wy) = (X.y, Y.z, Z.x)
  const float Um = cyc - pwr;                    // (xy-wz, yz-wx, xz-wy) = (Y.x, Z.y, X.z)
  const float Upr = qp<P_ZXY>(Up);               // (Z.x, X.y, Y.z)
  X = l0 ? D : (l1 ? Upr : Um);                  // (X.x, X.y, X.z)
  Y = l0 ? Um : (l1 ? D : Upr);                  // (Y.x, Y.y, Y.z)
  Z = l0 ? Upr : (l1 ? Um : D);                  // (Z.x, Z.y, Z.z)
}
// dot(a, b) = fma(a.x, b.x, fma(a.y, b.y, a.z b.z)) — the exact chain, result in lane x of the quad.  5 VALU.
__device__ __forceinline__ float dot_q(float a, float b) {
  float p = a * b;
  float u = ffma(a, b, qp<P_YZX>(p));
  return ffma(a, b, qp<P_YZX>(u));
}

struct QFrames { float ap, ac, Xp, Xc, Yc, Zc, ax1, ang; };  // ang: (ang0, ang1, ang2) in lanes x, y, z
struct QJointConst { float ap_pos, ac_pos; QConst ap_rot, ac_rot; };

__device__ __forceinline__ QFrames joint_frames_q(const QJointConst& jc, float Pp, float Pr, float Cp, float Cr,
                                                  bool l0, bool l1, bool l2) {
  QFrames f;
  float armp, armc;
  rot2_q(jc.ap_pos, Pr, jc.ac_pos, Cr, armp, armc);
  f.ap = Pp + armp;
  f.ac = Cp + armc;
  const float aprot = qmulc_q(Pr, jc.ap_rot), acrot = qmulc_q(Cr, jc.ac_rot);
  float AX, AY, AZ, CX, CY, CZ;
  qaxes_q(aprot, l0, l1, AX, AY, AZ);
  qaxes_q(acrot, l0, l1, CX, CY, CZ);
  f.Xp = AX; f.Xc = CX; f.Yc = CY; f.Zc = CZ;
  // the five direction cosines (lane x of each), then ONE scalar chain replicated in the quad
  const float d_zx = qp<P_B0>(dot_q(CZ, AX)), d_zy = qp<P_B0>(dot_q(CZ, AY)), d_zz = qp<P_B0>(dot_q(CZ, AZ));
  const float d_yx = qp<P_B0>(dot_q(CY, AX)), d_xx = qp<P_B0>(dot_q(CX, AX));
  const float sb = fclip(d_zx, -1.0f, 1.0f);
  const float cb = sqrt_floor(ffma(-sb, sb, 1.0f));
  const float inv = div_(1.0f, cb + 1e-10f);
  // lane x: angle(-zy inv, zz inv); lane y: angle(sb, cb); lane z: angle(-yx inv, xx inv) — one evaluation
  const float s_in = l0 ? -d_zy * inv : (l1 ? sb : -d_yx * inv);
  const float c_in = l0 ? d_zz * inv : (l1 ? cb : d_xx * inv);
  f.ang = angle_unit(s_in, c_in);
  f.ax1 = cross_q(CZ, AX) * inv;
  (void)l2;
  return f;
}

// ---------------------------------------------------------------------------------------------------------------
// harness
// ---------------------------------------------------------------------------------------------------------------
struct ProbeIn {   // per link slot (16 slots): parent pose, own pose, joint constants
  float Pp[3], Pr[4], Cp[3], Cr[4], ap_pos[3], ac_pos[3], ap_rot[4], ac_rot[4];
};

__global__ __launch_bounds__(256) void kernel_S(const ProbeIn* in, float* out, int iters) {
  const int slot = threadIdx.x & 15;
  const ProbeIn I = in[slot];
  JointConst jc;
  jc.ap_pos = mk3(I.ap_pos[0], I.ap_pos[1], I.ap_pos[2]); jc.ac_pos = mk3(I.ac_pos[0], I.ac_pos[1], I.ac_pos[2]);
  jc.ap_rot = q4{I.ap_rot[0], I.ap_rot[1], I.ap_rot[2], I.ap_rot[3]};
  jc.ac_rot = q4{I.ac_rot[0], I.ac_rot[1], I.ac_rot[2], I.ac_rot[3]};
  v3 Pp = mk3(I.Pp[0], I.Pp[1], I.Pp[2]), Cp = mk3(I.Cp[0], I.Cp[1], I.Cp[2]);
  q4 Pr = q4{I.Pr[0], I.Pr[1], I.Pr[2], I.Pr[3]}, Cr = q4{I.Cr[0], I.Cr[1], I.Cr[2], I.Cr[3]};
  JointFrames f{};
  for (int i = 0; i < iters; ++i) {
    f = joint_frames(jc, Pp, Pr, Cp, Cr, true);
    // feed a little of everything back so that nothing is hoisted or dropped (1e-9: the pose barely moves)
    const float k = 1e-9f;
    Cp = axpy(k, add(add(f.Xp, f.Xc), add(f.Yc, add(f.Zc, f.ax1))), Cp);
    Pp = axpy(k, add(f.ap, f.ac), Pp);
    Cr.w = ffma(k, f.ang0 + f.ang1 + f.ang2, Cr.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < 16) {
    float* o = out + slot * 12;
    o[0] = f.ap.x; o[1] = f.ap.y; o[2] = f.ap.z; o[3] = f.ac.x; o[4] = f.ac.y; o[5] = f.ac.z;
    o[6] = f.ang0; o[7] = f.ang1; o[8] = f.ang2; o[9] = f.ax1.x; o[10] = f.ax1.y; o[11] = f.ax1.z;
  }
}

__global__ __launch_bounds__(256) void kernel_Q(const ProbeIn* in, float* out, int iters) {
  const int lane = threadIdx.x & 63, slot = lane >> 2, c = lane & 3;
  const ProbeIn I = in[slot];
  const bool l0 = c == 0, l1 = c == 1, l2 = c == 2;
  auto v3l = [&](const float* v) { return c < 3 ? v[c] : 0.0f; };
  auto q4l = [&](const float* q) { return c < 3 ? q[c + 1] : q[0]; };  // (x, y, z, w) lane order from (w, x, y, z)
  auto qc = [&](const float* q) {  // sign-permuted images of a constant quaternion (w, x, y, z)
    const float bw = q[0], bx = q[1], by = q[2], bz = q[3];
    const float b0[4] = {bx, by, bz, bw}, bX[4] = {bw, -bz, by, -bx}, bY[4] = {bz, bw, -bx, -by}, bZ[4] = {-by, bx, bw, -bz};
    return QConst{b0[c], bX[c], bY[c], bZ[c]};
  };
  QJointConst jc{v3l(I.ap_pos), v3l(I.ac_pos), qc(I.ap_rot), qc(I.ac_rot)};
  float Pp = v3l(I.Pp), Cp = v3l(I.Cp), Pr = q4l(I.Pr), Cr = q4l(I.Cr);
  QFrames f{};
  for (int i = 0; i < iters; ++i) {
    f = joint_frames_q(jc, Pp, Pr, Cp, Cr, l0, l1, l2);
    const float k = 1e-9f;
    Cp = ffma(k, ((f.Xp + f.Xc) + (f.Yc + (f.Zc + f.ax1))), Cp);
    Pp = ffma(k, f.ap + f.ac, Pp);
    const float asum = qp<P_B0>(f.ang) + qp<P_B1>(f.ang) + qp<P_B2>(f.ang);
    Cr = c == 3 ? ffma(k, asum, Cr) : Cr;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float* o = out + slot * 12;
    if (c < 3) { o[c] = f.ap; o[3 + c] = f.ac; o[6 + c] = f.ang; o[9 + c] = f.ax1; }
  }
}

static void unit(float* q, unsigned& s) {
  float n = 0;
  for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; q[k] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; n += q[k] * q[k]; }
  n = std::sqrt(n);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}

int main() {
  std::vector<ProbeIn> h(16);
  unsigned s = 12345u;
  for (auto& I : h) {
    auto r3 = [&](float* v, float sc) { for (int k = 0; k < 3; ++k) { s = s * 1664525u + 1013904223u; v[k] = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * sc; } };
    r3(I.Pp, 2.0f); r3(I.Cp, 2.0f); r3(I.ap_pos, 0.6f); r3(I.ac_pos, 0.6f);
    unit(I.Pr, s); unit(I.ap_rot, s); unit(I.ac_rot, s);
    // child orientation close to the parent's joint frame (a moderately flexed joint)
    unit(I.Cr, s);
    for (int k = 0; k < 4; ++k) I.Cr[k] = I.Pr[k] + 0.3f * I.Cr[k];
    float n = std::sqrt(I.Cr[0] * I.Cr[0] + I.Cr[1] * I.Cr[1] + I.Cr[2] * I.Cr[2] + I.Cr[3] * I.Cr[3]);
    for (int k = 0; k < 4; ++k) I.Cr[k] /= n;
  }
  ProbeIn* d_in; float *d_s, *d_q;
  (void)hipMalloc(&d_in, sizeof(ProbeIn) * 16); (void)hipMalloc(&d_s, 16 * 12 * 4); (void)hipMalloc(&d_q, 16 * 12 * 4);
  (void)hipMemcpy(d_in, h.data(), sizeof(ProbeIn) * 16, hipMemcpyHostToDevice);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 20000;
  float ms_s = 0, ms_q = 0;
  for (int rep = 0; rep < 2; ++rep) {  // first pass warms up
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kernel_S, dim3(256), dim3(256), 0, 0, d_in, d_s, rep ? iters : 1);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms_s, a, b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kernel_Q, dim3(256), dim3(256), 0, 0, d_in, d_q, rep ? iters : 1);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms_q, a, b);
    if (rep == 0) {  // one evaluation from identical inputs: Q must reproduce S
      std::vector<float> os(16 * 12), oq(16 * 12);
      (void)hipMemcpy(os.data(), d_s, os.size() * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(oq.data(), d_q, oq.size() * 4, hipMemcpyDeviceToHost);
      double md = 0; int bit = 0;
      for (size_t k = 0; k < os.size(); ++k) { md = std::fmax(md, std::fabs((double)os[k] - oq[k])); bit += os[k] == oq[k]; }
      printf("one evaluation, 16 joints x (anchors, Euler angles, line of nodes): max |S - Q| = %.3g, %d of %zu values bit-identical\n",
             md, bit, os.size());
    }
  }
  const double ns_s = ms_s * 1e6 / iters, ns_q = ms_q * 1e6 / iters;
  printf("S (link per lane, packed pairs; 4 candidates per wave): %.1f ns per joint_frames = %.0f cycles @2.4 GHz\n", ns_s, ns_s * 2.4);
  printf("Q (component per lane, DPP quads; 1 candidate per wave): %.1f ns per joint_frames = %.0f cycles @2.4 GHz\n", ns_q, ns_q * 2.4);
  printf("issue-slot ratio S/Q = %.2f (kill criterion of the verdict: < 1.3)\n", ns_s / ns_q);
  return 0;
}
