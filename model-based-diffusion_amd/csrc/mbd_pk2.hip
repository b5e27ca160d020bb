// mbd_pk2.hip — translation unit of the two-candidates-per-lane rollouts (mbd_pk2.h).  Its own file because it is built
// with its own scheduler strategy (-mllvm -amdgpu-sched-strategy=iterative-ilp, __graft_entry__.build): a v_pk_*_f32
// result cannot be read by the very next instruction (the compiler inserts an s_nop), and these kernels are almost
// entirely dependent packed chains — the default strategy leaves 97 such wait states per substep, the iterative one 27.
// The one-candidate-per-lane kernels (mbd_env.hip) measured slower under that strategy and keep the default.
#define MBD_SHARED_ONLY 1
#include "mbd_pk2.h"
#include "mbd_launch.h"

namespace mbd {

// fam: 0 humanoid-shaped, 1 ant-shaped (mbd_pk2.h); out = (MAXCOL, RK, NFR) of the instantiation that serves the model
bool pk2_instantiation(int fam, int max_col, int rk, int nfr, int out[3]) {
  if (fam == 1) {
    if (max_col > 2 || rk != MBD_REW_ANT) return false;
    out[0] = 2; out[1] = -1; out[2] = 0;
    if (nfr == 10) { out[1] = rk; out[2] = 10; }
    return true;
  }
  if (max_col > 5) return false;
  if (rk != MBD_REW_HUMANOIDRUN && rk != MBD_REW_HUMANOIDTRACK && rk != MBD_REW_HUMANOIDSTANDUP) return false;
  out[0] = max_col <= 1 ? 1 : 5;
  out[1] = -1;
  out[2] = 0;
  if (out[0] == 1 && rk == MBD_REW_HUMANOIDRUN && nfr == 7) { out[1] = rk; out[2] = 7; }
  if (out[0] == 1 && rk == MBD_REW_HUMANOIDTRACK && nfr == 5) { out[1] = rk; out[2] = 5; }
  if (out[0] == 5 && rk == MBD_REW_HUMANOIDSTANDUP && nfr == 7) { out[1] = rk; out[2] = 7; }
  return true;
}

hipError_t launch_rollout_pk2(int fam, int maxcol, int rk, int nfr, int wpe, int device, dim3 grid, dim3 block, size_t lds,
                              hipStream_t stream, const RolloutParams& P) {
#define PK(...) return launch_rollout_kernel(rollout_pk2_kernel<__VA_ARGS__>, device, grid, block, lds, stream, P)
  if (fam == 1) {  // ant (the reference's default env_name: mbd_planner.py:28, run_mbd.py:14)
    // (capped at 256 registers for two wavefronts per SIMD it is SLOWER here — 78 scratch accesses per control step:
    // N = 16384 2.71 -> 2.95 ms — so ant always runs the one-wavefront-per-SIMD form)
    if (maxcol == 2 && rk == MBD_REW_ANT && nfr == 10) PK(2, MBD_REW_ANT, 10, 1, 1);
    if (maxcol == 2 && rk == -1 && nfr == 0) PK(2, -1, 0, 1, 1);
    return hipErrorInvalidValue;
  }
  // (two wavefronts per SIMD: the reference's own humanoids with one collider per link; humanoidstandup's five
  // colliders do not fit 256 registers without spilling inside the substep loop — N = 16384: 4.36 -> 4.57 ms)
  if (wpe == 2 && maxcol == 1 && rk == MBD_REW_HUMANOIDRUN && nfr == 7) PK(1, MBD_REW_HUMANOIDRUN, 7, 2);
  if (wpe == 2 && maxcol == 1 && rk == MBD_REW_HUMANOIDTRACK && nfr == 5) PK(1, MBD_REW_HUMANOIDTRACK, 5, 2);
  if (maxcol == 1 && rk == MBD_REW_HUMANOIDRUN && nfr == 7) PK(1, MBD_REW_HUMANOIDRUN, 7);
  if (maxcol == 1 && rk == MBD_REW_HUMANOIDTRACK && nfr == 5) PK(1, MBD_REW_HUMANOIDTRACK, 5);
  if (maxcol == 1 && rk == -1 && nfr == 0) PK(1, -1, 0);
  if (maxcol == 5 && rk == MBD_REW_HUMANOIDSTANDUP && nfr == 7) PK(5, MBD_REW_HUMANOIDSTANDUP, 7);
  if (maxcol == 5 && rk == -1 && nfr == 0) PK(5, -1, 0);
#undef PK
  return hipErrorInvalidValue;
}

}  // namespace mbd
