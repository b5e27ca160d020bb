// mbd_hot3d.hip — translation unit of the 3-D rollout instantiations the built-in humanoids and ant run (the DPP families
// (+1, -4, -6) and (+1, -2, -4, -6) of mbd_kernels.h).  Its own file because it is built with its own scheduler strategy
// (-mllvm -amdgpu-sched-strategy=iterative-ilp, __graft_entry__.build): same-box A/B against the default strategy,
// profiles/r03_hot3d_sched_ab.txt — humanoidrun N=1024 0.5499 -> 0.5427 ms (+1.3 %), N=4096 +1.7 %, humanoidtrack +0.8 %.
// (The general instantiations stay in mbd_env.hip with the default: one of them crashes this compiler's register
// allocator under the iterative strategy.)
#define MBD_SHARED_ONLY 1
#include "mbd_kernels.h"
#include "mbd_launch.h"

namespace mbd {

// which: 0 humanoid-shaped, one collider per link; 1 humanoid-shaped, up to five with helper lanes; 2 the same without
// helper lanes; 3 ant.  rk: the model's reward kind, or -1 (lever MBD_NO_REWARD_CONST); nfr: n_frames, or 0 (run-time).
hipError_t launch_rollout_hot3d(int which, int rk, int nfr, int device, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                const RolloutParams& P) {
#define HOT(...) return launch_rollout_kernel(rollout_kernel<__VA_ARGS__>, device, grid, block, lds, stream, P)
  constexpr int D0 = 1, D1 = -4, D2 = -6;
  if (which == 0) {
    if (rk == MBD_REW_HUMANOIDRUN && nfr == 7) HOT(16, true, false, 3, 1, D0, D1, D2, 0, false, true, 3, false, false, MBD_REW_HUMANOIDRUN, 7);
    if (rk == MBD_REW_HUMANOIDTRACK && nfr == 5) HOT(16, true, false, 3, 1, D0, D1, D2, 0, false, true, 3, false, false, MBD_REW_HUMANOIDTRACK, 5);
    HOT(16, true, false, 3, 1, D0, D1, D2, 0, false, true);
  }
  if (which == 1) {  // humanoidstandup: the torso's colliders 2..4 run stage (4) on two of the candidate's idle lanes (HELP)
    if (rk == MBD_REW_HUMANOIDSTANDUP && nfr == 7) HOT(16, true, false, 3, 5, D0, D1, D2, 0, false, true, 3, false, false, MBD_REW_HUMANOIDSTANDUP, 7, true, true);
    HOT(16, true, false, 3, 5, D0, D1, D2, 0, false, true, 3, false, false, -1, 0, true);
  }
  if (which == 2) {
    if (rk == MBD_REW_HUMANOIDSTANDUP && nfr == 7) HOT(16, true, false, 3, 5, D0, D1, D2, 0, false, true, 3, false, false, MBD_REW_HUMANOIDSTANDUP, 7);
    HOT(16, true, false, 3, 5, D0, D1, D2, 0, false, true);
  }
  if (which == 3) {  // ant (the reference's default env_name): reward kind and n_frames compiled in, like the humanoids
    if (rk == MBD_REW_ANT && nfr == 10) HOT(16, true, false, 4, 2, 1, -2, -4, -6, false, false, 3, false, false, MBD_REW_ANT, 10, false, true);
    HOT(16, true, false, 4, 2, 1, -2, -4, -6, false, false);
  }
#undef HOT
  return hipErrorInvalidValue;
}

}  // namespace mbd
