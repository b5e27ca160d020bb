// mbd_planar.hip — translation unit of the planar rollouts (mbd_planar.h).  Its own file because it is built with its own
// scheduler strategy (-mllvm -amdgpu-sched-strategy=max-ilp, __graft_entry__.build): the planar substeps are short
// dependent chains around a few packed instructions, and the default strategy leaves 11-12 hazard s_nop per substep
// where max-ilp leaves 4-5 (hopper 342 -> 336 instructions per substep, halfcheetah 398 -> 391, walker2d 371 -> 363:
// a lone wavefront's time is its instruction count).  The 3-D kernels (mbd_env.hip) keep the default.
#define MBD_SHARED_ONLY 1
#include "mbd_planar.h"
#include "mbd_launch.h"

namespace mbd {

// the early-out instantiations (EO: P.cpw candidates per wavefront, mbd_planar.h): (lps, family, fl, rk, nfr) of the built-in
// models with contacts — hopper, walker2d, halfcheetah
bool planar_has_early_out(int lps, int dpp_family, int max_col, int fl, int rk, int nfr) {
  if (max_col != 2 || (MBD_TUNED_SPEC & MBD_FLAG_CONTACT6_GAUSS_SEIDEL) != 0) return false;  // (stage (6) as a packed pair is Jacobi)
  if (lps == 4 && dpp_family == 2) return fl == 0 && rk == MBD_REW_HOPPER && nfr == 20;
  if (lps == 8 && dpp_family == 1)
    return (fl == 0 && rk == MBD_REW_HOPPER && nfr == 20) || (fl == 1 && rk == MBD_REW_HALFCHEETAH);
  return false;
}

hipError_t launch_rollout_planar(int lps, int dpp_family, int max_col, int fl, int rk, int nfr, bool no_fl, bool spec, int device,
                                 dim3 grid, dim3 block, size_t lds, hipStream_t stream, const RolloutParams& P) {
#define PL(...) return launch_rollout_kernel(rollout_planar_kernel<__VA_ARGS__>, device, grid, block, lds, stream, P)
  if (max_col > 2) {  // three or four spheres on a link (round 6: collide_all_capsules puts four on the halfcheetah's torso)
    if (spec) PL(16, 4, 0, 0, -1, -1, 0, true);
    if (lps == 8 && dpp_family == 1) {
      if (fl == 1 && rk == MBD_REW_HALFCHEETAH && !no_fl && nfr == 16) PL(8, 4, 1, -3, 1, MBD_REW_HALFCHEETAH, 16);
      PL(8, 4, 1, -3);
    }
    if (lps == 4) PL(4, 4, 0, 0);
    if (lps == 8) PL(8, 4, 0, 0);
    PL(16, 4, 0, 0);
  }
  if (spec) PL(16, 2, 0, 0, -1, -1, 0, true);  // specification switches at run time
#if (MBD_TUNED_SPEC & 8) == 0
  if (P.cpw > 0) {  // (the launch geometry asked planar_has_early_out first)
    if (lps == 4) PL(4, 2, 1, 0, 0, MBD_REW_HOPPER, 20, false, true);
    if (rk == MBD_REW_HOPPER) PL(8, 2, 1, -3, 0, MBD_REW_HOPPER, 20, false, true);
    PL(8, 2, 1, -3, 1, MBD_REW_HALFCHEETAH, 0, false, true);
  }
#endif
  // (... the reward kind: cartpole, hopper, walker2d, halfcheetah; and n_frames, for the values the built-in models have:
  // NFR; halfcheetah's 16 since round 6 — see below)
  if (lps == 4 && dpp_family == 2) {
    if (max_col == 0) {
      if (fl == 2 && rk == MBD_REW_CARTPOLE && !no_fl && nfr == 4) PL(4, 0, 1, 0, 2, MBD_REW_CARTPOLE, 4);
      else if (fl == 2 && rk == MBD_REW_CARTPOLE && !no_fl) PL(4, 0, 1, 0, 2, MBD_REW_CARTPOLE);
      else PL(4, 0, 1, 0);
    }
    else if (fl == 0 && rk == MBD_REW_HOPPER && !no_fl && nfr == 20) PL(4, 2, 1, 0, 0, MBD_REW_HOPPER, 20);
    else if (fl == 0 && rk == MBD_REW_HOPPER && !no_fl) PL(4, 2, 1, 0, 0, MBD_REW_HOPPER);
    else PL(4, 2, 1, 0);
  } else if (lps == 8 && dpp_family == 1) {
    if (fl == 0 && rk == MBD_REW_HOPPER && !no_fl && nfr == 20) PL(8, 2, 1, -3, 0, MBD_REW_HOPPER, 20);
    else if (fl == 0 && rk == MBD_REW_HOPPER && !no_fl) PL(8, 2, 1, -3, 0, MBD_REW_HOPPER);
    // (round 6: n_frames = 16 as 2 x 8 substeps in line now pays, +0.7 % — without the renormalisation's two branches per
    // substep the body is shorter; rounds 2-5 measured -0.2 %)
    else if (fl == 1 && rk == MBD_REW_HALFCHEETAH && !no_fl && nfr == 16) PL(8, 2, 1, -3, 1, MBD_REW_HALFCHEETAH, 16);
    else if (fl == 1 && rk == MBD_REW_HALFCHEETAH && !no_fl) PL(8, 2, 1, -3, 1, MBD_REW_HALFCHEETAH);
    else PL(8, 2, 1, -3);
  }
  else if (lps == 8 && dpp_family == 2) PL(8, 2, 1, 0);
  else if (lps == 4) PL(4, 2, 0, 0);
  else if (lps == 8) PL(8, 2, 0, 0);
  else PL(16, 2, 0, 0);
#undef PL
}

}  // namespace mbd
