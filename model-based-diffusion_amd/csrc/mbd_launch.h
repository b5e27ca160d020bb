// mbd_launch.h — the one launch site of every rollout instantiation (host side; shared by the translation units that
// hold rollout kernels: mbd_env.hip, mbd_hot3d.hip, mbd_pk2.hip, mbd_planar.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <set>
#include <utility>

#include "mbd_kernels.h"

namespace mbd {

// One launch site for every instantiation.  lds > 0 reserves dynamic LDS the kernel never touches: more than half
// of a CU's 160 KB keeps a second workgroup — of this or of a concurrent plan's launch — off the CU, so concurrent
// plans spread over the chip instead of piling onto the CUs the dispatcher fills first (tools/gpu_concurrent.sh).
template <typename K>
inline hipError_t launch_rollout_kernel(K kernel, int device, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                 const RolloutParams& P) {
  if (lds > 0) {
    // > 64 KB of dynamic LDS needs the attribute on EVERY instantiation that is launched that way; all of them share
    // this function's signature, so the bookkeeping is keyed on (kernel address, device)
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> raised;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair((const void*)kernel, device);
    if (!raised.count(key)) {
      hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e != hipSuccess) return e;
      raised.insert(key);
    }
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, P);
  return hipGetLastError();
}


// the two-candidates-per-lane rollouts of the humanoid family (mbd_pk2.h; their own translation unit, mbd_pk2.hip, built
// with the scheduler strategy that keeps dependent packed instructions apart).  hipErrorInvalidValue: no such instantiation.
// wpe: 2 asks for the instantiation whose registers leave room for two wavefronts per SIMD (where there is one).
// the 3-D instantiations the built-in humanoids and ant run (their own translation unit, mbd_hot3d.hip): which = 0 humanoid
// with one collider per link, 1 / 2 up to five with / without helper lanes, 3 ant; rk = -1, nfr = 0: the run-time forms
hipError_t launch_rollout_hot3d(int which, int rk, int nfr, int device, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                const RolloutParams& P);
// the planar rollouts (mbd_planar.h; their own translation unit, mbd_planar.hip): lps lanes per candidate, the env's DPP
// family, colliders per link, fl = the model's switches (1 springs | 2 slide limits | 4 elasticity), reward kind, n_frames
// (0: run-time), no_fl: the general instantiation (lever MBD_NO_PLANAR_FLAGS), spec: the model carries specification
// switches (MBD_SPEC_FLAGS): the SPEC instantiation (16 lanes, shuffle exchange) reads them at run time
hipError_t launch_rollout_planar(int lps, int dpp_family, int max_col, int fl, int rk, int nfr, bool no_fl, bool spec, int device,
                                 dim3 grid, dim3 block, size_t lds, hipStream_t stream, const RolloutParams& P);
// whether an early-out instantiation (RolloutParams::cpw) exists for such a model with its switches as compile-time constants
bool planar_has_early_out(int lps, int dpp_family, int max_col, int fl, int rk, int nfr);
// fam: 0 the humanoid family, 1 ant (mbd_pk2.h)
hipError_t launch_rollout_pk2(int fam, int maxcol, int rk, int nfr, int wpe, int device, dim3 grid, dim3 block, size_t lds,
                              hipStream_t stream, const RolloutParams& P);
// (maxcol, rk, nfr) the launcher would run for a model with `max_col` colliders per link, reward kind `rk`, `nfr` frames
bool pk2_instantiation(int fam, int max_col, int rk, int nfr, int out[3]);

}  // namespace mbd
