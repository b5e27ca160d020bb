/* mbd_hip_debug.h — test and A/B levers of libmbd_hip.so.  NOT part of the drop-in boundary (include/mbd_hip.h): nothing a
 * caller of the reference's planner needs is here, and results are bit-identical whatever the levers say — the test-suite
 * holds every alternative they select to the same checker. */
#ifndef MBD_HIP_DEBUG_H
#define MBD_HIP_DEBUG_H
#include "mbd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* One process-wide table of integer levers, initialised ONCE from the environment variables of the same names
 * (MBD_NO_DPP, MBD_NO_NFR_CONST, MBD_NO_REWARD_CONST, MBD_NO_PLANAR_FLAGS, MBD_NO_FAST_SLIDES, MBD_NO_FUSED_NOISE,
 * MBD_NO_LAZY, MBD_NO_PREFETCH, MBD_NO_AUX, MBD_WMEAN_SPLIT, MBD_NO_FUSED_SCORE, MBD_PK2, MBD_WPB, MBD_LDS_RESERVE,
 * MBD_NO_HELPERS);
 * -1 = not set: the library decides.  The launch paths read the table, never the environment. */
int mbd_debug_set(const char* name, int value);
int mbd_debug_get(const char* name, int* value_out);
/* the DPP layout family (0..3, -1: none) a model's link tree fits, with its lane <-> link table and shifts */
int mbd_debug_dpp_layout(const mbd_model_t* model, signed char tab[32], int shifts_out[4]);
/* per-wavefront clock records of the 3-D rollout kernels (tools/probes/rollout_timeline.py); d_buf: device, caller-owned */
int mbd_debug_set_clock_buffer(mbd_env* env, void* d_buf);
#ifdef __cplusplus
}
#endif
#endif
