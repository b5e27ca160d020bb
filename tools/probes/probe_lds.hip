// Cost of a lane exchange with ONE wavefront per SIMD (the N=1024 regime of the rollout kernel):
//   batch of K ds_bpermute_b32 + one s_waitcnt + K dependent adds, versus the same exchange through DPP
//   row shifts (v_mov_dpp), versus plain VALU work of the same length.  One workgroup of 64 lanes per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
  float x[K];
  const int lane = threadIdx.x;
  const int src = ((lane & ~15) | ((lane + 1) & 15)) * 4;
#pragma unroll
  for (int j = 0; j < K; ++j) x[j] = lane * 0.001f + j;
  for (int i = 0; i < iters; ++i) {
    float y[K];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < K; ++j) y[j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, x[j])));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < K; ++j)
        y[j] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[j]), 0x101 /*row_shl:1*/, 0xf, 0xf, true));
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) y[j] = x[j] * 1.0001f;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) x[j] = __builtin_fmaf(y[j], 0.5f, x[j] * 0.25f);
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) s += x[j];
  out[blockIdx.x * 64 + lane] = s;
}
template <int K, int MODE>
void run(const char* name) {
  float* out; hipMalloc(&out, 256 * 64 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  hipLaunchKernelGGL((k<K, MODE>), dim3(256), dim3(64), 0, 0, out, 100);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<K, MODE>), dim3(256), dim3(64), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s K=%2d  %.1f ns per iteration  (%.1f cycles at 2.4 GHz)\n", name, K, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
  hipFree(out);
}
int main() {
  run<1, 2>("valu only"); run<9, 2>("valu only"); run<18, 2>("valu only");
  run<1, 0>("bpermute+wait"); run<3, 0>("bpermute+wait"); run<7, 0>("bpermute+wait"); run<9, 0>("bpermute+wait"); run<18, 0>("bpermute+wait");
  run<1, 1>("dpp row_shl:1"); run<9, 1>("dpp row_shl:1"); run<18, 1>("dpp row_shl:1");
  return 0;
}
