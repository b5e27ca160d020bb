// Dependent-issue latency of FP32 VALU instructions with ONE wavefront per SIMD: C independent chains of
// v_fma_f32 / v_pk_fma_f32 / v_mul_f32 interleaved round-robin, 64 instructions per chain per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C, int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
  float x[8]; float2 p[8];
  for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 0.001f + j; p[j] = make_float2(x[j], x[j] + 1); }
  const float a = 0.999f, b = 0.001f;
  const float2 a2 = make_float2(a, a), b2 = make_float2(b, b);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(a2), "v"(b2));
        if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
        if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a));
        if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += x[j] + p[j].x + p[j].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int C, int KIND>
void run(const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 64 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 2000;
  hipLaunchKernelGGL((k<C, KIND>), dim3(256), dim3(64), 0, 0, out, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<C, KIND>), dim3(256), dim3(64), 0, 0, out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double per = ms * 1e6 / ((double)iters * 64 * C);
  printf("%-14s %d chain(s): %.2f ns per instruction = %.2f cycles at 2.4 GHz\n", name, C, per, per * 2.4);
  (void)hipFree(out);
}
int main() {
  run<1, 0>("v_fma_f32"); run<2, 0>("v_fma_f32"); run<3, 0>("v_fma_f32"); run<4, 0>("v_fma_f32"); run<8, 0>("v_fma_f32");
  run<1, 1>("v_pk_fma_f32"); run<2, 1>("v_pk_fma_f32"); run<3, 1>("v_pk_fma_f32"); run<4, 1>("v_pk_fma_f32"); run<8, 1>("v_pk_fma_f32");
  run<1, 2>("v_mul_f32"); run<2, 2>("v_mul_f32"); run<4, 2>("v_mul_f32");
  run<1, 3>("v_cndmask_b32"); run<2, 3>("v_cndmask_b32"); run<4, 3>("v_cndmask_b32");
  run<1, 4>("v_rcp_f32"); run<2, 4>("v_rcp_f32"); run<4, 4>("v_rcp_f32");
  return 0;
}
