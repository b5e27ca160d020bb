// Where do single-wave workgroups land? Records HW_ID / XCC_ID and start/end clocks per workgroup for a
// register-heavy (2 waves/SIMD) dependent-FMA kernel, then prints waves-per-CU / waves-per-SIMD histograms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(64) void probe(unsigned* out, float* sink, int iters) {
  unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float a[96];
#pragma unroll
  for (int k = 0; k < 96; ++k) a[k] = threadIdx.x * 0.001f + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 96; ++k) a[k] = __builtin_fmaf(a[k], 1.0001f, a[(k + 1) % 96] * 1e-6f);
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < 96; ++k) s += a[k];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)(t0 & 0xffffffffu); out[blockIdx.x * 4 + 3] = (unsigned)((t1 - t0) & 0xffffffffu);
  }
  if (s == 12345.678f) sink[0] = s;
}
int main() {
  unsigned* d; float* sink;
  hipMalloc(&d, 16384 * 16); hipMalloc(&sink, 4);
  for (int G : {256, 512, 1024, 2048, 4096}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, d, sink, 200);  // warm
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, d, sink, 2000); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(G * 4); hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> percu, persimd; std::map<unsigned,int> perxcc;
    double avgdur = 0;
    for (int b = 0; b < G; ++b) {
      unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
      unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      percu[cukey]++; persimd[(cukey << 2) | simd]++; perxcc[xcc]++;
      avgdur += h[b * 4 + 3];
    }
    std::map<int,int> hcu, hsimd;
    for (auto& kv : percu) hcu[kv.second]++;
    for (auto& kv : persimd) hsimd[kv.second]++;
    printf("G=%5d  %.3f ms  CUs used %zu  SIMDs used %zu  avg wave dur %.0f ticks | waves/CU histogram:", G, ms, percu.size(), persimd.size(), avgdur / G);
    for (auto& kv : hcu) printf(" %dx%d", kv.first, kv.second);
    printf(" | waves/SIMD:");
    for (auto& kv : hsimd) printf(" %dx%d", kv.first, kv.second);
    printf("\n");
  }
  return 0;
}
