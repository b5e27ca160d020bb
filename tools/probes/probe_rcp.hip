// Is a short reciprocal sequence bit-identical to the correctly rounded 1.0f / d (what div_(1.0f, d) of the contract
// computes)?  Exhaustive over every float32 d in [1e-20, 1e20], three candidates:
//   A  r = rcp(d); e = fma(-d, r, 1); r = fma(e, r, r)                                   (one Newton step)
//   B  A, then e = fma(-d, r, 1); r = fma(e, r, r)                                        (two Newton steps)
//   C  the contract's division sequence with n = 1 (what the kernels ran before)         (control: must be 0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float rcpA(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcpB(float d) {
  float r = rcpA(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcpC(float d) {
  float n = 1.0f;
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  e = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
__global__ void k(unsigned long long* bad, uint32_t* ex, uint32_t lo, uint32_t hi) {
  unsigned long long b[3] = {0, 0, 0};
  for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
    float d = __builtin_bit_cast(float, (uint32_t)u);
    float ref = 1.0f / d;  // (-fhip-fp32-correctly-rounded-divide-sqrt)
    float c[3] = {rcpA(d), rcpB(d), rcpC(d)};
    for (int j = 0; j < 3; ++j)
      if (__builtin_bit_cast(uint32_t, ref) != __builtin_bit_cast(uint32_t, c[j])) { if (!b[j]) ex[j] = (uint32_t)u; ++b[j]; }
  }
  for (int j = 0; j < 3; ++j) if (b[j]) atomicAdd(bad + j, b[j]);
}
int main() {
  unsigned long long* bad; uint32_t* ex;
  (void)hipMalloc(&bad, 24); (void)hipMalloc(&ex, 12);
  (void)hipMemset(bad, 0, 24);
  float lo = 1e-20f, hi = 1e20f;
  uint32_t ulo, uhi; memcpy(&ulo, &lo, 4); memcpy(&uhi, &hi, 4);
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, ex, ulo, uhi);
  (void)hipDeviceSynchronize();
  unsigned long long h[3]; uint32_t e[3];
  (void)hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost); (void)hipMemcpy(e, ex, 12, hipMemcpyDeviceToHost);
  printf("floats in [1e-20, 1e20]: %llu checked\n", (unsigned long long)uhi - ulo + 1);
  const char* nm[3] = {"A rcp + 1 Newton step", "B rcp + 2 Newton steps", "C contract division, n = 1"};
  for (int j = 0; j < 3; ++j) {
    printf("  %-28s %llu mismatches", nm[j], h[j]);
    if (h[j]) { float x; memcpy(&x, &e[j], 4); printf("  e.g. d=%.9g (0x%08x)", x, e[j]); }
    printf("\n");
  }
  return 0;
}
