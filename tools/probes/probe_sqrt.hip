// Is  r=rsq(x); g=x*r; h=0.5*r; e=fma(-h,g,0.5); g=fma(g,e,g); h=fma(h,e,h); d=fma(-g,g,x); s=fma(d,h,g)
// bit-identical to the correctly rounded sqrt(x)?  Exhaustive over every float32 in [1e-30, FLT_MAX].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float sqrt_seq(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  float g = x * r, h = 0.5f * r;
  float e = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, e, g);
  h = __builtin_fmaf(h, e, h);
  float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
__global__ void k(unsigned long long* bad, uint32_t* ex, uint32_t lo, uint32_t hi) {
  unsigned long long b = 0;
  for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
    float x = __builtin_bit_cast(float, (uint32_t)u);
    float a = __builtin_sqrtf(x), c = sqrt_seq(x);
    if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, c)) { if (!b) ex[0] = (uint32_t)u; ++b; }
  }
  if (b) atomicAdd(bad, b);
}
int main() {
  unsigned long long* bad; uint32_t* ex;
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&ex, 4);
  (void)hipMemset(bad, 0, 8);
  float lo = 1e-30f, hi = 3.4028234e38f;
  uint32_t ulo, uhi; memcpy(&ulo, &lo, 4); memcpy(&uhi, &hi, 4);
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, ex, ulo, uhi);
  (void)hipDeviceSynchronize();
  unsigned long long h; uint32_t e;
  (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&e, ex, 4, hipMemcpyDeviceToHost);
  printf("floats in [1e-30, FLT_MAX]: %llu checked, %llu mismatches", (unsigned long long)uhi - ulo + 1, h);
  if (h) { float x; memcpy(&x, &e, 4); printf("  e.g. x=%g (0x%08x)", x, e); }
  printf("\n");
  return 0;
}
