// Is  r=rcp(d); e=fma(-d,r,1); r=fma(e,r,r); q=n*r; e=fma(-d,q,n); q=fma(e,r,q); e=fma(-d,q,n); q=fma(e,r,q)
// bit-identical to the correctly rounded n/d on the solver's value ranges?  (the hardware expansion of '/' is
// this sequence plus v_div_scale / v_div_fixup for the extreme-exponent cases)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float div_seq(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  e = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
__device__ uint32_t rng(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
__device__ float logu(uint64_t& s, float lo_exp, float hi_exp) {  // 10^U(lo,hi) with random mantissa
  float u = rng(s) * (1.0f / 2147483648.0f);
  float x = exp10f(lo_exp + (hi_exp - lo_exp) * u);
  return x * (1.0f + (rng(s) & 0xffff) * (1.0f / 65536.0f) * 1e-3f);
}
__global__ void k(unsigned long long* bad, float* ex, int iters, float nlo, float nhi, float dlo, float dhi, int signs) {
  uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  unsigned long long b = 0;
  for (int i = 0; i < iters; ++i) {
    float n = logu(s, nlo, nhi), d = logu(s, dlo, dhi);
    if (signs && (rng(s) & 1)) n = -n;
    if ((rng(s) & 255) == 0) n = 0.0f;
    float a = n / d, c = div_seq(n, d);
    if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, c)) { if (!b) { ex[0] = n; ex[1] = d; ex[2] = a; ex[3] = c; } ++b; }
  }
  if (b) atomicAdd(bad, b);
}
int main() {
  unsigned long long* bad; float* ex;
  hipMalloc(&bad, 8); hipMalloc(&ex, 16);
  struct { float nlo, nhi, dlo, dhi; const char* name; } cases[] = {
    {-28, 8, -20, 8, "n in [1e-28,1e8], d in [1e-20,1e8]"},
    {-10, 2, -10, 2, "both in [1e-10,1e2]"},
    {-28, -20, -20, -10, "tiny/tiny"},
    {-3, 3, -3, 3, "O(1)"},
    {-36, -28, -3, 3, "n below the flush threshold (expected to differ sometimes)"}};
  for (auto& c : cases) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, bad, ex, 1024, c.nlo, c.nhi, c.dlo, c.dhi, 1);
    hipDeviceSynchronize();
    unsigned long long h; float e[4];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(e, ex, 16, hipMemcpyDeviceToHost);
    printf("%-60s mismatches %llu of %llu", c.name, h, 1024ull * 256 * 1024);
    if (h) printf("  e.g. n=%g d=%g  /=%g seq=%g", e[0], e[1], e[2], e[3]);
    printf("\n");
  }
  return 0;
}
