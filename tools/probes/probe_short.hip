// Shorter exact sequences.
// (1) division: with r = rcp(d) + one Newton step being the CORRECTLY ROUNDED reciprocal (probe_rcp.hip, exhaustive),
//     Markstein's theorem makes  q = n*r; e = fma(-d,q,n); q' = fma(e,r,q)  the correctly rounded n/d — one residual
//     step instead of the two of the hardware expansion.  Checked here on random pairs with full random mantissas
//     (exponents of the solver's ranges) and on the known hard denominators (mantissa all ones / all zeros).
// (2) square root: exhaustive over [1e-30, FLT_MAX] for two shorter candidates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float div5(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float div_raw(float n, float d) { return n * __builtin_amdgcn_rcpf(d); }  // control
__device__ uint32_t rng(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
__device__ float rfloat(uint64_t& s, int elo, int ehi, int kind) {  // 2^U[elo,ehi] with a random 23-bit mantissa
  uint32_t e = (uint32_t)(127 + elo + (int)(rng(s) % (uint32_t)(ehi - elo + 1)));
  uint32_t m = rng(s) & 0x7fffffu;
  if (kind == 1) m = 0x7fffffu - (rng(s) & 7u);        // mantissa (almost) all ones
  if (kind == 2) m = rng(s) & 7u;                       // mantissa (almost) all zeros
  return __builtin_bit_cast(float, (e << 23) | m);
}
__global__ void kdiv(unsigned long long* bad, float* ex, int iters, int nlo, int nhi, int dlo, int dhi, int dkind, int nkind) {
  uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
  unsigned long long b = 0, b2 = 0;
  for (int i = 0; i < iters; ++i) {
    float n = rfloat(s, nlo, nhi, nkind), d = rfloat(s, dlo, dhi, dkind);
    if (rng(s) & 1) n = -n;
    float a = n / d, c = div5(n, d);
    if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, c)) { if (!b) { ex[0] = n; ex[1] = d; ex[2] = a; ex[3] = c; } ++b; }
    if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, div_raw(n, d))) ++b2;
  }
  if (b) atomicAdd(bad, b);
  if (b2) atomicAdd(bad + 1, b2);
}
__device__ __forceinline__ float sqrtA(float x) {  // one correction with the unrefined half-reciprocal
  float r = __builtin_amdgcn_rsqf(x);
  float g = x * r, h = 0.5f * r;
  float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrtB(float x) {  // refine g only, then one correction
  float r = __builtin_amdgcn_rsqf(x);
  float g = x * r, h = 0.5f * r;
  float e = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, e, g);
  float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrtRaw(float x) { return x * __builtin_amdgcn_rsqf(x); }  // control: must mismatch
__global__ void ksqrt(unsigned long long* bad, uint32_t* ex, uint32_t lo, uint32_t hi) {
  unsigned long long b[3] = {0, 0, 0};
  for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
    float x = __builtin_bit_cast(float, (uint32_t)u);
    float ref = __builtin_sqrtf(x);
    float c[3] = {sqrtA(x), sqrtB(x), sqrtRaw(x)};
    for (int j = 0; j < 3; ++j)
      if (__builtin_bit_cast(uint32_t, ref) != __builtin_bit_cast(uint32_t, c[j])) { if (!b[j]) ex[j] = (uint32_t)u; ++b[j]; }
  }
  for (int j = 0; j < 3; ++j) if (b[j]) atomicAdd(bad + j, b[j]);
}
int main() {
  unsigned long long* bad; float* ex;
  (void)hipMalloc(&bad, 24); (void)hipMalloc(&ex, 16);
  struct { int nlo, nhi, dlo, dhi, dkind, nkind; const char* name; } cases[] = {
    {-93, 27, -66, 27, 0, 0, "n in 2^[-93,27], d in 2^[-66,27], random mantissas"},
    {-20, 10, -20, 10, 0, 0, "both in 2^[-20,10]"},
    {-93, 27, -66, 27, 1, 0, "d mantissa ~all ones"},
    {-93, 27, -66, 27, 2, 0, "d mantissa ~all zeros"},
    {-93, 27, -66, 27, 1, 1, "n and d mantissas ~all ones"},
    {-93, 27, -66, 27, 0, 2, "n mantissa ~all zeros"}};
  for (auto& c : cases) {
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL(kdiv, dim3(2048), dim3(256), 0, 0, bad, ex, 8192, c.nlo, c.nhi, c.dlo, c.dhi, c.dkind, c.nkind);
    (void)hipDeviceSynchronize();
    unsigned long long h, h2; float e[4];
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&h2, bad + 1, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(e, ex, 16, hipMemcpyDeviceToHost);
    printf("div5  %-52s mismatches %llu of %llu (control n*rcp(d): %llu)", c.name, h, 2048ull * 256 * 8192, h2);
    if (h) printf("  e.g. n=%.9g d=%.9g  /=%.9g seq=%.9g", e[0], e[1], e[2], e[3]);
    printf("\n");
  }
  (void)hipMemset(bad, 0, 24);
  float lo = 1e-30f, hi = 3.4028234e38f;
  uint32_t ulo, uhi; memcpy(&ulo, &lo, 4); memcpy(&uhi, &hi, 4);
  hipLaunchKernelGGL(ksqrt, dim3(4096), dim3(256), 0, 0, bad, (uint32_t*)ex, ulo, uhi);
  (void)hipDeviceSynchronize();
  unsigned long long h[3]; uint32_t e[3];
  (void)hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost); (void)hipMemcpy(e, ex, 12, hipMemcpyDeviceToHost);
  const char* nm[3] = {"sqrtA rsq + 1 correction (5 ops)", "sqrtB refine g, 1 correction (7 ops)", "control: x * rsq(x)"};
  for (int j = 0; j < 3; ++j) {
    printf("%-40s %llu mismatches of %llu", nm[j], h[j], (unsigned long long)uhi - ulo + 1);
    if (h[j]) { float x; memcpy(&x, &e[j], 4); printf("  e.g. x=%.9g", x); }
    printf("\n");
  }
  return 0;
}
