// Issue cost of SCALAR and BRANCH instructions with ONE wavefront per SIMD (the rollout kernels' regime), measured as the
// extra time of a pair "v_fmac ; X" over a lone v_fmac: 8 independent register chains, 64 pairs per chain per iteration.
// Round 6: the early-out of the planar kernels puts not-taken branches and a few scalar instructions on the common path of
// every substep — what do they cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const float a = 0.999f, b = 0.001f;
  asm volatile("s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[12:13], 0\n\ts_cmp_lg_u64 s[10:11], 0\n\ts_mov_b64 vcc, 0" ::: "s10", "s11", "s12", "s13", "scc", "vcc");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
#define F "v_fmac_f32_e32 %0, %1, %2\n\t"
#define S(n) if (KIND == 0) asm volatile(F : "+v"(x##n) : "v"(a), "v"(b));            \
             if (KIND == 1) asm volatile(F "s_cbranch_scc1 .Lx%=\n.Lx%=:" : "+v"(x##n) : "v"(a), "v"(b));           /* not taken: SCC = 0 */ \
             if (KIND == 2) asm volatile(F "s_cbranch_vccnz .Lx%=\n.Lx%=:" : "+v"(x##n) : "v"(a), "v"(b));          /* not taken: VCC = 0 */ \
             if (KIND == 3) asm volatile(F "s_cbranch_scc0 .Lx%=\n\ts_nop 0\n.Lx%=:" : "+v"(x##n) : "v"(a), "v"(b));  /* TAKEN over one s_nop */ \
             if (KIND == 4) asm volatile(F "s_and_b64 s[12:13], s[10:11], exec" : "+v"(x##n) : "v"(a), "v"(b) : "s12", "s13", "scc"); \
             if (KIND == 5) asm volatile(F "s_mov_b64 s[12:13], -1" : "+v"(x##n) : "v"(a), "v"(b) : "s12", "s13");     \
             if (KIND == 6) asm volatile(F "s_cmp_lg_u64 s[10:11], 0" : "+v"(x##n) : "v"(a), "v"(b) : "scc");          \
             if (KIND == 7) asm volatile(F "v_cmp_gt_f32_e64 s[12:13], %1, %0\n\ts_or_b64 s[14:15], s[14:15], s[12:13]" : "+v"(x##n) : "v"(a), "v"(b) : "s12", "s13", "s14", "s15", "scc"); \
             if (KIND == 8) asm volatile(F "v_cmp_gt_f32_e32 vcc, %1, %0\n\ts_cbranch_vccnz .Lx%=\n.Lx%=:" : "+v"(x##n) : "v"(b), "v"(a) : "vcc");  /* compare + not-taken branch on it */ \
             if (KIND == 9) asm volatile(F "v_cmp_gt_f32_e32 vcc, %1, %0" : "+v"(x##n) : "v"(b), "v"(a) : "vcc");     \
             if (KIND == 10) asm volatile(F "s_and_b64 s[12:13], s[10:11], exec\n\ts_cbranch_scc1 .Lx%=\n.Lx%=:" : "+v"(x##n) : "v"(a), "v"(b) : "s12", "s13", "scc"); \
             if (KIND == 11) asm volatile(F "s_nop 0" : "+v"(x##n) : "v"(a), "v"(b));
      REP8(S)
#undef S
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
static double base = 0;
template <int KIND>
void run(const char* name) {
  float* out; (void)hipMalloc(&out, 64 * 256 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND>), dim3(64), dim3(256), 0, 0, out, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<KIND>), dim3(64), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double per = ms * 1e6 / ((double)iters * 64 * 8) * 2.4;
  if (KIND == 0) base = per;
  printf("%-52s %6.2f cycles per pair  -> X = %6.2f cycles (at 2.4 GHz)\n", name, per, per - base);
  (void)hipFree(out);
}
int main() {
  run<0>("v_fmac alone");
  run<11>("v_fmac ; s_nop 0");
  run<1>("v_fmac ; s_cbranch_scc1 NOT taken");
  run<2>("v_fmac ; s_cbranch_vccnz NOT taken");
  run<3>("v_fmac ; s_cbranch_scc0 TAKEN (over one s_nop)");
  run<4>("v_fmac ; s_and_b64");
  run<5>("v_fmac ; s_mov_b64");
  run<6>("v_fmac ; s_cmp_lg_u64");
  run<9>("v_fmac ; v_cmp_e32 vcc");
  run<8>("v_fmac ; v_cmp_e32 vcc ; s_cbranch_vccnz NOT taken");
  run<7>("v_fmac ; v_cmp_e64 sgpr ; s_or_b64 (sticky mask)");
  run<10>("v_fmac ; s_and_b64 ; s_cbranch_scc1 NOT taken");
  return 0;
}
