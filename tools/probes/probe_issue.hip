// Issue cost per instruction TYPE with ONE wavefront per SIMD (4 waves per workgroup, one workgroup per CU):
// 8 independent register chains, 64 instructions per chain per iteration, hipEvent-timed; cycles at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int KIND, int LANES = 64>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  if (LANES < 64 && (threadIdx.x & 63) >= LANES) return;  // EXEC = the first LANES lanes for the whole kernel
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2 p0 = make_float2(x0, x1), p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
  const float a = 0.999f, b = 0.001f;
  const float2 a2 = make_float2(a, a), b2 = make_float2(b, b);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
#define S(n) if (KIND == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));            \
             if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));               \
             if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##n) : "v"(a2), "v"(b2));          \
             if (KIND == 3) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x##n) : "v"(a));                       \
             if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##n) : "v"(a2));                       \
             if (KIND == 5) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x##n) : "v"(a));              \
             if (KIND == 6) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x##n) : "v"(a) : "s10", "s11"); \
             if (KIND == 7) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(x##n) : "v"(a));                           \
             if (KIND == 8) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x##n) : "v"(a), "v"(b)); \
             if (KIND == 9) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3e2aab7a" : "+v"(x##n) : "v"(a));             \
             if (KIND == 10) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x##n) : "v"(a));                      \
             if (KIND == 11) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x##n) : "v"(a));                      \
             if (KIND == 12) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));               \
             if (KIND == 13) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x##n) : "v"(a) : "vcc"); \
             if (KIND == 14) asm volatile("v_cmp_gt_f32_e64 s[10:11], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x##n) : "v"(a) : "s10", "s11"); \
             if (KIND == 15) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0" : : "v"(x##n), "v"(a) : "vcc");           \
             if (KIND == 16) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x##n) : "v"(a));                       \
             if (KIND == 18) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x##n));                                     \
             if (KIND == 19) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(x##n));                                     \
             if (KIND == 20) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(x##n));                                    \
             if (KIND == 21) asm volatile("v_fmac_f32_e32 %0, %1, %2\n\ts_nop 0" : "+v"(x##n) : "v"(a), "v"(b));     \
             if (KIND == 22) asm volatile("v_fmac_f32_e32 %0, %1, %2\n\ts_nop 1" : "+v"(x##n) : "v"(a), "v"(b));     \
             if (KIND == 23) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x##n) : "v"(a)); \
             if (KIND == 24) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x##n));                                     \
             if (KIND == 17) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n\tv_mul_f32_e32 %0, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x##n) : "v"(a) : "vcc");
      REP8(S)
#undef S
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int KIND, int LANES = 64>
void run(const char* name) {
  float* out; (void)hipMalloc(&out, 64 * 256 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 4000;
  hipLaunchKernelGGL((k<KIND, LANES>), dim3(64), dim3(256), 0, 0, out, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<KIND, LANES>), dim3(64), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double per = ms * 1e6 / ((double)iters * 64 * 8);
  printf("%-34s %.3f ns per instruction = %.2f cycles at 2.4 GHz\n", name, per, per * 2.4);
  (void)hipFree(out);
}
int main() {
  run<0>("v_fmac_f32_e32 (VOP2, 4 B)"); run<1>("v_fma_f32 a*b+c, 3 distinct (VOP3)"); run<11>("v_fma_f32 x*a+x (VOP3, 2 regs)");
  run<2>("v_pk_fma_f32 (VOP3P)"); run<3>("v_mul_f32_e32"); run<10>("v_add_f32_e32"); run<4>("v_pk_mul_f32");
  run<5>("v_cndmask_b32_e32 (vcc)"); run<6>("v_cndmask_b32_e64 (sgpr pair)"); run<7>("v_mov_b32_e32");
  run<8>("v_fmac_f32_dpp"); run<9>("v_fmaak_f32 (literal, 8 B)"); run<12>("v_med3_f32 (VOP3)");
  run<15>("v_cmp_gt_f32_e32 (writes vcc)"); run<16>("v_max_f32_e32");
  run<13>("PAIR v_cmp->vcc ; v_cndmask vcc"); run<14>("PAIR v_cmp->sgpr ; v_cndmask sgpr"); run<17>("TRIPLE cmp ; mul ; cndmask");
  run<18>("v_rcp_f32"); run<19>("v_rsq_f32"); run<20>("v_sqrt_f32"); run<24>("v_exp_f32");
  run<21>("PAIR v_fmac ; s_nop 0"); run<22>("PAIR v_fmac ; s_nop 1"); run<23>("v_mov_b32_dpp");
  // does the SIMD skip 16-lane rows whose EXEC bits are all zero?
  run<0, 16>("v_fmac_f32, EXEC = lanes 0..15"); run<0, 32>("v_fmac_f32, EXEC = lanes 0..31");
  run<2, 16>("v_pk_fma_f32, EXEC = lanes 0..15"); run<18, 16>("v_rcp_f32, EXEC = lanes 0..15");
  return 0;
}
