// tools/valu_rate.hip -- instruction-issue rates of one gfx950 SIMD at 1 / 2 / 4 waves per SIMD (the path kernels run at
// 4): ns and cycles per wave-instruction per SIMD for the instruction kinds the kernels are made of.  The path kernels are
// bound by exactly this: instructions of ANY kind issue at about one per 2.3 cycles per SIMD.
//   hipcc -O3 -w --offload-arch=gfx950 tools/valu_rate.hip -o build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL(NAME, ASM8, CLOB...)                                                                              \
  __global__ void __launch_bounds__(1024) NAME(float* out, int iters, float a, float b) {                        \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3;                                                                   \
    for (int i = 0; i < iters; i++) {                                                                            \
      _Pragma("unroll") for (int k = 0; k < 8; k++)                                                              \
          asm volatile(ASM8 ASM8                                                                                 \
                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) \
                       : "v"(a), "v"(b)                                                                          \
                       : CLOB);                                                                                  \
    }                                                                                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + float(d0 + d1 + d2 + d3); \
  }
// operands: %0..%7 floats, %8..%11 doubles, %12 = a, %13 = b
#define I8(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"
KERNEL(k_mul, I8("v_mul_f32", ", %12"), "memory")
KERNEL(k_add, I8("v_add_f32", ", %12"), "memory")
KERNEL(k_fma, I8("v_fma_f32", ", %12, %13"), "memory")
KERNEL(k_max3, I8("v_max3_f32", ", %12, %13"), "memory")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n", "memory")
KERNEL(k_and, I8("v_and_b32", ", %12"), "memory")
KERNEL(k_addu, I8("v_add_u32", ", %12"), "memory")
KERNEL(k_lshl, I8("v_lshlrev_b32", ", 1"), "memory")
KERNEL(k_mullo, I8("v_mul_lo_u32", ", %12"), "memory")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %12\n v_cmp_lt_f32 vcc, %1, %12\n v_cmp_lt_f32 vcc, %2, %12\n v_cmp_lt_f32 vcc, %3, %12\n v_cmp_lt_f32 vcc, %4, %12\n v_cmp_lt_f32 vcc, %5, %12\n v_cmp_lt_f32 vcc, %6, %12\n v_cmp_lt_f32 vcc, %7, %12\n", "vcc")
KERNEL(k_cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %12\n v_cmp_lt_f32 s[22:23], %1, %12\n v_cmp_lt_f32 s[24:25], %2, %12\n v_cmp_lt_f32 s[26:27], %3, %12\n v_cmp_lt_f32 s[20:21], %4, %12\n v_cmp_lt_f32 s[22:23], %5, %12\n v_cmp_lt_f32 s[24:25], %6, %12\n v_cmp_lt_f32 s[26:27], %7, %12\n", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
KERNEL(k_cndmask, I8("v_cndmask_b32", ", %12, vcc"), "memory")
KERNEL(k_rcp, I8("v_rcp_f32", ""), "memory")
KERNEL(k_sqrt, I8("v_sqrt_f32", ""), "memory")
KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %12, %0\n v_div_scale_f32 %1, vcc, %1, %12, %1\n v_div_scale_f32 %2, vcc, %2, %12, %2\n v_div_scale_f32 %3, vcc, %3, %12, %3\n v_div_scale_f32 %4, vcc, %4, %12, %4\n v_div_scale_f32 %5, vcc, %5, %12, %5\n v_div_scale_f32 %6, vcc, %6, %12, %6\n v_div_scale_f32 %7, vcc, %7, %12, %7\n", "vcc")
KERNEL(k_divfmas, I8("v_div_fmas_f32", ", %12, %13"), "memory")
KERNEL(k_divfixup, I8("v_div_fixup_f32", ", %12, %13"), "memory")
KERNEL(k_fma64, "v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %9, %9, %9, %9\n v_fma_f64 %10, %10, %10, %10\n v_fma_f64 %11, %11, %11, %11\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %9, %9, %9, %9\n v_fma_f64 %10, %10, %10, %10\n v_fma_f64 %11, %11, %11, %11\n", "memory")
KERNEL(k_cvt64, "v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %1\n v_cvt_f64_f32 %10, %2\n v_cvt_f64_f32 %11, %3\n v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %9\n v_cvt_f32_f64 %6, %10\n v_cvt_f32_f64 %7, %11\n", "memory")
KERNEL(k_salu, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc")
KERNEL(k_salu64, "s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec\n s_and_b64 s[24:25], s[24:25], exec\n s_or_b64 s[26:27], s[26:27], exec\n s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec\n s_and_b64 s[24:25], s[24:25], exec\n s_or_b64 s[26:27], s[26:27], exec\n", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc")
KERNEL(k_valu_salu, "v_mul_f32 %0, %0, %12\n s_add_u32 s20, s20, 1\n v_mul_f32 %1, %1, %12\n s_add_u32 s21, s21, 1\n v_mul_f32 %2, %2, %12\n s_add_u32 s22, s22, 1\n v_mul_f32 %3, %3, %12\n s_add_u32 s23, s23, 1\n", "s20", "s21", "s22", "s23", "scc")
KERNEL(k_readfirst, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n v_readfirstlane_b32 s24, %4\n v_readfirstlane_b32 s25, %5\n v_readfirstlane_b32 s26, %6\n v_readfirstlane_b32 s27, %7\n", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
KERNEL(k_nop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n", "memory")
KERNEL(k_branch, "s_cbranch_scc1 1f\n1: s_cbranch_scc1 2f\n2: s_cbranch_scc1 3f\n3: s_cbranch_scc1 4f\n4: s_cbranch_scc1 5f\n5: s_cbranch_scc1 6f\n6: s_cbranch_scc1 7f\n7: s_cbranch_scc1 8f\n8:\n", "memory")

template <typename K>
static void run(K kernel, const char* what, int insts_per_asm) {
  float* out;
  (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 10000;
  printf("%-34s", what);
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 0, 0, out, 100, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = double(iters) * 8 * 2 * insts_per_asm;
    const double ns = ms * 1e6 / (insts_per_wave * (threads / 256.0));
    printf("  %dw/SIMD %6.3f ns = %5.2f cyc", threads / 256, ns, ns * 2.4);
  }
  printf("\n");
  (void)hipFree(out);
}

int main() {
  printf("ns per wave-instruction per SIMD (and cycles at a nominal 2.4 GHz), 256 workgroups = one per CU\n");
#define RUN(k, n) run(k, #k, n)
  RUN(k_mul, 8); RUN(k_add, 8); RUN(k_fma, 8); RUN(k_max3, 8); RUN(k_mov, 8); RUN(k_and, 8); RUN(k_addu, 8); RUN(k_lshl, 8); RUN(k_mullo, 8);
  RUN(k_cmp, 8); RUN(k_cmp_sgpr, 8); RUN(k_cndmask, 8); RUN(k_rcp, 8); RUN(k_sqrt, 8); RUN(k_divscale, 8); RUN(k_divfmas, 8); RUN(k_divfixup, 8);
  RUN(k_fma64, 8); RUN(k_cvt64, 8); RUN(k_salu, 8); RUN(k_salu64, 8); RUN(k_valu_salu, 8); RUN(k_readfirst, 8); RUN(k_nop, 8); RUN(k_branch, 8);
  return 0;
}
