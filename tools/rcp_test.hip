// tools/rcp_test.hip -- EXHAUSTIVE check on the GPU (all 2^32 float bit patterns) of candidate short
// sequences for a correctly rounded reciprocal 1.0f/x and square root, against the IEEE-correct
// expansions hipcc emits by default.  Prints, per candidate, the mismatch count inside the guard range
// and the first few offending inputs.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/rcp_test.hip -o build/rcp_test && build/rcp_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ uint32_t fbits(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __builtin_bit_cast(float, u); }

// candidate A: one Newton step on v_rcp_f32
__device__ __forceinline__ float rcp_a(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float e0 = __builtin_fmaf(-x, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}
// candidate B: two Newton steps
__device__ __forceinline__ float rcp_b(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float e0 = __builtin_fmaf(-x, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float e1 = __builtin_fmaf(-x, r1, 1.0f);
  return __builtin_fmaf(e1, r1, r1);
}
// candidate C: the raw instruction
__device__ __forceinline__ float rcp_c(float x) { return __builtin_amdgcn_rcpf(x); }

// sqrt candidate A: v_sqrt_f32 + one residual correction with v_rsq_f32
__device__ __forceinline__ float sqrt_a(float x) {
  const float s0 = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float r = __builtin_fmaf(-s0, s0, x);
  return __builtin_fmaf(r, h, s0);
}
// sqrt candidate B: raw instruction
__device__ __forceinline__ float sqrt_b(float x) { return __builtin_amdgcn_sqrtf(x); }

// sqrt candidate C: ONE transcendental (v_rsq_f32), Goldschmidt-style: s = x*y, h = y/2, one residual step
__device__ __forceinline__ float sqrt_c(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y;
  const float h = 0.5f * y;
  const float r = __builtin_fmaf(-s0, s0, x);
  return __builtin_fmaf(r, h, s0);
}
// sqrt candidate D: C + a second residual step
__device__ __forceinline__ float sqrt_d(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y;
  const float h = 0.5f * y;
  const float r0 = __builtin_fmaf(-s0, s0, x);
  const float s1 = __builtin_fmaf(r0, h, s0);
  const float r1 = __builtin_fmaf(-s1, s1, x);
  return __builtin_fmaf(r1, h, s1);
}
constexpr int kCand = 7;
__global__ void k(uint32_t exp_lo, uint32_t exp_hi, unsigned long long* bad, uint32_t* ex) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;  // 2^24 threads x 256 values
  unsigned long long nb[kCand] = {0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < 256; it++) {
    const uint32_t u = uint32_t(t * 256 + it);
    const uint32_t e = (u >> 23) & 0xffu;
    if (e < exp_lo || e > exp_hi) continue;
    const float x = bitsf(u);
    const float ref = 1.0f / x;
    const float c[3] = {rcp_a(x), rcp_b(x), rcp_c(x)};
    for (int j = 0; j < 3; j++)
      if (fbits(c[j]) != fbits(ref)) {
        if (nb[j] == 0 && atomicAdd(&ex[j * 8], 1u) < 7u) ex[j * 8 + 1 + (atomicAdd(&ex[j * 8 + 7], 1u) % 6u)] = u;
        nb[j]++;
      }
    if (!(u >> 31)) {
      const float sref = sqrtf(x);
      const float s[4] = {sqrt_a(x), sqrt_b(x), sqrt_c(x), sqrt_d(x)};
      for (int j = 0; j < 4; j++)
        if (fbits(s[j]) != fbits(sref)) {
          if (nb[3 + j] == 0 && atomicAdd(&ex[(3 + j) * 8], 1u) < 7u) ex[(3 + j) * 8 + 1 + (atomicAdd(&ex[(3 + j) * 8 + 7], 1u) % 6u)] = u;
          nb[3 + j]++;
        }
    }
  }
  for (int j = 0; j < kCand; j++)
    if (nb[j]) atomicAdd(&bad[j], nb[j]);
}
int main(int argc, char** argv) {
  unsigned long long* bad;
  uint32_t* ex;
  hipMalloc(&bad, 8 * kCand);
  hipMalloc(&ex, 4 * 8 * kCand);
  const char* names[kCand] = {"rcp: v_rcp + 1 Newton", "rcp: v_rcp + 2 Newton", "rcp: raw v_rcp_f32", "sqrt: v_sqrt + rsq residual step",
                              "sqrt: raw v_sqrt_f32", "sqrt: rsq, s=x*y, 1 residual step", "sqrt: rsq, s=x*y, 2 residual steps"};
  const uint32_t ranges[6][2] = {{0, 255}, {2, 252}, {16, 240}, {26, 230}, {32, 222}, {64, 190}};
  for (auto& rg : ranges) {
    hipMemset(bad, 0, 8 * kCand);
    hipMemset(ex, 0, 4 * 8 * kCand);
    hipLaunchKernelGGL(k, dim3(1u << 16), dim3(256), 0, 0, rg[0], rg[1], bad, ex);
    unsigned long long hb[kCand];
    uint32_t he[8 * kCand];
    hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost);
    hipMemcpy(he, ex, sizeof he, hipMemcpyDeviceToHost);
    printf("biased exponent in [%u, %u]:\n", rg[0], rg[1]);
    for (int j = 0; j < kCand; j++) {
      printf("  %-34s mismatches %llu", names[j], hb[j]);
      for (int q = 1; q < 7 && hb[j]; q++)
        if (he[j * 8 + q]) {
          float f;
          memcpy(&f, &he[j * 8 + q], 4);
          printf("  %a", f);
        }
      printf("\n");
    }
  }
  return 0;
}
