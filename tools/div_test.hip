// tools/div_test.hip -- brute-force check on the GPU that pine_math.h's guarded fast division is
// bit-identical to the IEEE-correct `/` hipcc emits by default.  hipcc -O3 -ffp-contract=off
// --offload-arch=gfx950 tools/div_test.hip -o div_test && ./div_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../pine_amd/csrc/pine_math.h"
using namespace pine_gpu;

__device__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

__global__ void k(uint64_t seed, int mode, unsigned long long* bad, unsigned long long* fast_taken, float* ex) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  unsigned long long nb = 0, nf = 0;
  for (int it = 0; it < 4096; it++) {
    uint64_t h = mix(seed + i * 4096 + it);
    uint32_t a = (uint32_t)h, b = (uint32_t)(h >> 32);
    float n, d;
    if (mode == 0) { n = as_float((int)a); d = as_float((int)b); }                 // all bit patterns
    else if (mode == 1) { n = as_float((int)((a & 0x807fffffu) | ((100 + (a >> 23) % 56) << 23))); d = as_float((int)((b & 0x807fffffu) | ((100 + (b >> 23) % 56) << 23))); }  // moderate exponents
    else { n = (float)(a % 100000) * 1e-3f; d = (float)(b % 4096 + 1) * 0.25f; }   // "renderer-like" values
    float q0 = n / d;
    bool fast;
    float q1 = pdiv_checked(n, d, fast);
    nf += fast;
    if (as_int(q0) != as_int(q1) && !(q0 != q0 && q1 != q1)) { nb++; if (nb == 1 && ex[0] == 0) { ex[0] = n; ex[1] = d; ex[2] = q0; ex[3] = q1; } }
  }
  atomicAdd(bad, nb);
  atomicAdd(fast_taken, nf);
}
int main() {
  unsigned long long *bad, *fast; float* ex;
  hipMalloc(&bad, 8); hipMalloc(&fast, 8); hipMalloc(&ex, 16);
  for (int mode = 0; mode < 3; mode++) {
    hipMemset(bad, 0, 8); hipMemset(fast, 0, 8); hipMemset(ex, 0, 16);
    for (int r = 0; r < 8; r++) hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, 0x1234567ull * (r + 1) + mode, mode, bad, fast, ex);
    unsigned long long hb, hf; float he[4];
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fast, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost);
    printf("mode %d: pairs %llu fast-path %llu mismatches %llu", mode, 8ull * 4096 * 256 * 4096, hf, hb);
    if (hb) printf("  e.g. %a / %a = %a vs %a", he[0], he[1], he[2], he[3]);
    printf("\n");
  }
  return 0;
}
