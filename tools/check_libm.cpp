// tools/check_libm.cpp -- exhaustive host check of pine_amd/csrc/pine_libm.h against the
// container's libm (the one the reference links).  Usage: check_libm [stride]   (stride 1 = all floats with |x|<120)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>
#include "../pine_amd/csrc/pine_libm.h"

int main(int argc, char** argv) {
  uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
  const uint32_t hi = 0x42F00000u;  // 120.0f
  unsigned nt = std::thread::hardware_concurrency();
  std::atomic<uint64_t> bad_sin{0}, bad_cos{0}, total{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t]() {
      uint64_t bs = 0, bc = 0, n = 0;
      for (uint64_t b = (uint64_t)t * stride; b < hi; b += (uint64_t)nt * stride) {
        for (int sgn = 0; sgn < 2; sgn++) {
          uint32_t u = (uint32_t)b | (sgn ? 0x80000000u : 0);
          float x;
          memcpy(&x, &u, 4);
          float a = sinf(x), c = cosf(x);
          float a2 = pine_libm::sinf_glibc(x), c2 = pine_libm::cosf_glibc(x);
          if (memcmp(&a, &a2, 4)) { if (bs < 3 && t == 0) printf("sin(%a): libm %a mine %a\n", x, a, a2); bs++; }
          if (memcmp(&c, &c2, 4)) { if (bc < 3 && t == 0) printf("cos(%a): libm %a mine %a\n", x, c, c2); bc++; }
          float a3, c3;
          pine_libm::sincosf_glibc(x, a3, c3);  // the branch-free shared-reduction form the kernels call
          if (memcmp(&a, &a3, 4)) { if (bs < 3 && t == 0) printf("sincos.sin(%a): libm %a mine %a\n", x, a, a3); bs++; }
          if (memcmp(&c, &c3, 4)) { if (bc < 3 && t == 0) printf("sincos.cos(%a): libm %a mine %a\n", x, c, c3); bc++; }
          n++;
        }
      }
      bad_sin += bs; bad_cos += bc; total += n;
    });
  for (auto& x : th) x.join();
  printf("{\"checked\": %llu, \"sin_mismatch\": %llu, \"cos_mismatch\": %llu}\n",
         (unsigned long long)total.load(), (unsigned long long)bad_sin.load(), (unsigned long long)bad_cos.load());
  return (bad_sin.load() || bad_cos.load()) ? 1 : 0;
}
