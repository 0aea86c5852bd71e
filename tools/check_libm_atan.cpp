// tools/check_libm_atan.cpp -- host check of pine_libm.h's atanf / atan2f / acosf restatements against the container's
// libm (the one the reference links): every binary32 argument of atanf and acosf; atan2f over every special case, a
// dense grid of exponent differences and random pairs.   g++ -O2 -ffp-contract=off -std=c++17 -pthread
// Usage: check_libm_atan [random pairs per thread, default 500000000]
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../pine_amd/csrc/pine_libm.h"

static bool same(float a, float b) {
  if (a != a && b != b) return true;  // (any NaN: payloads are not compared)
  return memcmp(&a, &b, 4) == 0;
}
int main(int argc, char** argv) {
  const unsigned long long per_thread = argc > 1 ? strtoull(argv[1], nullptr, 10) : 500000000ull;
  const unsigned nt = std::thread::hardware_concurrency();
  std::atomic<unsigned long long> bad_atan{0}, bad_acos{0}, bad_atan2{0}, n1{0}, n2{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t]() {
      unsigned long long ba = 0, bc = 0, b2 = 0, c1 = 0, c2 = 0;
      for (unsigned long long b = t; b < (1ull << 32); b += nt) {
        const uint32_t u = uint32_t(b);
        float x;
        memcpy(&x, &u, 4);
        const float a = atanf(x), a2 = pine_libm::atanf_glibc(x);
        if (!same(a, a2)) { if (ba < 3) printf("atanf(%a): libm %a mine %a\n", x, a, a2); ba++; }
        const float c = acosf(x), cc = pine_libm::acosf_glibc(x);
        if (!same(c, cc)) { if (bc < 3) printf("acosf(%a): libm %a mine %a\n", x, c, cc); bc++; }
        c1++;
      }
      auto chk = [&](float y, float x) {
        const float a = atan2f(y, x), a2 = pine_libm::atan2f_glibc(y, x);
        if (!same(a, a2)) { if (b2 < 3) printf("atan2f(%a, %a): libm %a mine %a\n", y, x, a, a2); b2++; }
        c2++;
      };
      // special values x special values, and every exponent difference with a few mantissas
      const uint32_t specials[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x007fffffu, 0x00800000u, 0x3f800000u, 0xbf800000u, 0x3f000000u, 0x40490fdbu,
                                   0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x3f7fffffu, 0x3f800001u, 0x4c000000u, 0x31000000u};
      if (t == 0)
        for (uint32_t a : specials)
          for (uint32_t b : specials) {
            float y, x;
            memcpy(&y, &a, 4), memcpy(&x, &b, 4);
            chk(y, x);
          }
      for (uint32_t ey = t; ey < 256; ey += nt)
        for (uint32_t ex = 0; ex < 256; ex++)
          for (uint32_t my : {0u, 1u, 0x400000u, 0x7fffffu, 0x123456u})
            for (uint32_t mx : {0u, 1u, 0x400000u, 0x7fffffu, 0x654321u})
              for (uint32_t sg = 0; sg < 4; sg++) {
                const uint32_t a = (ey << 23) | my | ((sg & 1) << 31), b = (ex << 23) | mx | ((sg >> 1) << 31);
                float y, x;
                memcpy(&y, &a, 4), memcpy(&x, &b, 4);
                chk(y, x);
              }
      unsigned long long s = 0x9e3779b97f4a7c15ull * (t + 1);
      for (unsigned long long i = 0; i < per_thread; i++) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        uint32_t a = uint32_t(s), b = uint32_t(s >> 32);
        if (i & 1) {  // unit-vector-like arguments (what a Sphere's normal gives): exponents near 0
          a = (a & 0x80ffffffu) | ((0x70u + ((a >> 24) & 0xf)) << 23);
          b = (b & 0x80ffffffu) | ((0x70u + ((b >> 24) & 0xf)) << 23);
        }
        float y, x;
        memcpy(&y, &a, 4), memcpy(&x, &b, 4);
        chk(y, x);
      }
      bad_atan += ba, bad_acos += bc, bad_atan2 += b2, n1 += c1, n2 += c2;
    });
  for (auto& x : th) x.join();
  printf("{\"atanf_acosf_arguments\": %llu, \"atanf_mismatch\": %llu, \"acosf_mismatch\": %llu, \"atan2f_pairs\": %llu, \"atan2f_mismatch\": %llu}\n",
         n1.load(), bad_atan.load(), bad_acos.load(), n2.load(), bad_atan2.load());
  return (bad_atan.load() || bad_acos.load() || bad_atan2.load()) ? 1 : 0;
}
