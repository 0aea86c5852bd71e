// tools/check_libm_pow.cpp -- host check of pine_libm.h's powf_glibc / logf_glibc against the container's libm.
//   g++ -O2 -march=native -ffp-contract=off tools/check_libm_pow.cpp -o build/check_libm_pow -lpthread && build/check_libm_pow
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../pine_amd/csrc/pine_libm.h"

static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static float f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static bool same(float a, float b) { return !memcmp(&a, &b, 4) || (a != a && b != b); }

int main(int argc, char** argv) {
  const uint64_t pairs = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
  unsigned nt = std::thread::hardware_concurrency();
  std::atomic<uint64_t> bad_schlick{0}, bad_log{0}, bad_pairs{0}, bad_special{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t]() {
      uint64_t b1 = 0, b2 = 0, b3 = 0;
      // (1) Schlick: every float in [0, 1], exponent 5 (and 2.5, 0.5)
      for (uint64_t u = t; u <= 0x3f800000u; u += nt) {
        const float x = f32(uint32_t(u));
        for (float y : {5.0f, 2.5f, 0.5f})
          if (!same(powf(x, y), pine_libm::powf_glibc(x, y))) { if (b1 < 3 && t == 0) printf("powf(%a, %a): libm %a mine %a\n", x, y, powf(x, y), pine_libm::powf_glibc(x, y)); b1++; }
      }
      // (2) logf: every positive float (incl. subnormals, inf)
      for (uint64_t u = t; u <= 0x7f800000u; u += nt) {
        const float x = f32(uint32_t(u));
        if (!same(logf(x), pine_libm::logf_glibc(x))) { if (b2 < 3 && t == 0) printf("logf(%a): libm %a mine %a\n", x, logf(x), pine_libm::logf_glibc(x)); b2++; }
      }
      // (3) random pairs: all bit patterns for x and y, plus moderate magnitudes
      for (uint64_t k = t; k < pairs; k += nt) {
        const uint64_t h = mix(k * 2 + 1);
        float x = f32(uint32_t(h)), y = f32(uint32_t(h >> 32));
        if (k & 1) { x = f32((uint32_t(h) & 0x807fffffu) | ((110u + (uint32_t(h) >> 23) % 30u) << 23)); y = f32((uint32_t(h >> 32) & 0x807fffffu) | ((120u + (uint32_t(h >> 40)) % 12u) << 23)); }
        if (!same(powf(x, y), pine_libm::powf_glibc(x, y))) { if (b3 < 3 && t == 0) printf("powf(%a, %a): libm %a mine %a\n", x, y, powf(x, y), pine_libm::powf_glibc(x, y)); b3++; }
      }
      bad_schlick += b1; bad_log += b2; bad_pairs += b3;
    });
  for (auto& x : th) x.join();
  // (4) special values
  const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, 2.0f, -2.0f, 0.5f, -0.5f, 3.0f, -3.0f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 1e38f, 5.0f, 4.0f, 1e-40f};
  uint64_t b4 = 0;
  for (float x : sp)
    for (float y : sp)
      if (!same(powf(x, y), pine_libm::powf_glibc(x, y))) { printf("special powf(%a, %a): libm %a mine %a\n", x, y, powf(x, y), pine_libm::powf_glibc(x, y)); b4++; }
  for (float x : sp)
    if (!same(logf(x), pine_libm::logf_glibc(x))) { printf("special logf(%a): libm %a mine %a\n", x, logf(x), pine_libm::logf_glibc(x)); b4++; }
  printf("{\"schlick_mismatch\": %llu, \"logf_mismatch\": %llu, \"pair_mismatch\": %llu, \"special_mismatch\": %llu, \"pairs\": %llu}\n",
         (unsigned long long)bad_schlick.load(), (unsigned long long)bad_log.load(), (unsigned long long)bad_pairs.load(), (unsigned long long)b4, (unsigned long long)pairs);
  return (bad_schlick.load() || bad_log.load() || bad_pairs.load() || b4) ? 1 : 0;
}
