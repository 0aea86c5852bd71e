// pine_amd/csrc/pine_libm.h -- single-precision sin/cos for the device that round exactly like the
// host libm the reference links against (glibc >= 2.28 sinf/cosf, i.e. the ARM "optimized routines"
// algorithm: double-precision range reduction by pi/2 and two degree-8/9 minimax polynomials
// evaluated in binary64, result rounded once to binary32).
//
// Why: the reference takes cos/sin through libm (src/psl/math.h:311-330 -> std::cos/std::sin), and
// on the cbox path they are the ONLY non-correctly-rounded operations (sampling.h:24-40
// sample_disk_concentric, :17-22 sample_disk_polar).  +,-,*,/ and sqrt are IEEE-exact on gfx950
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt, and we build with -ffp-contract=off), so
// with these two functions the device film can match the CPU reference bit for bit.
//
// The algorithm and coefficients are glibc's published ones (sysdeps/ieee754/flt-32/s_sincosf.h,
// s_sincosf_data.c; LGPL/MIT "ARM optimized routines" sinf.c); restated here, not copied from the
// reference repo (the reference contains no libm).  Verified exhaustively against the container's
// libm for |x| < 120 by tests/test_libm.py (host build of this same header).
//
// Domain: |x| < 120 handled exactly like glibc's fast path; larger |x| (never produced by the
// samplers: arguments are in [-pi/4, 2*pi]) fall back to a double-precision fmod reduction.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef PINE_HD
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PINE_HD __host__ __device__ __forceinline__
#else
#define PINE_HD inline
#endif
#endif

namespace pine_libm {

PINE_HD uint32_t asuint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
PINE_HD uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

// Polynomial data: glibc __sincosf_table[0]; table[1] is the same with c0..c4 negated.
PINE_HD float sin_poly(double x, double x2, int n, double csign) {
  // csign = +1 for table[0], -1 for table[1] (applies to the cosine coefficients only)
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
               c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  // Every a + b*c below is a fused multiply-add: x86-64 glibc dispatches sinf/cosf to its
  // -mfma build (ifunc __sinf_fma/__cosf_fma) on every FMA-capable CPU, where gcc contracts
  // exactly these patterns; with explicit fma() this header matches that build on all 2.2e9
  // floats with |x| < 120 (tools/check_libm.cpp), without them 34 inputs differ by 1 ulp.
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double t1 = __builtin_fma(x2, s3, s2);
    double x7 = x3 * x2;
    double s = __builtin_fma(x3, s1, x);
    return (float)__builtin_fma(x7, t1, s);
  } else {
    double x4 = x2 * x2;
    double t2 = __builtin_fma(x2, csign * c4, csign * c3);
    double t1 = __builtin_fma(x2, csign * c1, csign * c0);
    double x6 = x4 * x2;
    double c = __builtin_fma(x4, csign * c2, t1);
    return (float)__builtin_fma(x6, t2, c);
  }
}

// reduce_fast: x - n*pi/2 with n = round(x * 2/pi), done as ((int32)(x * 2^24 * 2/pi) + 2^23) >> 24
PINE_HD double reduce_fast(double x, int* np) {
  const double hpi_inv = 0x1.45F306DC9C883p+23;
  const double hpi = 0x1.921FB54442D18p0;
  double r = x * hpi_inv;
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return __builtin_fma(-(double)n, hpi, x);
}

PINE_HD double reduce_slow(double x, int* np) {
  // not on any sampler path (|x| >= 120); plain double reduction, within 1 ulp of libm
  const double hpi = 0x1.921FB54442D18p0;
  double q = x / hpi;
  double nq = q < 0 ? (double)(long long)(q - 0.5) : (double)(long long)(q + 0.5);
  *np = (int)((long long)nq & 3);
  return x - nq * hpi;
}

PINE_HD float sinf_glibc(float y) {
  double x = y;
  int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    double s = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return sin_poly(x, s, 0, 1.0);
  }
  if (abstop12(y) < abstop12(120.0f))
    x = reduce_fast(x, &n);
  else if (abstop12(y) < abstop12(__builtin_inff()))
    x = reduce_slow(x, &n);
  else
    return y - y;  // inf/nan -> nan
  const double sign = (n & 3) == 1 || (n & 3) == 2 ? -1.0 : 1.0;  // {1,-1,-1,1}[n&3]
  const double csign = (n & 2) ? -1.0 : 1.0;
  return sin_poly(x * sign, x * x, n, csign);
}

PINE_HD float cosf_glibc(float y) {
  double x = y;
  int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return sin_poly(x, x2, 1, 1.0);
  }
  if (abstop12(y) < abstop12(120.0f))
    x = reduce_fast(x, &n);
  else if (abstop12(y) < abstop12(__builtin_inff()))
    x = reduce_slow(x, &n);
  else
    return y - y;
  const double sign = (n & 3) == 1 || (n & 3) == 2 ? -1.0 : 1.0;
  const double csign = (n & 2) ? -1.0 : 1.0;
  return sin_poly(x * sign, x * x, n ^ 1, csign);
}

// sin and cos of the same argument with one shared range reduction (the two results are exactly
// sinf_glibc(y) and cosf_glibc(y): same reduction, same polynomials).
PINE_HD void sincosf_glibc(float y, float& sn, float& cs) {
  double x = y;
  int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    const double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) {
      sn = y;
      cs = 1.0f;
      return;
    }
    sn = sin_poly(x, x2, 0, 1.0);
    cs = sin_poly(x, x2, 1, 1.0);
    return;
  }
  if (abstop12(y) < abstop12(120.0f))
    x = reduce_fast(x, &n);
  else if (abstop12(y) < abstop12(__builtin_inff()))
    x = reduce_slow(x, &n);
  else {
    sn = cs = y - y;
    return;
  }
  const double sign = (n & 3) == 1 || (n & 3) == 2 ? -1.0 : 1.0;
  const double csign = (n & 2) ? -1.0 : 1.0;
  const double xs = x * sign, x2 = x * x;
  sn = sin_poly(xs, x2, n, csign);
  cs = sin_poly(xs, x2, n ^ 1, csign);
}

}  // namespace pine_libm
