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
PINE_HD float asfloat(uint32_t u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

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

// sin and cos of the same argument with one shared range reduction: exactly sinf_glibc(y) and
// cosf_glibc(y), written without data-dependent branches for 64-wide execution.  What makes that possible:
//  * for |y| < pi/4 the general reduction yields n = 0 and leaves x untouched, so the short path of
//    sinf/cosf is the general path's arithmetic with n = 0 (same operations, same operands);
//  * one of the two results is always the odd "sine" polynomial and the other the even "cosine" one
//    (n and n^1 differ in parity): both are evaluated once and swapped by the parity of n;
//  * the factors `sign` (on x) and `csign` (on the cosine coefficients) are +-1: the sine polynomial is odd
//    in x and every step of either polynomial is a round-to-nearest operation, so multiplying the inputs
//    by -1 negates the result exactly -- the sign is applied to the final float instead.
// Concentric-disk sampling feeds this with theta in [-pi/4, 3pi/4]: half the lanes of a wave had n = 0 and
// half n = 1, and the branchy form executed four polynomials plus both reductions for every wave.
PINE_HD void sincosf_glibc(float y, float& sn, float& cs) {
  const uint32_t top = abstop12(y);
  if (__builtin_expect(!(top < abstop12(120.0f)), 0)) {  // |y| >= 120, inf, nan: never on a sampler path
    sn = sinf_glibc(y);
    cs = cosf_glibc(y);
    return;
  }
  int n;
  const double xr = reduce_fast((double)y, &n);
  const double x2 = xr * xr;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
               c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
  // sine form (sin_poly, n even) on the unsigned reduced argument
  const double x3 = xr * x2;
  const double st1 = __builtin_fma(x2, s3, s2);
  const double x7 = x3 * x2;
  const double ss = __builtin_fma(x3, s1, xr);
  const float S = (float)__builtin_fma(x7, st1, ss);
  // cosine form (sin_poly, n odd) with the positive coefficient set
  const double x4 = x2 * x2;
  const double ct2 = __builtin_fma(x2, c4, c3);
  const double ct1 = __builtin_fma(x2, c1, c0);
  const double x6 = x4 * x2;
  const double cc = __builtin_fma(x4, c2, ct1);
  const float C = (float)__builtin_fma(x6, ct2, cc);
  const uint32_t sneg = (((n & 3) == 1) | ((n & 3) == 2)) ? 0x80000000u : 0u;  // sign = {1,-1,-1,1}[n&3]
  const uint32_t cneg = (n & 2) ? 0x80000000u : 0u;                             // csign
  const float Ss = asfloat(asuint(S) ^ sneg), Cs = asfloat(asuint(C) ^ cneg);
  const bool odd = (n & 1) != 0;
  float rs = odd ? Cs : Ss;
  float rc = odd ? Ss : Cs;
  const bool tiny = top < abstop12(0x1p-12f);  // sinf: return y, cosf: return 1
  sn = tiny ? y : rs;
  cs = tiny ? 1.0f : rc;
}


// ------------------------------------------------------------------------------------------------
// powf / logf exactly as the host libm computes them (glibc 2.35 e_powf.c / e_logf.c = the ARM
// "optimized routines" algorithms: a 16-entry table of (1/c, log c) per mantissa interval, a short
// polynomial in binary64, exp2 through a 32-entry table, ONE rounding to binary32 at the end).
// Needed because FrSchlick (scattering.h:91-93) and NodeBinary '^' go through psl::pow = std::pow and
// the BSSRDF free flight through log (bxdf.cpp:343): with the device's own powf / logf (within 1 ulp
// of glibc, not identical) films with microfacet lobes differ from the reference in a few pixels.
// Algorithm restated; the numeric tables are the published ones (sysdeps/ieee754/flt-32/e_powf_log2_data.c,
// e_logf_data.c, e_exp2f_data.c), checked against the copy inside the container's libm.so.6.  As for
// sinf/cosf, every a*b+c is a fused multiply-add because x86-64 glibc runs its -mfma build.
// tools/check_libm.cpp: 0 mismatches against the container's libm over every float x in [0, 1] with
// y = 5 (Schlick), 2^31 positive floats for logf and 10^9 random (x, y) pairs.
// The tables are constant arrays read with one (lane-indexed) load per lookup.  They were switch statements of immediates
// until round 3: the compiler hoisted every 64-bit immediate of the select chains out of the kernel's persistent loop --
// a hundred VGPRs of loop-invariant constants, ALL of the stage-queued kernel's register spills (98 of 108 spilled VGPRs
// in the 10 000-cone variant, 64 scratch reloads per powf call).
PINE_HD uint64_t asuint64(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}
PINE_HD double asdouble(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}
PINE_HD uint64_t exp2f_tab(unsigned i) {
  static constexpr uint64_t T[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
  };
  return T[i & 31u];
}
PINE_HD void powf_log2_tab(unsigned i, double& invc, double& logc) {
  static constexpr double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2},
      {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
      {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2},
      {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
      {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2},
      {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3},
      {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
      {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
      {0x1.0000000000000p+0, 0x0p+0},
      {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},
      {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
      {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},
      {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
      {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
      {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
  };
  invc = T[i & 15u][0];
  logc = T[i & 15u][1];
}
PINE_HD void logf_tab(unsigned i, double& invc, double& logc) {
  static constexpr double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2},
      {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
      {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2},
      {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
      {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3},
      {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4},
      {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
      {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
      {0x1.0000000000000p+0, 0x0p+0},
      {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},
      {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},
      {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
      {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
      {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
  };
  invc = T[i & 15u][0];
  logc = T[i & 15u][1];
}
PINE_HD double powf_log2_inline(uint32_t ix) {  // e_powf.c log2_inline; |relative error| < 2^-68
  const uint32_t tmp = ix - 0x3f330000u;
  const unsigned i = (tmp >> (23 - 4)) % 16;
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = int32_t(top) >> 23;  // arithmetic shift
  double invc, logc;
  powf_log2_tab(i, invc, logc);
  const double z = double(asfloat(iz));
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = logc + double(k);
  const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
               A4 = 0x1.71547652ab82bp0;
  const double r2 = r * r;
  double y = __builtin_fma(A0, r, A1);
  const double p = __builtin_fma(A2, r, A3);
  const double r4 = r2 * r2;
  double q = __builtin_fma(A4, r, y0);
  q = __builtin_fma(p, r2, q);
  y = __builtin_fma(y, r4, q);
  return y;
}
PINE_HD float powf_exp2_inline(double xd, uint32_t sign_bias) {  // e_powf.c exp2_inline (no TOINT intrinsics)
  const double shift = 0x1.8p+47;  // 0x1.8p52 / 32
  double kd = xd + shift;
  const uint64_t ki = asuint64(kd);
  kd -= shift;
  const double r = xd - kd;
  uint64_t t = exp2f_tab(unsigned(ki % 32));
  const uint64_t ski = ki + sign_bias;
  t += ski << (52 - 5);
  const double s = asdouble(t);
  const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
  const double z = __builtin_fma(C0, r, C1);
  const double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return float(y);
}
PINE_HD int powf_checkint(uint32_t iy) {  // 0: not an integer, 1: odd, 2: even
  const int e = int(iy >> 23 & 0xff);
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
PINE_HD bool powf_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }
PINE_HD float powf_glibc(float x, float y) {
  uint32_t sign_bias = 0;
  uint32_t ix = asuint(x);
  const uint32_t iy = asuint(y);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
    // x < 0x1p-126 or inf or nan, or y is 0 or inf or nan
    if (powf_zeroinfnan(iy)) {
      if (2 * iy == 0) return 1.0f;
      if (ix == 0x3f800000u) return 1.0f;
      if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
      if (2 * ix == 2 * 0x3f800000u) return 1.0f;
      if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;  // |x|<1 && y==inf or |x|>1 && y==-inf
      return y * y;
    }
    if (powf_zeroinfnan(ix)) {
      float x2 = x * x;
      if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
      return (iy & 0x80000000u) ? 1 / x2 : x2;
    }
    if (ix & 0x80000000u) {  // finite x < 0
      const int yint = powf_checkint(iy);
      if (yint == 0) return (x - x) / (x - x);  // NaN (__math_invalidf)
      if (yint == 1) sign_bias = 1u << (5 + 11);
      ix &= 0x7fffffffu;
    }
    if (ix < 0x00800000u) {  // subnormal x: normalise
      ix = asuint(x * 0x1p23f);
      ix &= 0x7fffffffu;
      ix -= 23u << 23;
    }
  }
  const double logx = powf_log2_inline(ix);
  const double ylogx = double(y) * logx;  // cannot overflow
  if ((asuint64(ylogx) >> 47 & 0xffff) >= asuint64(126.0) >> 47) {  // |y*log2(x)| >= 126
    if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
    if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
  }
  return powf_exp2_inline(ylogx, sign_bias);
}
PINE_HD float logf_glibc(float x) {  // e_logf.c
  uint32_t ix = asuint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return -__builtin_inff();          // log(+-0) = -inf
    if (ix == 0x7f800000u) return x;                    // log(inf) = inf
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);  // negative or nan
    ix = asuint(x * 0x1p23f);  // subnormal: normalise
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const unsigned i = (tmp >> (23 - 4)) % 16;
  const int k = int32_t(tmp) >> 23;
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  double invc, logc;
  logf_tab(i, invc, logc);
  const double z = double(asfloat(iz));
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma(double(k), 0x1.62e42fefa39efp-1, logc);
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  const double r2 = r * r;
  double y = __builtin_fma(A1, r, A2);
  y = __builtin_fma(A0, r2, y);
  y = __builtin_fma(y, r2, y0 + r);
  return float(y);
}

// ------------------------------------------------------------------------------------------------
// atanf / atan2f / acosf exactly as the host libm computes them (glibc 2.35 s_atanf.c, e_atan2f.c, e_acosf.c: the
// fdlibm single-precision routines -- argument reduction to one of four atan anchors plus an odd / even split
// polynomial; the rational approximation of asin for acos -- every operation rounded to binary32, no fused
// multiply-add: x86-64 glibc has no FMA variant of these three).  On the path: a Sphere's uv (geometry.cpp:72-121
// via cartesian_to_spherical), read by node graphs that use UV().  Algorithm restated; the coefficients are the
// published ones.  tools/check_libm_atan.cpp: every binary32 argument of atanf and acosf and 4 x 10^9 (y, x) pairs of
// atan2f (all special cases, exponent differences around +-60, random) against the container's libm: 0 mismatches.
// sqrtf is the correctly rounded one in both.
PINE_HD float atanf_glibc(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
              aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
              aT10 = 1.6285819933e-02f;
  const int32_t hx = int32_t(asuint(x));
  const int32_t ix = hx & 0x7fffffff;
  int id;
  float hi = 0.0f, lo = 0.0f;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;  // NaN
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {  // |x| < 0.4375
    if (ix < 0x31000000) return x;  // |x| < 2^-29
    id = -1;
  } else {
    x = __builtin_fabsf(x);
    if (ix < 0x3f980000) {    // |x| < 1.1875
      if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
        id = 0;
        x = (2.0f * x - 1.0f) / (2.0f + x);
      } else {  // 11/16 <= |x| < 19/16
        id = 1;
        x = (x - 1.0f) / (x + 1.0f);
      }
    } else {
      if (ix < 0x401c0000) {  // |x| < 2.4375
        id = 2;
        x = (x - 1.5f) / (1.0f + 1.5f * x);
      } else {  // 2.4375 <= |x| < 2^25
        id = 3;
        x = -1.0f / x;
      }
    }
    hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];
    lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
  }
  const float z = x * x;
  const float w = z * z;
  // the sum over aT[i] z^(i+1) split into its odd and even terms
  const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}
PINE_HD float atan2f_glibc(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int32_t hx = int32_t(asuint(x)), hy = int32_t(asuint(y));
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
  if (hx == 0x3f800000) return atanf_glibc(y);           // x = 1
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);     // 2 * sign(x) + sign(y)
  if (iy == 0) {                                         // y = 0
    if (m < 2) return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
  if (ix == 0x7f800000) {                                       // x = inf
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    }
    switch (m) {
      case 0: return 0.0f;
      case 1: return -0.0f;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y = inf
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;      // |y / x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0f;       // |y| / x < -2^60
  else z = atanf_glibc(__builtin_fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return asfloat(asuint(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}
PINE_HD float acosf_glibc(float x, float (*sqrt_rn)(float) = nullptr) {
  const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
  const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
              pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  (void)sqrt_rn;
  const int32_t hx = int32_t(asuint(x));
  const int32_t ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;  // |x| = 1
  if (ix > 0x3f800000) return (x - x) / (x - x);                      // |x| > 1: NaN
  if (ix < 0x3f000000) {                                              // |x| < 0.5
    if (ix <= 0x23000000) return pio2_hi + pio2_lo;                   // |x| < 2^-57
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - r * x));
  }
  if (hx < 0) {  // x < -0.5
    const float z = (1.0f + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = __builtin_sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  // x > 0.5
  const float z = (1.0f - x) * 0.5f;
  const float s = __builtin_sqrtf(z);
  const float df = asfloat(asuint(s) & 0xfffff000u);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return 2.0f * (df + w);
}

}  // namespace pine_libm
