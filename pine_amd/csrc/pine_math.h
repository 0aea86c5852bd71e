// pine_amd/csrc/pine_math.h -- float vector/matrix math shared by the host scene builder and the
// gfx950 kernels.  Everything is IEEE binary32, one rounding per operation, in the operand order of
// the reference (compile with -ffp-contract=off; hipcc's default correctly-rounded fp32 divide and
// sqrt are required).  Reference: src/pine/core/vecmath.h, src/psl/math.h, src/pine/core/math.h.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "pine_libm.h"

namespace pine_gpu {

constexpr float kPi = 3.14159265358979323846f;         // src/psl/math.h:16
constexpr float kEpsilon = 1.1920928955078125e-07f;     // src/pine/core/math.h:13
constexpr float kOneMinusEps = 0x1.fffffep-1f;          // :14
constexpr float kFloatMax = 3.40282346638528859812e+38f;  // :15

// psl::min / psl::max (src/psl/math.h:19-26): comparisons, not IEEE minNum/maxNum.
PINE_HD float pmin(float a, float b) { return a < b ? a : b; }
PINE_HD float pmax(float a, float b) { return a > b ? a : b; }
PINE_HD float sqr(float v) { return v * v; }
PINE_HD float pclamp(float v, float a, float b) { return pmin(pmax(v, a), b); }
PINE_HD float pabs(float v) { return fabsf(v); }
// Correctly rounded sqrt / reciprocal, short forms.  hipcc's IEEE expansions cost 16 (sqrtf) and 11
// (1.0f/x) VALU instructions because they also cover operands near the exponent limits.  For
// ordinary operands a shorter sequence returns the SAME correctly rounded result; that is not an
// argument but a measurement: tools/rcp_test.hip compares them with the expansions over all 2^32 bit
// patterns on gfx950 and finds zero mismatches for biased exponents in [26, 230] (sqrt: v_rsq_f32,
// s = x*y, h = y/2, one residual step) and [2, 252] (reciprocal: v_rcp_f32 + one Newton step).  The
// guards below are strictly inside those ranges; anything else (zero, denormal, huge, inf, NaN)
// redoes the operation with the full expansion in a rarely taken branch.
// (-DPINE_NO_SHORT_FORMS: plain expansions everywhere, for A/B runs.)
// PINE_FAST_MATH (pine_kernels_fast.hip only -- the declared-tolerance variants behind PINE_GPU_FLAG_FAST): the hardware's
// 1-ulp v_sqrt_f32 / v_rcp_f32 without refinement or range guards, the device's native sin / cos / pow / log instead of
// the glibc-exact binary64 restatements, and (by that file's compile flags) contracted multiply-adds and reciprocal
// division.  Never defined in the parity build.
PINE_HD float psqrt(float v) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PINE_FAST_MATH)
  return __builtin_amdgcn_sqrtf(v);
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(PINE_NO_SHORT_FORMS)
  const float y = __builtin_amdgcn_rsqf(v);
  const float s0 = v * y;
  const float h = 0.5f * y;
  const float r = __builtin_fmaf(-s0, s0, v);
  float s1 = __builtin_fmaf(r, h, s0);
  if (__builtin_expect(!(v >= 0x1p-96f && v <= 0x1p+96f), 0)) s1 = sqrtf(v);
  return s1;
#else
  return sqrtf(v);
#endif
}
PINE_HD float prcp(float v) {  // == 1.0f / v
#if defined(__HIP_DEVICE_COMPILE__) && defined(PINE_FAST_MATH)
  return __builtin_amdgcn_rcpf(v);
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(PINE_NO_SHORT_FORMS)
  const float r0 = __builtin_amdgcn_rcpf(v);
  const float e0 = __builtin_fmaf(-v, r0, 1.0f);
  float r1 = __builtin_fmaf(e0, r0, r0);
  const float a = fabsf(v);
  if (__builtin_expect(!(a >= 0x1p-120f && a <= 0x1p+120f), 0)) r1 = 1.0f / v;
  return r1;
#else
  return 1.0f / v;
#endif
}
// libm-exact sin/cos (see pine_libm.h); on the host these equal std::sin/std::cos of glibc.
#if defined(__HIP_DEVICE_COMPILE__) && defined(PINE_FAST_MATH)
PINE_HD float psin(float v) { return __sinf(v); }
PINE_HD float pcos(float v) { return __cosf(v); }
PINE_HD float ppow(float a, float b) { return __powf(a, b); }
PINE_HD float plog(float v) { return __logf(v); }
PINE_HD void psincos(float v, float& sn, float& cs) {
  sn = __sinf(v);
  cs = __cosf(v);
}
PINE_HD float patan2(float y, float x) { return atan2f(y, x); }
PINE_HD float pacos(float v) { return acosf(v); }
#else
PINE_HD float psin(float v) { return pine_libm::sinf_glibc(v); }
PINE_HD float pcos(float v) { return pine_libm::cosf_glibc(v); }
// libm-exact pow / log (pine_libm.h): psl::pow == std::pow (src/psl/math.h:201-202), psl::log == std::log
PINE_HD float ppow(float a, float b) { return pine_libm::powf_glibc(a, b); }
PINE_HD float plog(float v) { return pine_libm::logf_glibc(v); }
PINE_HD void psincos(float v, float& sn, float& cs) {
  pine_libm::sincosf_glibc(v, sn, cs);
}
// libm-exact atan2 / acos (pine_libm.h): psl::atan2 == std::atan2, psl::acos == std::acos (src/psl/math.h:333-334, 388-389)
PINE_HD float patan2(float y, float x) { return pine_libm::atan2f_glibc(y, x); }
PINE_HD float pacos(float v) { return pine_libm::acosf_glibc(v); }
#endif

// ---- division ------------------------------------------------------------------------------------
// `a / b` in device code is hipcc's IEEE-correct expansion (v_div_scale x2, v_rcp, 5 fma/mul, v_div_fmas,
// v_div_fixup).  A guarded short form (shared refined reciprocal + two residual steps, full expansion
// outside a safe exponent range) was built, verified bit for bit on 1e11 operand pairs, and measured
// SLOWER in the path kernel; it is not kept.  Plain division everywhere.

struct f2 {
  float x, y;
};
struct f3 {
  float x, y, z;
};
PINE_HD f2 mk2(float x, float y) { return f2{x, y}; }
PINE_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PINE_HD f3 mk3(float v) { return f3{v, v, v}; }
PINE_HD f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
PINE_HD float get(const f3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
PINE_HD void set(f3& v, int i, float s) {
  if (i == 0) v.x = s;
  else if (i == 1) v.y = s;
  else v.z = s;
}
PINE_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PINE_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PINE_HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PINE_HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
PINE_HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PINE_HD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
PINE_HD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PINE_HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
PINE_HD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
PINE_HD f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
PINE_HD f2 operator*(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
PINE_HD f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }
PINE_HD f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }
PINE_HD bool is_zero(f3 v) { return v.x == 0 && v.y == 0 && v.z == 0; }

PINE_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vecmath.h:751
PINE_HD float absdot(f3 a, f3 b) { return pabs(dot(a, b)); }                  // :775
PINE_HD f3 cross(f3 a, f3 b) {                                                 // :780
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PINE_HD float length_squared(f3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
PINE_HD float length(f3 v) { return psqrt(length_squared(v)); }
PINE_HD float distance(f3 a, f3 b) { return length(a - b); }
// (Tried: v / len with one shared refined reciprocal applied to the three components behind a
//  range guard.  Identical bits, 29 instead of 42 VALU instructions, and 3.5 % SLOWER in the path kernel:
//  the guard's compares and the rare-path branch cost more than the divisions they save.  Plain division.)
PINE_HD f3 div_by_length(f3 v, float d) { return v / d; }
PINE_HD f3 normalize(f3 v) {  // :736-741
  const float len = length(v);
  const f3 q = div_by_length(v, len);
  return len == 0 ? v : q;
}
PINE_HD f3 normalize(f3 v, float& len) {  // :742-747
  len = length(v);
  const f3 q = div_by_length(v, len);
  return len == 0 ? v : q;
}
PINE_HD f3 vmin(f3 a, f3 b) { return {pmin(a.x, b.x), pmin(a.y, b.y), pmin(a.z, b.z)}; }
PINE_HD f3 vmax(f3 a, f3 b) { return {pmax(a.x, b.x), pmax(a.y, b.y), pmax(a.z, b.z)}; }
PINE_HD f3 vabs(f3 a) { return {pabs(a.x), pabs(a.y), pabs(a.z)}; }
PINE_HD int max_axis(f3 v) {  // vecmath.h:1263-1268
  if (v.x > v.y) return v.x > v.z ? 0 : 2;
  return v.y > v.z ? 1 : 2;
}
// lerp(u, v, a, b, c) = (1 - u - v) * a + u * b + v * c   (vecmath.h:882-884)
PINE_HD f3 lerp3(float u, float v, f3 a, f3 b, f3 c) { return (1.0f - u - v) * a + u * b + v * c; }

struct m3 {  // column vectors
  f3 x, y, z;
};
PINE_HD f3 mul(const m3& m, f3 v) { return m.x * v.x + m.y * v.y + m.z * v.z; }  // vecmath.h:695
PINE_HD m3 transpose(const m3& m) {
  return {f3{m.x.x, m.y.x, m.z.x}, f3{m.x.y, m.y.y, m.z.y}, f3{m.x.z, m.y.z, m.z.z}};
}
PINE_HD void coordinate_system(f3 n, f3& t, f3& b) {  // vecmath.h:1182-1188
  // (the axis is selected first, so that the cross product + normalize run once: same operations on the same values)
  const bool along_y = pabs(n.x) > pabs(n.y);
  t = normalize(cross(n, f3{along_y ? 0.0f : 1.0f, along_y ? 1.0f : 0.0f, 0.0f}));
  b = cross(n, t);
}
PINE_HD m3 coordinate_system(f3 n) {  // :1190-1195
  m3 m;
  m.z = n;
  coordinate_system(n, m.x, m.y);
  return m;
}
PINE_HD m3 inverse(const m3& m) {  // vecmath.cpp:80-102 (m[c][r] = column c, row r)
  const float m00 = m.x.x, m01 = m.x.y, m02 = m.x.z;
  const float m10 = m.y.x, m11 = m.y.y, m12 = m.y.z;
  const float m20 = m.z.x, m21 = m.z.y, m22 = m.z.z;
  float det = m00 * (m11 * m22 - m21 * m12) + m10 * (m21 * m02 - m01 * m22) +
              m20 * (m01 * m12 - m11 * m02);
  m3 r{f3{1, 0, 0}, f3{0, 1, 0}, f3{0, 0, 1}};
  if (det == 0) return r;
  r.x.x = m11 * m22 - m21 * m12;
  r.x.y = m21 * m02 - m01 * m22;
  r.x.z = m01 * m12 - m11 * m02;
  r.y.x = m12 * m20 - m22 * m10;
  r.y.y = m22 * m00 - m02 * m20;
  r.y.z = m02 * m10 - m12 * m00;
  r.z.x = m10 * m21 - m20 * m11;
  r.z.y = m20 * m01 - m00 * m21;
  r.z.z = m00 * m11 - m10 * m01;
  r.x = r.x / det;
  r.y = r.y / det;
  r.z = r.z / det;
  return r;
}

// Affine 3x4 part of a mat4 (columns x,y,z,w; the reference's mat4*vec3 drops the 4th row,
// vecmath.h:705-707).  p -> m.x*p.x + m.y*p.y + m.z*p.z + m.w, evaluated left to right.
struct m34 {
  f3 x, y, z, w;
};
PINE_HD f3 mul_point(const m34& m, f3 v) { return m.x * v.x + m.y * v.y + m.z * v.z + m.w; }
PINE_HD m3 linear(const m34& m) { return {m.x, m.y, m.z}; }
PINE_HD m34 ld34(const float* p) { return {ld3(p), ld3(p + 3), ld3(p + 6), ld3(p + 9)}; }

PINE_HD f3 face_same_hemisphere(f3 v, f3 ref) { return dot(v, ref) < 0 ? -v : v; }  // :1220
PINE_HD float safe_rcp(float v) { return v == 0.0f ? 1e+20f : prcp(v); }            // :1062

PINE_HD float as_float(int32_t i) {
  float f;
  memcpy(&f, &i, 4);
  return f;
}
PINE_HD int32_t as_int(float f) {
  int32_t i;
  memcpy(&i, &f, 4);
  return i;
}

}  // namespace pine_gpu
