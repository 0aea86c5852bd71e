// pine_amd/csrc/pine_device.h -- device-side restatement of the reference's per-ray algorithms:
// sampler / RNG, ray spawn, shape hit / intersect / surface info / light sampling, BSDFs.
// Pure functions over the POD records of pine_types.h; no memory allocation, no recursion.
// The traversal loop and the path state machine live in pine_kernels.hip.
//
// Reference citations are paths relative to the reference repository (wicstas/pine).  Floating
// point: binary32, no contraction, operand order of the reference; sin/cos via pine_libm.h.
#pragma once
#include "pine_math.h"
#include "pine_types.h"

namespace pine_gpu {

// Kernel specialisation flags: the path kernel is compiled once per feature set actually present
// in a scene (shape kinds, material kinds), so a Rect/Box + Diffuse scene (cbox) does not carry the
// registers and code of cones, meshes, microfacet lobes or the BSSRDF walk.
enum : unsigned {
  F_AABB = 1u << 0,
  F_OBB = 1u << 1,
  F_SPHERE = 1u << 2,
  F_DISK = 1u << 3,
  F_CONE = 1u << 4,
  F_MESH = 1u << 5,
  F_UBER = 1u << 6,
  F_SSS = 1u << 7,
  F_LDS_SCENE = 1u << 8,  // scene records staged in LDS
  F_NODES = 1u << 9,      // material parameters may be shading-node programs (node.h)
  F_LIGHTS = 1u << 10,    // Point / Spot / Directional lights or an environment light are present
  F_XSHAPES = 1u << 11,   // Plane / Line / Cylinder / stand-alone Triangle
  F_SOBOL = 1u << 12,     // the sampler may be SobolSampler (DTables::kind) instead of BlueSampler
  F_LDS_TOP = 1u << 13,   // scene in global memory, but the first DeviceScene::lds_nodes BVH nodes (breadth-first numbering:
                          // the levels every ray visits) are staged in LDS, and the traversal stack holds 16-bit node ids
  F_LDS_REST = 1u << 14,  // with F_LDS_TOP: everything of the scene blob except the BVH nodes (shape / leaf / material / light records:
                          // small when the geometry count is, whatever the meshes' sizes) is staged in LDS as well
  F_XSTAGE = 1u << 15,    // with F_LDS_TOP (stage-queued kernel): traversal is a stage of its own (XS / XC queues, lanes refilled) instead
                          // of a loop inside stages S and T
  F_VLOG = 1u << 16,      // test hook (stage-queued kernel): the per-vertex log of WorkParams::vertex_log is compiled in -- two twin
                          // variants only (pine_variants.h), chosen by PINE_GPU_FLAG_VERTEX_LOG; every other variant carries no trace of it
  F_BAKED = 1u << 17,     // scene-specialised builds (pine_specialize.h): the scene's BVH and primitive records are baked into the kernel
                          // (a -DPINE_BAKED_SCENE compile; the bit only makes such a kernel's name its own in profiles)
  F_EMBREE = 1u << 18,    // PINE_GPU_FLAG_ORDER_EMBREE: closest-hit queries hand the shapes to their tests in the order of the reference's
                          // EmbreeAccel (scene_traverse_embree, pine_kernels_device.h) instead of in pine-BVH order;
                          // a few variants only (pine_variants.h), never chosen without the flag
  F_ALL = 0xffu | F_NODES | F_LIGHTS | F_XSHAPES | F_SOBOL,
};

// ------------------------------------------------------------------------------------------------
// hash + RNG (src/pine/core/rng.h:9-144) -- integer exact
// ------------------------------------------------------------------------------------------------
PINE_HD uint64_t hash_pixel(int px, int py, int sample_index) {
  // murmur_hash64A over the 12 bytes {px, py, sample_index}, seed 0 (rng.h:9-49, :60-65)
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  const int r = 47;
  uint64_t h = 0 ^ (12ull * m);
  uint64_t k = uint64_t(uint32_t(px)) | (uint64_t(uint32_t(py)) << 32);
  k *= m;
  k ^= k >> r;
  k *= m;
  h ^= k;
  h *= m;
  uint32_t t = uint32_t(sample_index);  // tail of 4 bytes: cases 4..1
  h ^= uint64_t((t >> 24) & 0xff) << 24;
  h ^= uint64_t((t >> 16) & 0xff) << 16;
  h ^= uint64_t((t >> 8) & 0xff) << 8;
  h ^= uint64_t(t & 0xff);
  h *= m;
  h ^= h >> r;
  h *= m;
  h ^= h >> r;
  return h;
}
struct DRng {
  uint64_t s0, s1;
};
PINE_HD uint64_t split_mix_64(uint64_t& s) {  // rng.h:72-77
  uint64_t r = s += 0x9E3779B97f4A7C15ULL;
  r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ULL;
  r = (r ^ (r >> 27)) * 0x94D049BB133111EBULL;
  return r ^ (r >> 31);
}
PINE_HD DRng rng_seed(uint64_t seed) {  // RNG::RNG rng.h:98-101
  DRng g;
  g.s0 = split_mix_64(seed);
  g.s1 = split_mix_64(seed);
  return g;
}
PINE_HD uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
PINE_HD uint64_t rng_next64(DRng& g) {  // rng.h:116-126
  const uint64_t s0 = g.s0;
  uint64_t s1 = g.s1;
  const uint64_t result = s0 + s1;
  s1 ^= s0;
  g.s0 = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
  g.s1 = rotl64(s1, 37);
  return result;
}
PINE_HD float rng_nextf(DRng& g) {  // rng.h:132-135
  uint64_t u = rng_next64(g);
  return pmin(float(uint32_t(u ^ (u >> 32))) * 0x1p-32f, kOneMinusEps);
}

// ------------------------------------------------------------------------------------------------
// BlueSobolSampler (src/pine/core/sampler.h:166-201, src/contrib/bluesobol/bluenoise_*spp.cpp:14-34)
// ------------------------------------------------------------------------------------------------
constexpr int kLdsSamplerDims = 40;     // sampler dimensions served from the LDS cache
constexpr int kLdsTileDwords = 12;      // per lane: 10 dwords of rank bytes (dims 0..39) + 2 of scramble bytes
constexpr int kLdsLaneStride = 256;     // = workgroup size: per-lane LDS slices are [dword][lane]

struct DTables {
  const uint8_t* sobol;     // [256 dims][256 samples]: TRANSPOSED relative to the published table
  const uint8_t* scramble;  // [128*128*8] of the selected spp variant
  const uint8_t* rank;      // [128*128*8] (+64 bytes of wrap-around padding on the device)
  // Optional workgroup-local cache (LDS): the first kLdsSamplerDims rows of the transposed Sobol
  // table, and for THIS lane's current pixel the 40 ranking bytes + 8 scrambling bytes the
  // lookup needs (refreshed whenever the lane takes a new work item).
  const uint8_t* lds_sobol;
  const uint32_t* lds_tile;  // ranking bytes: dword j (dims 4j..4j+3) at lds_tile[j * tile_stride]
  const uint32_t* lds_scr;   // scrambling bytes: dword k (dims 4k..4k+3 mod 8) at lds_scr[k * tile_stride]
  int tile_stride;           // = workgroup size: the slices are [dword][thread]
  int win_lo, win_len;       // dimensions [win_lo, win_lo + win_len) are present in the cache
  // SobolSampler instead (F_SOBOL variants only; sampler.h:83-164): nothing is read from the tables
  int kind;                  // 0 = BlueSampler, 1 = SobolSampler, 2 = HaltonSampler
  int sobol_log2_spp;        // psl::log2i(spp) (sampler.h:127-129)
  int sobol_digits;          // nbase4_digits (sampler.cpp:81-84)
  // HaltonSampler (sampler.h:40-81): the first kHaltonDims primes, their prefix sums, and the digit permutation of each
  // (HaltonSampler::radicalInversePermutations, derived on the host: pine_host.cpp)
  const int* halton_primes;  // [kHaltonDims] primes, then [kHaltonDims] prefix sums
  const uint16_t* halton_perms;
};
constexpr int kHaltonDims = 1000;  // PrimeTablesize (primes.h): HaltonSampler::get1d / get2d wrap the dimension to 2 there (sampler.h:52-63)
// value = sobol[dim + (index ^ rank[(dim + pix*8) % N]) * 256] ^ scramble[dim % 8 + pix*8]
// (bluenoise_*spp.cpp:14-34); LDS = true reads dims < kLdsSamplerDims from the workgroup cache.
template <bool LDS = false>
PINE_HD float blue_sample_dimension(const DTables& t, int px, int py, int index, int dim) {
  const int si = index & 255;
  const int sd = dim & 255;
  if constexpr (LDS) {
    if (unsigned(sd - t.win_lo) < unsigned(t.win_len)) {
      const unsigned sh = 8u * unsigned(sd & 3);
      const uint32_t rw = t.lds_tile[(sd >> 2) * t.tile_stride];
      const int ranked = si ^ int((rw >> sh) & 255u);
      int value = t.lds_sobol[sd * 256 + ranked];
      const uint32_t sw = t.lds_scr[((sd & 7) >> 2) * t.tile_stride];
      value = value ^ int((sw >> sh) & 255u);
      return (0.5f + float(value)) / 256.0f;
    }
  }
  const int pix = (px & 127) + (py & 127) * 128;
  const int ranked = si ^ int(t.rank[(sd + pix * 8) % (128 * 128 * 8)]);
  int value = t.sobol[sd * 256 + ranked];  // = sobol_256spp_256d[sd + ranked*256]
  value = value ^ int(t.scramble[(sd % 8) + pix * 8]);
  return (0.5f + float(value)) / 256.0f;
}
struct DSampler {
  int px, py;
  int index;      // sample index within the pixel
  int dimension;  // BlueSobolSampler::dimension
};
// ---- SobolSampler (sampler.h:83-164, sampler.cpp:81-113, lowdiscrepancy.h:73-80): integer exact ----
PINE_HD uint64_t mix_bits(uint64_t v) {  // rng.h:81-88
  v ^= (v >> 31);
  v *= 0x7fb5d329728ea185ull;
  v ^= (v >> 27);
  v *= 0x81dadef4bc2dd44dull;
  v ^= (v >> 33);
  return v;
}
PINE_HD uint32_t reverse_bits32(uint32_t x) { return __builtin_bitreverse32(x); }  // math.h:20-27 (one v_bfrev_b32)
PINE_HD uint64_t left_shift_64x2(uint64_t x) {  // vecmath.h:1231-1239
  x &= 0xffffffffull;
  x = (x ^ (x << 16)) & 0x0000ffff0000ffffull;
  x = (x ^ (x << 8)) & 0x00ff00ff00ff00ffull;
  x = (x ^ (x << 4)) & 0x0f0f0f0f0f0f0f0full;
  x = (x ^ (x << 2)) & 0x3333333333333333ull;
  x = (x ^ (x << 1)) & 0x5555555555555555ull;
  return x;
}
PINE_HD uint64_t hash_int(int v) {  // hash(dimension): murmur_hash64A over 4 bytes, seed 0 (rng.h:9-49, :60-65)
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  const int r = 47;
  uint64_t h = 0 ^ (4ull * m);
  const uint32_t t = uint32_t(v);
  h ^= uint64_t((t >> 24) & 0xff) << 24;
  h ^= uint64_t((t >> 16) & 0xff) << 16;
  h ^= uint64_t((t >> 8) & 0xff) << 8;
  h ^= uint64_t(t & 0xff);
  h *= m;
  h ^= h >> r;
  h *= m;
  h ^= h >> r;
  return h;
}
PINE_HD uint32_t fast_owen(uint32_t v, uint32_t seed) {  // FastOwenScrambler sampler.h:95-109
  v = reverse_bits32(v);
  v ^= v * 0x3d20adeau;
  v += seed;
  v *= (seed >> 16) | 1u;
  v ^= v * 0x05526c56u;
  v ^= v * 0x53a22864u;
  return reverse_bits32(v);
}
// sobol_sample (lowdiscrepancy.h:73-80) for the two dimensions SobolSampler reads.  Their generator
// matrices (sobolmatrices.cpp:40-, 52 columns each) have closed forms: dimension 0 is the bit reversal of
// the low 32 index bits (column i = 2^31 >> i, zero from i = 32); dimension 1's columns follow
// v[0] = 2^31, v[i+1] = v[i] ^ (v[i] >> 1) through all 52 (the oracle checks both against films rendered
// by the reference).
PINE_HD float sobol_sample01(uint64_t a, int dim, uint32_t seed) {
  uint32_t v = 0;
  if (dim == 0) {
    v = reverse_bits32(uint32_t(a));
  } else {
    uint32_t col = 0x80000000u;
    for (; a != 0; a >>= 1) {
      if (a & 1) v ^= col;
      col ^= col >> 1;
    }
  }
  v = fast_owen(v, seed);
  return pmin(float(v) * 0x1p-32f, kOneMinusEps);
}
PINE_HD uint64_t sobol_index_of(const DTables& t, int px, int py, int index) {  // sampler.h:135-142
  const uint64_t morton = (left_shift_64x2(uint32_t(py)) << 1) | left_shift_64x2(uint32_t(px));  // vecmath.h:1243-1245
  return (morton << t.sobol_log2_spp) + uint64_t(uint32_t(index));  // start_pixel ORs sample 0 in, start_next_sample increments
}
PINE_HD uint64_t sobol_compute_sample_index(const DTables& t, uint64_t sobol_index, int dimension) {  // sampler.cpp:86-113
  // the 24 permutations of {0,1,2,3} in the reference's order, one per byte-quadruple: digit d of
  // permutation p is (kPerm[p] >> (2 * d)) & 3
  constexpr uint8_t kPerm[24] = {
      0xe4, 0xb4, 0xd8, 0x78, 0x6c, 0x9c, 0xe1, 0xb1, 0xc9, 0x39, 0x2d, 0x8d,
      0xc6, 0x36, 0xd2, 0x72, 0x4e, 0x1e, 0x27, 0x87, 0x1b, 0x4b, 0x63, 0x93};
  uint64_t si = 0;
  const bool only_power_of_2 = (t.sobol_log2_spp & 1) != 0;
  const int last_digit = only_power_of_2 ? 1 : 0;
  const uint64_t dim_mix = uint64_t(0x55555555u * uint32_t(dimension));
  for (int i = t.sobol_digits - 1; i >= last_digit; --i) {
    const int digit_shift = 2 * i - (only_power_of_2 ? 1 : 0);
    const int digit = int((sobol_index >> digit_shift) & 3);
    const uint64_t higher_digits = sobol_index >> (digit_shift + 2);
    const int p = int((mix_bits(higher_digits ^ dim_mix) >> 24) % 24);
    si |= uint64_t((kPerm[p] >> (2 * digit)) & 3) << digit_shift;
  }
  if (only_power_of_2) {
    const int digit = int(sobol_index & 1);
    si |= uint64_t(digit ^ int(mix_bits((sobol_index >> 1) ^ dim_mix) & 1));
  }
  return si;
}

// HaltonSampler::start_pixel + start_next_sample (sampler.cpp:64-79, sampler.h:48-51): the index of sample `index` of
// pixel (px, py) in the Halton sequence -- the pixel's offset modulo 128 x 243 (from its coordinates' reversed base-2 /
// base-3 digits through the Chinese remainder theorem), plus 31 104 per sample.  baseScales = {128, 243},
// baseExponents = {7, 5}, multInverse = {multiplicativeInverse(128, 243), multiplicativeInverse(243, 128)} = {131, 59}.
PINE_HD uint32_t halton_index_of(int px, int py, int index) {
  const uint32_t pm0 = uint32_t(px) & 127u;          // psl::mod(p, 128), p >= 0
  const uint32_t pm1 = uint32_t(py) & 127u;
  uint32_t off0 = 0, off1 = 0;                       // inverse_radical_inverse (lowdiscrepancy.h:42-51)
  for (uint32_t v = pm0, i = 0; i < 7; i++, v >>= 1) off0 = off0 * 2u + (v & 1u);
  for (uint32_t v = pm1, i = 0; i < 5; i++, v /= 3u) off1 = off1 * 3u + v % 3u;
  uint32_t h = off0 * 243u * 59u + off1 * 128u * 131u;  // dimOffset * baseScales[1 - i] * multInverse[1 - i]
  h %= 31104u;
  return h + uint32_t(index) * 31104u;
}
// scrambled_radical_inverse (lowdiscrepancy.h:26-40); a < 2^27 on the device (4096 samples x 31 104), so the digit
// extraction runs in 32 bits; the reversed digits need 64
PINE_HD float halton_sample_dimension(const DTables& t, int dim, uint32_t a) {
  const int base = t.halton_primes[dim];
  const uint16_t* perm = t.halton_perms + t.halton_primes[kHaltonDims + dim];
  const float inv_base = 1.0f / float(base);
  float inv_base_n = 1.0f;
  uint64_t reversed = 0;
  while (a) {
    const uint32_t next = a / uint32_t(base);
    const uint32_t digit = a - next * uint32_t(base);
    reversed = reversed * uint64_t(base) + perm[digit];
    inv_base_n *= inv_base;
    a = next;
  }
  const float series = float(perm[0]) / (float(base) + 1.0f);
  return pmin((float(reversed) + series) * inv_base_n, kOneMinusEps);
}

// Sampler front: MODE bit 0 = the workgroup has the BlueSampler LDS cache, bit 1 = the variant also
// carries SobolSampler and HaltonSampler (selected at run time by DTables::kind).
constexpr int kSmLds = 1, kSmSobol = 2;
template <int MODE = 0>
PINE_HD float sampler_get1d(const DTables& t, DSampler& s) {  // sampler.h:183-187 / :143-148
  if constexpr (MODE & kSmSobol) {
    if (t.kind == 1) {
      const uint64_t si = sobol_compute_sample_index(t, sobol_index_of(t, s.px, s.py, s.index), s.dimension);
      s.dimension += 1;
      const uint64_t u = hash_int(s.dimension);
      return sobol_sample01(si, 0, uint32_t(u));
    }
    if (t.kind == 2) {  // sampler.h:52-56
      if (s.dimension >= kHaltonDims) s.dimension = 2;
      return halton_sample_dimension(t, s.dimension++, halton_index_of(s.px, s.py, s.index));
    }
  }
  if (s.dimension >= 256) s.dimension = 2;
  return blue_sample_dimension<(MODE & kSmLds) != 0>(t, s.px, s.py, s.index, s.dimension++);
}
template <int MODE = 0>
PINE_HD f2 sampler_get2d(const DTables& t, DSampler& s) {  // sampler.h:188-194 / :149-155
  if constexpr (MODE & kSmSobol) {
    if (t.kind == 1) {
      const uint64_t si = sobol_compute_sample_index(t, sobol_index_of(t, s.px, s.py, s.index), s.dimension);
      s.dimension += 2;
      const uint64_t u = hash_int(s.dimension);
      const float a = sobol_sample01(si, 0, uint32_t(u));
      const float b = sobol_sample01(si, 1, uint32_t(u >> 32));
      return f2{a, b};
    }
    if (t.kind == 2) {  // sampler.h:57-63
      const uint32_t hi = halton_index_of(s.px, s.py, s.index);
      if (s.dimension + 1 >= kHaltonDims) s.dimension = 2;
      const int dim = s.dimension;
      s.dimension += 2;
      const float a = halton_sample_dimension(t, dim, hi);
      const float b = halton_sample_dimension(t, dim + 1, hi);
      return f2{a, b};
    }
  }
  if (s.dimension + 1 >= 256) s.dimension = 2;
  const int dim = s.dimension;
  s.dimension += 2;
  const float a = blue_sample_dimension<(MODE & kSmLds) != 0>(t, s.px, s.py, s.index, dim);
  const float b = blue_sample_dimension<(MODE & kSmLds) != 0>(t, s.px, s.py, s.index, dim + 1);
  return f2{a, b};
}
// with_probability (sampler.h:317-324): consumes an RNG float only for prob strictly in (0,1)
PINE_HD bool with_probability(float prob, DRng& g) {
  if (prob == 0) return false;
  if (prob == 1) return true;
  return rng_nextf(g) < prob;
}

// ------------------------------------------------------------------------------------------------
// sampling.h:8-89
// ------------------------------------------------------------------------------------------------
PINE_HD f2 sample_disk_polar(f2 u) {
  const float r = psqrt(u.x);
  const float theta = 2 * kPi * u.y;
  float sn, cs;
  psincos(theta, sn, cs);
  return f2{r * cs, r * sn};
}
PINE_HD f2 sample_disk_concentric(f2 u) {
  u = f2{u.x * 2 - 1.0f, u.y * 2 - 1.0f};
  // (one division for both branches of sampling.h:30-36: (Pi/4 * u.y) / u.x  or  Pi/2 - Pi/4 * (u.x / u.y))
  const bool wide = pabs(u.x) > pabs(u.y);
  const float r = wide ? u.x : u.y;
  const float q = (wide ? kPi / 4.0f * u.y : u.x) / (wide ? u.x : u.y);
  const float theta = wide ? q : kPi / 2.0f - kPi / 4.0f * q;
  float sn, cs;
  psincos(theta, sn, cs);
  return r * f2{cs, sn};
}
PINE_HD f3 cosine_weighted_hemisphere(f2 u) {
#if defined(PINE_DUP_COSHEMI) && defined(__HIP_DEVICE_COMPILE__)  /* cost measurement only: the map is evaluated PINE_DUP_COSHEMI more times on an opaque copy (same film; the extra time is its cost) */
  for (int rep_ = 0; rep_ < PINE_DUP_COSHEMI; rep_++) {
    f2 uu = u;
    asm volatile("" : "+v"(uu.x), "+v"(uu.y));
    const f2 dd = sample_disk_concentric(uu);
    float zz = psqrt(pmax(1.0f - dd.x * dd.x - dd.y * dd.y, 0.0f)) + dd.x + dd.y;
    asm volatile("" : : "v"(zz));
  }
#endif
  const f2 d = sample_disk_concentric(u);
  const float z = psqrt(pmax(1.0f - d.x * d.x - d.y * d.y, 0.0f));
  return f3{d.x, d.y, z};
}
PINE_HD f3 uniform_sphere(f2 u) {
  const float phi = u.x * kPi * 2;
  const float cos_theta = 1 - 2 * u.y;
  const float sin_theta = psqrt(1.0f - sqr(cos_theta));
  float sn, cs;
  psincos(phi, sn, cs);
  return f3{sin_theta * cs, sin_theta * sn, cos_theta};
}
PINE_HD float balance_heuristic(float pF, float pG) { return pF / (pF + pG); }

// ------------------------------------------------------------------------------------------------
// Ray, spawn_ray (src/pine/core/ray.h:8-46, src/pine/core/interaction.cpp:6-13)
// ------------------------------------------------------------------------------------------------
struct DRay {
  f3 o, d;
  float tmin, tmax;
};
PINE_HD f3 ray_at(const DRay& r, float t) { return r.o + t * r.d; }
PINE_HD f3 offset_ray_origin(f3 p, f3 n) {  // ray.h:25-37: integer ULP stepping on the float bits
  const float origin = 1.0f / 32.0f;
  const float float_scale = 1.0f / 65536.0f;
  const float int_scale = 256.0f;
  const int ox = int(int_scale * n.x), oy = int(int_scale * n.y), oz = int(int_scale * n.z);
  const float pix = as_float(as_int(p.x) + (p.x < 0 ? -ox : ox));
  const float piy = as_float(as_int(p.y) + (p.y < 0 ? -oy : oy));
  const float piz = as_float(as_int(p.z) + (p.z < 0 ? -oz : oz));
  return f3{pabs(p.x) < origin ? p.x + n.x * float_scale : pix,
            pabs(p.y) < origin ? p.y + n.y * float_scale : piy,
            pabs(p.z) < origin ? p.z + n.z * float_scale : piz};
}
// SurfaceInteraction::spawn_ray (interaction.cpp:6-13)
PINE_HD DRay spawn_ray(f3 p, f3 n, f3 wo, float tmax) {
  DRay r;
  r.d = wo;
  r.o = offset_ray_origin(p, face_same_hemisphere(n, wo));
  r.tmin = 0.0f;
  r.tmax = tmax * (1.0f - 1e-3f);
  return r;
}
// free spawn_ray(p, n, wo) (ray.h:39-46) -- no hemisphere flip; used by the BSSRDF walk
PINE_HD DRay spawn_ray_raw(f3 p, f3 n, f3 wo) {
  DRay r;
  r.o = offset_ray_origin(p, n);
  r.d = wo;
  r.tmin = 0.0f;
  r.tmax = kFloatMax * (1.0f - 1e-3f);
  return r;
}

// ------------------------------------------------------------------------------------------------
// AABB slab tests
// ------------------------------------------------------------------------------------------------
struct DRayOct {  // RayOctant bbox.h:18-27
  f3 dir_inv, org_div_dir;
  int neg;  // bit i set when d[i] < 0
};
PINE_HD DRayOct make_oct(const DRay& r) {
  DRayOct o;
  o.dir_inv = f3{safe_rcp(r.d.x), safe_rcp(r.d.y), safe_rcp(r.d.z)};
  o.org_div_dir = r.o * o.dir_inv;
  o.neg = (r.d.x < 0 ? 1 : 0) | (r.d.y < 0 ? 2 : 0) | (r.d.z < 0 ? 4 : 0);
  return o;
}
// AABB::hit(RayOctant, tmin, tmax&) bbox.h:59-72
PINE_HD bool box_hit_oct(const float* lo, const float* hi, const DRayOct& r, float tmin, float& tmax) {
  const bool nx = r.neg & 1, ny = r.neg & 2, nz = r.neg & 4;
  const float tmin0 = (nx ? hi[0] : lo[0]) * r.dir_inv.x - r.org_div_dir.x;
  const float tmin1 = (ny ? hi[1] : lo[1]) * r.dir_inv.y - r.org_div_dir.y;
  const float tmin2 = (nz ? hi[2] : lo[2]) * r.dir_inv.z - r.org_div_dir.z;
  const float tmax0 = (nx ? lo[0] : hi[0]) * r.dir_inv.x - r.org_div_dir.x;
  const float tmax1 = (ny ? lo[1] : hi[1]) * r.dir_inv.y - r.org_div_dir.y;
  const float tmax2 = (nz ? lo[2] : hi[2]) * r.dir_inv.z - r.org_div_dir.z;
  // psl::max(a,b,c,d) src/psl/math.h:52-61 is the chain a > x ? a : x from the right.  Its right-hand
  // operand is never NaN here (it starts from the ray's tmin / tmax), and a NaN on the left selects the
  // right-hand operand: exactly the "ignore NaN" maximum the hardware's v_max3_f32 / v_min3_f32 compute.
  // Only the sign of a zero result can differ, and the result is only ever compared.
  tmin = __builtin_fmaxf(__builtin_fmaxf(tmin0, __builtin_fmaxf(tmin1, tmin2)), tmin);
  tmax = __builtin_fminf(__builtin_fminf(tmax0, __builtin_fminf(tmax1, tmax2)), tmax);
  return tmin <= tmax;
}
// AABB::intersect(o, d, tmin&, tmax&) bbox.cpp:94-111 (also the body of AABB::hit(Ray) :75-93)
// Straight-line form (selects instead of the reference's early returns): in a wave some lane almost
// always needs every step, so the early exits only added exec-mask bookkeeping and branch latency.
// The decisions are the same comparisons on the same values; after a failed axis tmin / tmax are
// dead in every caller (they return false at once).
PINE_HD bool box_slabs(f3 lo, f3 hi, f3 o, f3 d, float& tmin, float& tmax) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float di = get(d, i), oi = get(o, i), l = get(lo, i), h = get(hi, i);
    const bool parallel = pabs(di) < 1e-6f;
    const float inv_d = prcp(di);  // (unused when parallel)
    float t_near = (l - oi) * inv_d;
    float t_far = (h - oi) * inv_d;
    const bool neg = inv_d < 0.0f;
    const float tn = neg ? t_far : t_near;
    const float tf = neg ? t_near : t_far;
    const float ntmin = pmax(tn, tmin);
    const float ntmax = pmin(tf, tmax);
    const bool axis_ok = parallel ? !(oi < l || oi > h) : !(ntmin > ntmax);
    ok = ok & axis_ok;
    tmin = parallel ? tmin : ntmin;
    tmax = parallel ? tmax : ntmax;
  }
  return ok;
}

// ------------------------------------------------------------------------------------------------
// Shapes: any-hit, closest-hit (shrinks ray.tmax), surface info.  S is the 128-byte record.
// ------------------------------------------------------------------------------------------------
struct DSurface {
  f3 p, n;
  f2 uv;
};

PINE_HD bool intersect_quadratic(float a, float b, float c, float tmin, float& tmax) {  // geometry.cpp:20-29
  float d = b * b - 4 * a * c;
  if (d <= 0.0f) return false;
  d = psqrt(d);
  float t = (-b - d) / (2 * a);
  if (t < tmin) t += d / a;
  if (t < tmin || t > tmax) return false;
  tmax = t;
  return true;
}
PINE_HD float sphere_compute_t(f3 ro, f3 rd, float tmin, f3 p, float r) {  // geometry.cpp:73-83
  const f3 ro_p = ro - p;
  const float b = dot(ro_p, rd);
  const float c = dot(ro_p, ro_p) - r * r;
  float d = b * b - c;
  if (d <= 0.0f) return -1.0f;
  d = psqrt(d);
  float t = -b - d;
  if (t < tmin) t = -b + d;
  return t;
}

// Rect::hit / Rect::intersect share the plane + extent test (geometry.cpp:275-299)
PINE_HD bool rect_test(const float* f, const DRay& ray, float& t_out) {
  // Straight-line form of the reference's chain of early returns (same comparisons, same values,
  // combined with & instead of branches): with 64 rays per wave some lane needs every step, so the
  // exits only cost exec-mask bookkeeping and seven dependent branch points per test.
  const f3 position = ld3(f), n = ld3(f + 9);
  const float denom = dot(ray.d, n);
  const float num = dot(position - ray.o, n);
  bool ok = !(denom == 0.0f);
  const float t = num / denom;
  ok = ok & !(t <= ray.tmin || t >= ray.tmax);
  const f3 p = ray_at(ray, t) - position;
  const float u = dot(p, ld3(f + 14));
  ok = ok & !(u < -0.5f || u > 0.5f);
  const float v = dot(p, ld3(f + 17));
  ok = ok & !(v < -0.5f || v > 0.5f);
  t_out = t;
  return ok;
}
// OBB: ray into object space with a re-normalised direction but WORLD tmin/tmax (bbox.cpp:145-163)
PINE_HD void obb_local_ray(const float* f, f3 o, f3 d, f3& lo_o, f3& lo_d) {
  const m34 mi = ld34(f + 18);
  lo_o = mul_point(mi, o);
  lo_d = normalize(mul(linear(mi), d));
}
PINE_HD bool cone_quadratic(const float* f, const DRay& ray, float& tmax, float& side) {  // geometry.cpp:415-434
  const f3 p = ld3(f), n = ld3(f + 3);
  const float A2 = f[9];
  const f3 o = ray.o - p;
  const f3 d = ray.d;
  const float a = -A2 * sqr(dot(d, n)) + dot(d, d);
  const float b = 2 * (-A2 * dot(o, n) * dot(d, n) + dot(o, d));
  const float c = -A2 * sqr(dot(o, n)) + dot(o, o);
  tmax = ray.tmax;
  if (!intersect_quadratic(a, b, c, ray.tmin, tmax)) return false;
  side = dot(o + tmax * d, n);
  return true;
}


PINE_HD bool tri_hit(const float* v, const DRay& ray);
PINE_HD bool tri_intersect(const float* v, DRay& ray);
// ---- Plane / Line / Cylinder (geometry.cpp:31-70, 171-244, 466-523) ---------------------------
// inverse(mat4) vecmath.cpp:103-132 on column-major m[c][r]; only rows 0..2 of the result are produced
// (the callers transform points).  Same expression order as the reference, nothing simplified.
PINE_HD bool mat4_inverse_rows3(const float (&m)[4][4], float (&r)[4][3]) {
  float det = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    det += (m[(1 + i) % 4][0] * (m[(2 + i) % 4][1] * m[(3 + i) % 4][2] - m[(3 + i) % 4][1] * m[(2 + i) % 4][2]) +
            m[(2 + i) % 4][0] * (m[(3 + i) % 4][1] * m[(1 + i) % 4][2] - m[(1 + i) % 4][1] * m[(3 + i) % 4][2]) +
            m[(3 + i) % 4][0] * (m[(1 + i) % 4][1] * m[(2 + i) % 4][2] - m[(2 + i) % 4][1] * m[(1 + i) % 4][2])) *
           m[i % 4][3] * float(i % 2 ? -1 : 1);
  if (det == 0) return false;
#pragma unroll
  for (int v = 0; v < 4; v++)
#pragma unroll
    for (int i = 0; i < 3; i++)
      r[v][i] = (m[(1 + i) % 4][(1 + v) % 4] * (m[(2 + i) % 4][(2 + v) % 4] * m[(3 + i) % 4][(3 + v) % 4] -
                                                 m[(3 + i) % 4][(2 + v) % 4] * m[(2 + i) % 4][(3 + v) % 4]) +
                 m[(2 + i) % 4][(1 + v) % 4] * (m[(3 + i) % 4][(2 + v) % 4] * m[(1 + i) % 4][(3 + v) % 4] -
                                                 m[(1 + i) % 4][(2 + v) % 4] * m[(3 + i) % 4][(3 + v) % 4]) +
                 m[(3 + i) % 4][(1 + v) % 4] * (m[(1 + i) % 4][(2 + v) % 4] * m[(2 + i) % 4][(3 + v) % 4] -
                                                 m[(2 + i) % 4][(2 + v) % 4] * m[(1 + i) % 4][(3 + v) % 4])) *
                float((v + i) % 2 ? 1 : -1) / det;
  return true;
}
// Line::hit / intersect share everything up to the acceptance test (geometry.cpp:181-192, 197-208):
// the segment in the ray's look_at frame, its closest approach to the z axis, clamped to the ray span
PINE_HD bool line_closest(const float* f, const DRay& ray, float& z) {
  const f3 up = mk3(0.0f, 1.0f, 0.0f);
  f3 zz = normalize((ray.o + ray.d) - ray.o);  // look_at(ray.o, ray.o + ray.d) vecmath.h:1172-1180
  if (pabs(dot(zz, up)) > 0.999f) zz = normalize(zz + mk3(0.0f, 0.0f, 1e-5f));
  const f3 xx = normalize(cross(up, zz));
  const f3 yy = cross(zz, xx);
  const float m[4][4] = {{xx.x, xx.y, xx.z, 0.0f}, {yy.x, yy.y, yy.z, 0.0f}, {zz.x, zz.y, zz.z, 0.0f},
                         {ray.o.x, ray.o.y, ray.o.z, 1.0f}};
  float r[4][3];
  if (!mat4_inverse_rows3(m, r)) {  // inverse() returns the identity for a singular matrix
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int k = 0; k < 3; k++) r[c][k] = c == k ? 1.0f : 0.0f;
  }
  const f3 cx = mk3(r[0][0], r[0][1], r[0][2]), cy = mk3(r[1][0], r[1][1], r[1][2]),
           cz = mk3(r[2][0], r[2][1], r[2][2]), cw = mk3(r[3][0], r[3][1], r[3][2]);
  const f3 a = ld3(f), b = ld3(f + 3);
  const f3 o = cx * a.x + cy * a.y + cz * a.z + cw;  // mat4 * vec3 vecmath.h:705
  const f3 q1 = cx * b.x + cy * b.y + cz * b.z + cw;
  const f3 d = q1 - o;
  // inverse(mat2(dot(d,d), -d.z, -d.z, 1)) * vec2(-dot(o,d), o.z): only .x is read
  const float m00 = dot(d, d), m01 = -d.z, m10 = -d.z, m11 = 1.0f;
  const float det2 = m00 * m11 - m10 * m01;
  const float ix0 = m11 / det2, iy0 = -m10 / det2;
  const float b0 = -dot(o, d), b1 = o.z;
  const float t = pclamp(ix0 * b0 + iy0 * b1, 0.0f, 1.0f);
  const float th = f[15];
  z = pclamp(o.z + t * d.z, ray.tmin + th, ray.tmax);
  const float D = length(o + t * d - mk3(0.0f, 0.0f, z));
  return D <= th;
}
// Cylinder::hit == the acceptance part of Cylinder::intersect (geometry.cpp:466-511): side surface only
PINE_HD bool cylinder_solve(const float* f, const DRay& ray, float& t) {
  const f3 p0 = ld3(f), p1 = ld3(f + 3), n = ld3(f + 6);
  const float r = f[9];
  const f3 m = ray.o - p0;
  const f3 v = ray.d - dot(ray.d, n) * n;
  const f3 w = m - dot(m, n) * n;
  const float a = dot(v, v);
  const float b = 2 * dot(v, w);
  const float c = dot(w, w) - r * r;
  const float discriminant = b * b - 4 * a * c;
  if (discriminant < 0) return false;
  const float sqrtDisc = psqrt(discriminant);
  t = (-b - sqrtDisc) / (2 * a);
  if (t < ray.tmin) t = (-b + sqrtDisc) / (2 * a);
  if (t > ray.tmax) return false;
  const f3 hit_point = ray_at(ray, t);
  const f3 projection = p0 + dot(hit_point - p0, n) * n;
  if (dot(projection - p0, n) < 0 || dot(projection - p1, n) > 0) return false;
  return true;
}

template <unsigned F = F_ALL>
PINE_HD bool shape_hit(int kind, const DShape* S, const DRay& ray) {
  const float* f = S->f;
  switch (kind) {
    case SHAPE_RECT: {
      float t;
      return rect_test(f, ray, t);
    }
    case SHAPE_AABB: if constexpr (!(F & F_AABB)) __builtin_unreachable(); else {  // AABB::hit(Ray) bbox.cpp:75-93
      float tmin = ray.tmin, tmax = ray.tmax;
      if (tmin > tmax) return false;
      return box_slabs(ld3(f), ld3(f + 3), ray.o, ray.d, tmin, tmax);
    }
    case SHAPE_OBB: if constexpr (!(F & F_OBB)) __builtin_unreachable(); else {  // OBB::hit bbox.cpp:145-149
      f3 o, d;
      obb_local_ray(f, ray.o, ray.d, o, d);
      float tmin = ray.tmin, tmax = ray.tmax;
      if (tmin > tmax) return false;
      return box_slabs(ld3(f), ld3(f + 3), o, d, tmin, tmax);
    }
    case SHAPE_SPHERE: if constexpr (!(F & F_SPHERE)) __builtin_unreachable(); else {  // geometry.cpp:84-87
      const float t = sphere_compute_t(ray.o, ray.d, ray.tmin, ld3(f), f[3]);
      return t > ray.tmin && t < ray.tmax;
    }
    case SHAPE_DISK: if constexpr (!(F & F_DISK)) __builtin_unreachable(); else {  // geometry.cpp:128-137
      const f3 position = ld3(f), n = ld3(f + 3);
      const float denom = dot(ray.d, n);
      if (denom == 0.0f) return false;
      const float t = (dot(position, n) - dot(ray.o, n)) / denom;
      if (t < ray.tmin) return false;
      if (t >= ray.tmax) return false;
      const f3 p = ray_at(ray, t) - position;
      if (length_squared(p) > sqr(f[12])) return false;
      return true;
    }
    case SHAPE_CONE: if constexpr (!(F & F_CONE)) __builtin_unreachable(); else {  // geometry.cpp:415-427
      float tmax, side;
      return cone_quadratic(f, ray, tmax, side) && side <= 0;
    }
    case SHAPE_PLANE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:35-39
      const f3 position = ld3(f), n = ld3(f + 3);
      const float t = (dot(position, n) - dot(ray.o, n)) / dot(ray.d, n);
      if (t <= ray.tmin) return false;
      return t < ray.tmax;
    }
    case SHAPE_LINE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:180-195
      float z;
      return line_closest(f, ray, z);
    }
    case SHAPE_CYLINDER: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:466-489
      float t;
      return cylinder_solve(f, ray, t);
    }
    case SHAPE_TRIANGLE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else { return tri_hit(f, ray); }  // :566
    default: return false;
  }
}

template <unsigned F = F_ALL>
PINE_HD bool shape_hit(const DShape* S, const DRay& ray) {
  return shape_hit<F>(S->kind, S, ray);
}

template <unsigned F = F_ALL>
PINE_HD bool shape_intersect(int kind, const DShape* S, DRay& ray) {
  const float* f = S->f;
  switch (kind) {
    case SHAPE_RECT: {
      float t;
      if (!rect_test(f, ray, t)) return false;
      ray.tmax = t;
      return true;
    }
    case SHAPE_AABB: if constexpr (!(F & F_AABB)) __builtin_unreachable(); else {  // bbox.cpp:112-121
      float tmin = ray.tmin, tmax = ray.tmax;
      if (!box_slabs(ld3(f), ld3(f + 3), ray.o, ray.d, tmin, tmax)) return false;
      ray.tmax = tmin > ray.tmin ? tmin : tmax;
      return true;
    }
    case SHAPE_OBB: if constexpr (!(F & F_OBB)) __builtin_unreachable(); else {  // bbox.cpp:150-172: endpoints mapped back to world distances
      f3 o, d;
      obb_local_ray(f, ray.o, ray.d, o, d);
      float tmin = ray.tmin, tmax = ray.tmax;
      if (!box_slabs(ld3(f), ld3(f + 3), o, d, tmin, tmax)) return false;
      const m34 m = ld34(f + 6);
      const f3 ps = o + tmin * d;
      const f3 pe = o + tmax * d;
      // bbox.cpp:166-171 maps both endpoints back; the far one is only read when the near distance is not
      // beyond tmin (origin on or inside the box): compute it on demand, the values are the same
      float tw = distance(mul_point(m, ps), ray.o);
      if (__builtin_expect(!(tw > ray.tmin), 0)) tw = distance(mul_point(m, pe), ray.o);
      ray.tmax = tw;
      return true;
    }
    case SHAPE_SPHERE: if constexpr (!(F & F_SPHERE)) __builtin_unreachable(); else {  // geometry.cpp:88-93
      const float t = sphere_compute_t(ray.o, ray.d, ray.tmin, ld3(f), f[3]);
      if (t < ray.tmin || t > ray.tmax) return false;
      ray.tmax = t;
      return true;
    }
    case SHAPE_DISK: if constexpr (!(F & F_DISK)) __builtin_unreachable(); else {  // geometry.cpp:138-148
      const f3 position = ld3(f), n = ld3(f + 3);
      const float denom = dot(ray.d, n);
      if (denom == 0.0f) return false;
      const float t = (dot(position, n) - dot(ray.o, n)) / denom;
      if (t < ray.tmin || t > ray.tmax) return false;
      const f3 p = ray_at(ray, t) - position;
      if (length_squared(p) > sqr(f[12])) return false;
      ray.tmax = t;
      return true;
    }
    case SHAPE_CONE: if constexpr (!(F & F_CONE)) __builtin_unreachable(); else {  // geometry.cpp:428-454
      float tmax, side;
      if (cone_quadratic(f, ray, tmax, side) && side < 0) {
        ray.tmax = tmax;
        return true;
      }
      return false;
    }
    case SHAPE_PLANE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:40-45
      const f3 position = ld3(f), n = ld3(f + 3);
      const float t = (dot(position, n) - dot(ray.o, n)) / dot(ray.d, n);
      if (t < ray.tmin || t > ray.tmax) return false;
      ray.tmax = t;
      return true;
    }
    case SHAPE_LINE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:196-214
      float z;
      if (!line_closest(f, ray, z)) return false;
      ray.tmax = z;
      return true;
    }
    case SHAPE_CYLINDER: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:490-517
      float t;
      if (!cylinder_solve(f, ray, t)) return false;
      ray.tmax = t;
      return true;
    }
    case SHAPE_TRIANGLE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else { return tri_intersect(f, ray); }  // :567
    default: return false;
  }
}

template <unsigned F = F_ALL>
PINE_HD bool shape_intersect(const DShape* S, DRay& ray) {
  return shape_intersect<F>(S->kind, S, ray);
}

// AABB::compute_surface_info bbox.cpp:122-129
PINE_HD void aabb_surface_info(f3 lo, f3 hi, f3 p, DSurface& it) {
  it.p = p;
  const f3 pu = (p - (lo + hi) / 2.0f) / (hi - lo);
  const int axis = max_axis(vabs(pu));
  it.n = mk3(0.0f);
  const bool pos = get(pu, axis) > 0;
  set(it.n, axis, pos ? 1.0f : -1.0f);
  set(it.p, axis, pos ? get(hi, axis) : get(lo, axis));
}

template <unsigned F = F_ALL>
PINE_HD void shape_surface_info(const DShape* S, f3 p, DSurface& it) {
  const float* f = S->f;
  it.uv = f2{0, 0};
  switch (S->kind) {
    case SHAPE_RECT: {  // geometry.cpp:300-307
      const f3 position = ld3(f);
      const f3 rp = p - position;
      const float u = dot(rp, ld3(f + 14));
      const float v = dot(rp, ld3(f + 17));
      it.p = position + f[12] * ld3(f + 3) * u + f[13] * ld3(f + 6) * v;
      it.n = ld3(f + 9);
      it.uv = f2{u, v} + f2{0.5f, 0.5f};
      break;
    }
    case SHAPE_AABB: if constexpr (!(F & F_AABB)) __builtin_unreachable(); else { aabb_surface_info(ld3(f), ld3(f + 3), p, it); } break;
    case SHAPE_OBB: if constexpr (!(F & F_OBB)) __builtin_unreachable(); else {  // bbox.cpp:173-177
      const m34 mi = ld34(f + 18);
      aabb_surface_info(ld3(f), ld3(f + 3), mul_point(mi, p), it);
      it.p = mul_point(ld34(f + 6), it.p);
      it.n = normalize(mul(transpose(linear(mi)), it.n));
      break;
    }
    case SHAPE_SPHERE: if constexpr (!(F & F_SPHERE)) __builtin_unreachable(); else {  // geometry.cpp:94-98
      const f3 c = ld3(f);
      it.n = normalize(p - c);
      it.p = c + it.n * f[3];
      // uv = cartesian_to_spherical(n) (vecmath.h:1209-1215): only shading nodes read it; glibc's atan2f / acosf restated (pine_libm.h)
      float phi = patan2(it.n.y, it.n.x);
      phi = phi < 0.0f ? kPi * 2 + phi : phi;
      it.uv = f2{phi, pacos(it.n.z)};
      break;
    }
    case SHAPE_DISK: if constexpr (!(F & F_DISK)) __builtin_unreachable(); else {  // geometry.cpp:149-155
      const f3 position = ld3(f), u = ld3(f + 6), v = ld3(f + 9);
      it.n = ld3(f + 3);
      const float ex = dot(p - position, u);
      const float ey = dot(p - position, v);
      it.uv = f2{ex, ey};
      it.p = position + ex * u + ey * v;
      break;
    }
    case SHAPE_CONE: if constexpr (!(F & F_CONE)) __builtin_unreachable(); else {  // geometry.cpp:455-460
      const f3 apex = ld3(f), n = ld3(f + 3);
      const float l = length(p - apex) * f[8];
      const f3 x = apex - n * l;
      it.n = normalize(p - x);
      it.p = x + it.n * l * f[10];
      break;
    }
    case SHAPE_PLANE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:46-51
      const f3 position = ld3(f), u = ld3(f + 6), v = ld3(f + 9);
      it.n = ld3(f + 3);
      const f3 dp = p - position;
      it.uv = f2{dot(dp, u), dot(dp, v)};
      it.p = position + it.uv.x * u + it.uv.y * v;
      break;
    }
    case SHAPE_LINE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:215-222
      const f3 p0 = ld3(f), p1 = ld3(f + 3);
      const float lt = dot(p - p0, ld3(f + 12));
      const f3 lp = lt * p1 + (1.0f - lt) * p0;  // lerp(lt, p0, p1) vecmath.h:877-880
      it.p = p;
      it.n = normalize(p - lp);
      it.uv = f2{lt, 0.0f};
      break;
    }
    case SHAPE_CYLINDER: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {
      // Cylinder::intersect itself leaves it.n / it.p (geometry.cpp:512-515; compute_surface_info is empty):
      // the winner's intersect is the last that succeeded, so recomputing from the final ray gives its values
      const f3 p0 = ld3(f), n = ld3(f + 6);
      const f3 projection = p0 + dot(p - p0, n) * n;
      it.n = normalize(p - projection);
      it.p = p;
      break;
    }
    case SHAPE_TRIANGLE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:568-574
      const f3 v0 = ld3(f), v1 = ld3(f + 3), v2 = ld3(f + 6);
      const float u = dot(p - v0, v1 - v0);
      const float v = dot(p - v0, v2 - v0);
      it.uv = f2{u, v};
      it.p = lerp3(u, v, v0, v1, v2);
      it.n = ld3(f + 9);
      break;
    }
    default: it.p = p; it.n = mk3(0.0f); break;
  }
}

// Triangle tests on 9 floats v0,v1,v2 (geometry.cpp:532-565)
PINE_HD bool tri_hit(const float* v, const DRay& ray) {
  const f3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
  const f3 E1 = v1 - v0, E2 = v2 - v0, T = ray.o - v0;
  const f3 P = cross(ray.d, E2), Q = cross(T, E1);
  const float D = dot(P, E1);
  if (D == 0.0f) return false;
  const float t = dot(Q, E2) / D;
  if (t < ray.tmin || t > ray.tmax) return false;
  const float u = dot(P, T) / D;
  if (u < 0.0f || u > 1.0f) return false;
  const float w = dot(Q, ray.d) / D;
  if (w < 0.0f || w > 1.0f) return false;
  return u + w < 1.0f;
}
PINE_HD bool tri_intersect(const float* v, DRay& ray) {
  const f3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
  const f3 E1 = v1 - v0, E2 = v2 - v0, T = ray.o - v0;
  const f3 P = cross(ray.d, E2), Q = cross(T, E1);
  const float D = dot(P, E1);
  if (D == 0.0f) return false;
  const float t = dot(Q, E2) / D;
  if (t <= ray.tmin || t >= ray.tmax) return false;
  const float u = dot(P, T) / D;
  if (u < 0.0f || u > 1.0f) return false;
  const float w = dot(Q, ray.d) / D;
  if (w < 0.0f || w > 1.0f) return false;
  if (u + w > 1.0f) return false;
  ray.tmax = t;
  return true;
}
// Mesh::compute_surface_info without normals/texcoords (geometry.cpp:632-646)
PINE_HD void tri_surface_info(const float* v, f3 p, DSurface& it) {
  const f3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
  const f3 e1 = v1 - v0, e2 = v2 - v0;
  const f3 n = cross(e1, e2);
  const m3 tbn = inverse(m3{e1, e2, n});
  const f3 q = mul(tbn, p - v0);
  it.uv = f2{q.x, q.y};
  it.p = lerp3(q.x, q.y, v0, v1, v2);
  it.n = normalize(n);
}

// Mesh::compute_surface_info (geometry.cpp:632-646) for a mesh that may carry per-vertex normals / texcoords: after the
// geometric part above, it.n = normalize(lerp(u, v, n0, n1, n2)) (geometry.h:199-204) and it.uv = lerp(u, v, t0, t1, t2)
// (:205-210), both read with the BARYCENTRIC uv.  attrs: 16 floats per triangle (n0 n1 n2, t0 t1 t2, pad).
PINE_HD void mesh_surface_info(const float* tri_verts, const float* tri_attrs, int flags, int prim, f3 p, DSurface& it) {
  tri_surface_info(tri_verts + size_t(prim) * 9, p, it);
  if (flags != 0) {
    const float* a = tri_attrs + size_t(prim) * 16;
    const f2 bary = it.uv;
    if (flags & 1) it.n = normalize(lerp3(bary.x, bary.y, ld3(a), ld3(a + 3), ld3(a + 6)));
    if (flags & 2) it.uv = (1.0f - bary.x - bary.y) * f2{a[9], a[10]} + bary.x * f2{a[11], a[12]} + bary.y * f2{a[13], a[14]};
  }
}

// ------------------------------------------------------------------------------------------------
// Light sampling on shapes (Shape::sample geometry.h:331-340) and Shape::pdf
// ------------------------------------------------------------------------------------------------
struct DShapeSample {
  f3 p, n, w;
  float distance, pdf;
};
PINE_HD bool tri_sample(const float* v, f3 p, f2 u, DShapeSample& ss) {  // Triangle::sample :575-584
  const f3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
  f3 n = normalize(cross(v0 - v1, v0 - v2));  // Triangle(v0,v1,v2) :528-531
  if (is_zero(n)) n = mk3(0, 0, 1);
  if (u.x + u.y > 1.0f) u = f2{1.0f, 1.0f} - u;
  ss.p = lerp3(u.x, u.y, v0, v1, v2);
  ss.n = n;
  ss.w = normalize(ss.p - p, ss.distance);
  const float area = length(cross(v1 - v0, v2 - v0)) / 2;
  ss.pdf = sqr(ss.distance) / pmax(absdot(ss.w, ss.n) * area, kEpsilon);
  return true;
}
template <unsigned F = F_ALL>
PINE_HD bool shape_sample(const DShape* S, const float* tri_verts, f3 o, f2 u, float u1, DShapeSample& ss) {
  const float* f = S->f;
  switch (S->kind) {
    case SHAPE_RECT: {  // geometry.cpp:308-316
      ss.p = ld3(f) + (u.x - 0.5f) * ld3(f + 3) * f[12] + (u.y - 0.5f) * ld3(f + 6) * f[13];
      ss.n = ld3(f + 9);
      ss.w = normalize(ss.p - o, ss.distance);
      ss.pdf = sqr(ss.distance) / (absdot(ss.w, ss.n) * f[20]);
      break;
    }
    case SHAPE_SPHERE: if constexpr (!(F & F_SPHERE)) __builtin_unreachable(); else {  // geometry.cpp:99-114
      const f3 c = ld3(f);
      const float r = f[3];
      const float l = length(c - o);
      const float cos_theta = psqrt(1 - sqr(r / l));
      const float Sa = 2 * kPi * (1 - cos_theta);
      const float cos_theta_wo = 1 - u.y * (1 - cos_theta);
      const float sin_theta_wo = psqrt(1 - cos_theta_wo * cos_theta_wo);
      const float phi = u.x * 2 * kPi;
      f3 w = f3{sin_theta_wo * pcos(phi), sin_theta_wo * psin(phi), cos_theta_wo};
      w = mul(coordinate_system((c - o) / l), w);
      ss.w = w;
      ss.distance = sphere_compute_t(o, w, 0.0f, c, r);
      ss.pdf = prcp(Sa);
      ss.p = o + w * ss.distance;
      ss.n = (ss.p - c) / r;
      break;
    }
    case SHAPE_DISK: if constexpr (!(F & F_DISK)) __builtin_unreachable(); else {  // geometry.cpp:156-165
      const f2 uv = sample_disk_concentric(u);
      const float r = f[12];
      ss.p = ld3(f) + r * ld3(f + 6) * uv.x + r * ld3(f + 9) * uv.y;
      ss.n = ld3(f + 3);
      ss.w = normalize(ss.p - o, ss.distance);
      ss.pdf = sqr(ss.distance) / pmax(absdot(ss.w, ss.n) * f[13], kEpsilon);
      break;
    }
    case SHAPE_PLANE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:57-69
      const f3 position = ld3(f), n = ld3(f + 3), pu = ld3(f + 6), pv = ld3(f + 9);
      const float phi = u.x * kPi * 2;  // uniform_hemisphere sampling.h:56-62
      const float cos_theta = u.y;
      const float sin_theta = psqrt(1.0f - sqr(cos_theta));
      const f3 ps = f3{sin_theta * pcos(phi), sin_theta * psin(phi), cos_theta};
      const float l = absdot(o - position, n);
      const float ex = l * ps.x / ps.z;
      const float ey = l * ps.y / ps.z;
      const f3 dp = o - position;
      ss.p = (position + pu * dot(pu, dp) + pv * dot(pv, dp)) + pu * ex + pv * ey;
      ss.n = n;
      ss.w = normalize(ss.p - o, ss.distance);
      ss.pdf = 1.0f / (2 * kPi);
      break;
    }
    case SHAPE_LINE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else {  // geometry.cpp:223-233
      const f3 p0 = ld3(f), p1 = ld3(f + 3), tx = ld3(f + 6), ty = ld3(f + 9);
      const float th = f[15];
      const float phi = u.y * 2 * kPi;
      ss.p = (u.x * p1 + (1.0f - u.x) * p0) + th * pcos(phi) * tx + th * psin(phi) * ty;
      ss.n = pcos(phi) * tx + psin(phi) * ty;
      ss.w = normalize(ss.p - o, ss.distance);
      ss.pdf = sqr(ss.distance) / (absdot(ss.w, ss.n) * f[16]);
      break;
    }
    case SHAPE_TRIANGLE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else { tri_sample(f, o, u, ss); } break;  // :575-584
    case SHAPE_MESH: if constexpr (!(F & F_MESH)) __builtin_unreachable(); else {  // Mesh::sample geometry.h:170-178
      const int first = as_int(f[0]), nt = as_int(f[1]);
      if (nt == 0) return false;
      const int ti = int(float(size_t(nt)) * u1);
      tri_sample(tri_verts + size_t(first + ti) * 9, o, u, ss);
      ss.pdf /= float(size_t(nt));
      break;
    }
    default: return false;  // Cone::sample returns {} (geometry.cpp:461); boxes are not samplable
  }
  if (ss.pdf <= 0 || isinf(ss.pdf)) return false;
  return true;
}
template <unsigned F = F_ALL>
PINE_HD float shape_pdf(const DShape* S, const DRay& ray, f3 ns) {
  const float* f = S->f;
  switch (S->kind) {
    case SHAPE_RECT: return sqr(ray.tmax) / f[20] * absdot(ns, ray.d);  // geometry.cpp:368-370 (quirk A4)
    case SHAPE_SPHERE: if constexpr (!(F & F_SPHERE)) __builtin_unreachable(); else {                                               // :115-120
      const float l = length(ld3(f) - ray.o);
      const float cos_theta = psqrt(1 - sqr(f[3] / l));
      const float Sa = 2 * kPi * (1 - cos_theta);
      return prcp(Sa);
    }
    case SHAPE_DISK: if constexpr (!(F & F_DISK)) __builtin_unreachable(); else { return sqr(ray.tmax) / (f[13] * absdot(ns, ray.d)); }  // :166-168
    case SHAPE_CONE: if constexpr (!(F & F_CONE)) __builtin_unreachable(); else { return sqr(ray.tmax) / f[11] * absdot(ns, ray.d); }  // :462-464
    case SHAPE_MESH: if constexpr (!(F & F_MESH)) __builtin_unreachable(); else { return sqr(ray.tmax) / (f[3] * absdot(ns, ray.d)); }  // geometry.h:180-182
    case SHAPE_PLANE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else { return 1.0f / (2 * kPi); }  // geometry.cpp:70
    case SHAPE_LINE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else { return sqr(ray.tmax) / (f[16] * absdot(ns, ray.d)); }  // :234-236
    case SHAPE_TRIANGLE: if constexpr (!(F & F_XSHAPES)) __builtin_unreachable(); else { return sqr(ray.tmax) / (f[12] * absdot(ns, ray.d)); }  // :585-587
    default: return 0.0f;
  }
}

// ------------------------------------------------------------------------------------------------
// BSDFs (src/pine/core/bxdf.cpp, scattering.h)
// ------------------------------------------------------------------------------------------------
PINE_HD float CosTheta(f3 w) { return w.z; }
PINE_HD float Cos2Theta(f3 w) { return sqr(w.z); }
PINE_HD float AbsCosTheta(f3 w) { return pabs(w.z); }
PINE_HD float Sin2Theta(f3 w) { return 1.0f - Cos2Theta(w); }
PINE_HD float SinTheta(f3 w) { return psqrt(Sin2Theta(w)); }
PINE_HD float Tan2Theta(f3 w) { return Sin2Theta(w) / pmax(Cos2Theta(w), kEpsilon); }
PINE_HD float CosPhi(f3 w) {
  const float s = SinTheta(w);
  return (s == 0) ? 1 : pclamp(w.x / s, -1.0f, 1.0f);
}
PINE_HD float SinPhi(f3 w) {
  const float s = SinTheta(w);
  return (s == 0) ? 1 : pclamp(w.y / s, -1.0f, 1.0f);
}
PINE_HD bool SameHemisphere(f3 a, f3 b) { return a.z * b.z > 0.0f; }
PINE_HD f3 FaceNormal(f3 v) { return v.z < 0.0f ? -v : v; }
PINE_HD f3 Reflect(f3 w) { return f3{-w.x, -w.y, w.z}; }
PINE_HD f3 Reflect(f3 wi, f3 n) { return 2.0f * dot(wi, n) * n - wi; }
PINE_HD bool Refract(f3 wi, f3 n, float eta, f3& wt, float* etap) {  // scattering.h:58-77
  float cosThetaI = dot(n, wi);
  if (cosThetaI < 0) {
    eta = 1.0f / eta;
    cosThetaI = -cosThetaI;
    n = -n;
  }
  const float sin2ThetaI = pmax(0.0f, 1.0f - sqr(cosThetaI));
  const float sin2ThetaT = sin2ThetaI / sqr(eta);
  if (sin2ThetaT >= 1) return false;
  const float cosThetaT = psqrt(1.0f - sin2ThetaT);
  wt = -wi / eta + (cosThetaI / eta - cosThetaT) * n;
  if (etap) *etap = eta;
  return true;
}
PINE_HD float FrDielectric(float cosThetaI, float eta) {  // scattering.h:79-94
  if (cosThetaI < 0) {
    eta = 1 / eta;
    cosThetaI = -cosThetaI;
  }
  const float sin2ThetaI = 1.0f - sqr(cosThetaI);
  const float sin2ThetaT = sin2ThetaI / sqr(eta);
  if (sin2ThetaT >= 1.0f) return 1.0f;
  const float cosThetaT = psqrt(1.0f - sin2ThetaT);
  const float rParl = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
  const float rPerp = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
  return (sqr(rParl) + sqr(rPerp)) / 2.0f;
}
PINE_HD f3 FrSchlick(f3 F0, float cosTheta) {  // scattering.h:96-98 (powf: device libm, tolerance)
  return F0 + (mk3(1.0f) - F0) * ppow(1.0f - cosTheta, 5.0f);
}
struct TRDist {  // TrowbridgeReitzDistribution scattering.h:100-150
  float ax, ay;
};
PINE_HD float tr_D(const TRDist& t, f3 wm) {
  const float tan2Theta = Tan2Theta(wm);
  const float cos4Theta = sqr(Cos2Theta(wm));
  if (cos4Theta < 1e-6f) return 0.0f;
  const float e = tan2Theta * (sqr(CosPhi(wm) / t.ax) + sqr(SinPhi(wm) / t.ay));
  return 1.0f / (kPi * t.ax * t.ay * cos4Theta * sqr(1 + e));
}
PINE_HD float tr_Lambda(const TRDist& t, f3 w) {
  const float tan2Theta = Tan2Theta(w);
  const float alpha2 = sqr(CosPhi(w) * t.ax) + sqr(SinPhi(w) * t.ay);
  return (psqrt(1.0f + alpha2 * tan2Theta) - 1.0f) / 2.0f;
}
PINE_HD float tr_G1(const TRDist& t, f3 w) { return 1.0f / (1.0f + tr_Lambda(t, w)); }
PINE_HD float tr_G(const TRDist& t, f3 wi, f3 wo) { return 1.0f / (1.0f + tr_Lambda(t, wi) + tr_Lambda(t, wo)); }
PINE_HD float tr_D_G(const TRDist& t, f3 wi, f3 wm, f3 wo) { return tr_D(t, wm) * tr_G(t, wi, wo); }
PINE_HD float tr_Dw(const TRDist& t, f3 w, f3 wm) { return tr_G1(t, w) / AbsCosTheta(w) * tr_D(t, wm) * absdot(w, wm); }
PINE_HD float tr_pdf(const TRDist& t, f3 w, f3 wm) { return pmax(tr_Dw(t, w, wm), kEpsilon); }
PINE_HD f3 tr_SampleWm(const TRDist& t, f3 w, f2 u) {
  f3 wh = normalize(f3{t.ax * w.x, t.ay * w.y, w.z});
  if (wh.z < 0.0f) wh = -wh;
  const f3 T1 = (wh.z < 0.99999f) ? normalize(cross(mk3(0, 0, 1), wh)) : mk3(1, 0, 0);
  const f3 T2 = cross(wh, T1);
  f2 p = sample_disk_polar(u);
  const float h = psqrt(1.0f - sqr(p.x));
  const float tt = (1.0f + wh.z) / 2;
  p.y = h * (1.0f - tt) + p.y * tt;  // psl::lerp(t, a, b) = a*(1-t) + b*t (src/psl/math.h:118-121)
  const float pz = psqrt(pmax(0.0f, 1.0f - (p.x * p.x + p.y * p.y)));
  const f3 nh = p.x * T1 + p.y * T2 + pz * wh;
  return normalize(f3{t.ax * nh.x, t.ay * nh.y, pmax(1e-6f, nh.z)});
}

enum BxdfKind : int { BX_DIFFUSE, BX_CONDUCTOR, BX_REFRACTIVE, BX_REFR_DIEL, BX_DIFF_DIEL, BX_BSSRDF };
struct DBxdf {
  int kind;
  f3 albedo;
  f3 albedo_over_pi;  // albedo / Pi precomputed on the host (identical IEEE division)
  float roughness, ior;
  f3 wi;  // local frame
};
template <unsigned F = F_ALL>
PINE_HD bool bxdf_is_delta(const DBxdf& b) {
  if constexpr (!(F & (F_UBER | F_SSS))) return false;
  switch (b.kind) {
    case BX_DIFFUSE:
    case BX_DIFF_DIEL:
    case BX_BSSRDF: return false;
    default: return b.roughness < 1e-2f;
  }
}
// ------------------------------------------------------------------------------------------------
// Shading nodes: postfix programs over vec3 (pine_types.h DNodeOp; node.h:13-297, node.cpp:15-18).
// A float node travels as a splat; every operation is componentwise, so the float result is each
// component of the vec3 result.
// ------------------------------------------------------------------------------------------------
PINE_HD f3 node_program_eval(const DNodeOp* ops, int start, f3 p, f3 n, f2 uv) {
  f3 st[kNodeStack];
  int sp = 0;
  for (int pc = start;; pc++) {
    const DNodeOp o = ops[pc];
    if (o.op == N_END) break;
    switch (o.op) {
      case N_CONST: st[sp++] = f3{o.x, o.y, o.z}; break;
      case N_POS: st[sp++] = p; break;
      case N_NORMAL: st[sp++] = n; break;
      case N_UV: st[sp++] = f3{uv.x, uv.y, 0.0f}; break;  // explicit Vector3(Vector2) vecmath.h:167
      case N_ADD: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
      case N_SUB: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
      case N_MUL: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
      case N_DIV: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
      case N_POW: {
        sp--;
        const f3 a = st[sp - 1], b = st[sp];
        st[sp - 1] = f3{ppow(a.x, b.x), ppow(a.y, b.y), ppow(a.z, b.z)};
        break;
      }
      case N_NEG: st[sp - 1] = -st[sp - 1]; break;
      case N_ABS: st[sp - 1] = vabs(st[sp - 1]); break;
      case N_SQR: st[sp - 1] = st[sp - 1] * st[sp - 1]; break;
      case N_SQRT: {
        const f3 a = st[sp - 1];
        st[sp - 1] = f3{psqrt(a.x), psqrt(a.y), psqrt(a.z)};
        break;
      }
      case N_FRACT: {
        const f3 a = st[sp - 1];
        st[sp - 1] = f3{a.x - floorf(a.x), a.y - floorf(a.y), a.z - floorf(a.z)};
        break;
      }
      case N_COMP: {
        const float v = get(st[sp - 1], int(o.x));
        st[sp - 1] = f3{v, v, v};
        break;
      }
      case N_TOVEC3: sp -= 2; st[sp - 1] = f3{st[sp - 1].x, st[sp].x, st[sp + 1].x}; break;
      case N_CHECKER: {
        const f3 a = st[sp - 1];
        const f3 x{(a.x - floorf(a.x)) - o.x, (a.y - floorf(a.y)) - o.x, (a.z - floorf(a.z)) - o.x};
        const float v = float(x.x * x.y * x.z > 0);
        st[sp - 1] = f3{v, v, v};
        break;
      }
      default: break;
    }
  }
  return st[0];
}
// Lights other than emissive geometry (light.cpp:11-84).  Returns false when the light does not reach p.
PINE_HD f3 sky_color_of(f3 sun_color, f3 wo) {  // Sky::color light.cpp:71-73, sky_color color.cpp:100-103
  const float t = wo.y / 2 + 0.7f;
  const f3 a{1.0f, 0.8f, 0.6f}, b{0.6f, 0.8f, 1.0f};
  const f3 l = a * (1 - t) + b * t;  // psl::lerp(t, a, b) math.h:118-121
  return sun_color * (l * l);
}
PINE_HD bool light_sample_other(const DLight* L, f3 p, f2 u2, f3& w, float& distance, float& pdf, f3& le) {
  switch (L->kind) {
    case LIGHT_POINT:  // light.cpp:11-17
      w = normalize(ld3(L->position) - p, distance);
      pdf = sqr(distance);
      le = ld3(L->color);
      return true;
    case LIGHT_SPOT: {  // light.cpp:35-47
      w = normalize(ld3(L->position) - p, distance);
      const float cs = -dot(w, ld3(L->direction));
      if (cs > L->falloff_cos) le = ld3(L->color);
      else if (cs > L->cutoff_cos) le = ld3(L->color) * (cs - L->cutoff_cos) / (L->falloff_cos - L->cutoff_cos);
      else return false;
      pdf = sqr(distance);
      return true;
    }
    case LIGHT_DIRECTIONAL:  // light.cpp:48-54
      distance = 1e+10f;
      w = ld3(L->direction);
      pdf = 1.0f;
      le = ld3(L->color);
      return true;
    case LIGHT_SKY:  // light.cpp:74-81
      w = uniform_sphere(u2);
      pdf = 1 / (4 * kPi);
      distance = kFloatMax;
      le = sky_color_of(ld3(L->color), w);
      return true;
    default: return false;
  }
}

// A material's parameters at a surface point: the literals of the record, or (F_NODES variants) the
// values of its node programs there -- BxdfSampleCtx -> NodeEvalCtx(it) (bxdf.h:10-21, node.h:13-20).
struct MatParams {
  f3 albedo, albedo_over_pi;
  float roughness, metallic, transmission, ior;
};
template <unsigned F = F_ALL>
PINE_HD MatParams material_params(const DMaterial* m, const DNodeOp* ops, f3 p, f3 n, f2 uv) {
  MatParams r{ld3(m->color), ld3(m->color_over_pi), m->roughness, m->metallic, m->transmission, m->ior};
  if constexpr (F & F_NODES) {
    if (m->prog[0] >= 0) {
      r.albedo = node_program_eval(ops, m->prog[0], p, n, uv);
      r.albedo_over_pi = r.albedo / kPi;  // the Lambertian f (bxdf.cpp:21,27)
    }
    if (m->prog[1] >= 0) r.roughness = node_program_eval(ops, m->prog[1], p, n, uv).x;
    if (m->prog[2] >= 0) r.metallic = node_program_eval(ops, m->prog[2], p, n, uv).x;
    if (m->prog[3] >= 0) {
      const float v = node_program_eval(ops, m->prog[3], p, n, uv).x;
      if (m->kind == MAT_GLOSSY || m->kind == MAT_GLASS) r.ior = v;
      else r.transmission = v;
    }
  }
  return r;
}

struct DBsdfSample {
  f3 wo, f;
  float pdf;
  bool is_delta;
};
template <unsigned F = F_ALL, int LDS = 0>
PINE_HD bool bxdf_sample(const DBxdf& b, const DTables& T, DSampler& sampler, DBsdfSample& bs) {
  const f3 wi = b.wi;
  bs.is_delta = false;
  switch (b.kind) {
    case BX_DIFFUSE: {  // bxdf.cpp:11-23
      f3 wo = cosine_weighted_hemisphere(sampler_get2d<LDS>(T, sampler));
      if (CosTheta(wi) < 0) wo = -wo;
      bs.wo = wo;
      bs.pdf = AbsCosTheta(wo) / kPi;
      bs.f = b.albedo_over_pi;
      return true;
    }
    case BX_CONDUCTOR: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:39-64
      const float alpha = sqr(b.roughness);
      if (alpha < 1e-4f) {
        bs.wo = Reflect(wi);
        bs.f = FrSchlick(b.albedo, AbsCosTheta(bs.wo)) / AbsCosTheta(bs.wo);
        bs.pdf = 1.0f;
        bs.is_delta = true;
        return true;
      }
      const TRDist d{alpha, alpha};
      const f3 wm = tr_SampleWm(d, wi, sampler_get2d<LDS>(T, sampler));
      const f3 wo = Reflect(wi, wm);
      if (!SameHemisphere(wi, wo)) return false;
      const f3 fr = FrSchlick(b.albedo, absdot(wi, wm));
      bs.wo = wo;
      bs.pdf = tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
      bs.f = fr * (tr_D_G(d, wo, wm, wi) / (4 * CosTheta(wi) * CosTheta(wo)));
      return true;
    }
    case BX_REFRACTIVE: if constexpr (!(F & F_SSS)) __builtin_unreachable(); else {  // bxdf.cpp:102-124
      const float alpha = sqr(b.roughness);
      if (alpha < 1e-4f) {
        bs.wo = Reflect(wi);
        bs.f = b.albedo;
        bs.pdf = AbsCosTheta(bs.wo);
        bs.is_delta = true;
        return true;
      }
      const TRDist d{alpha, alpha};
      const f3 wm = tr_SampleWm(d, wi, sampler_get2d<LDS>(T, sampler));
      const f3 wo = Reflect(wi, wm);
      if (!SameHemisphere(wi, wo)) return false;
      bs.wo = wo;
      bs.pdf = tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
      bs.f = b.albedo * (tr_D_G(d, wo, wm, wi) / (4 * CosTheta(wi) * CosTheta(wo)));
      return true;
    }
    case BX_REFR_DIEL: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:162-208
      const float fr = FrDielectric(CosTheta(wi), b.ior);
      const float alpha = sqr(b.roughness);
      if (alpha < 1e-4f) {
        if (sampler_get1d<LDS>(T, sampler) < fr) {
          bs.wo = Reflect(wi);
          bs.f = b.albedo * (fr / AbsCosTheta(bs.wo));
          bs.pdf = fr;
          bs.is_delta = true;
        } else {
          if (!Refract(wi, mk3(0, 0, 1), b.ior, bs.wo, nullptr)) return false;
          bs.f = b.albedo * ((1 - fr) / AbsCosTheta(bs.wo));
          bs.pdf = 1 - fr;
          bs.is_delta = true;
        }
        return true;
      }
      const TRDist d{alpha, alpha};
      const f3 wm = tr_SampleWm(d, wi, sampler_get2d<LDS>(T, sampler));
      if (sampler_get1d<LDS>(T, sampler) < fr) {
        const f3 wo = Reflect(wi, wm);
        if (!SameHemisphere(wi, wo)) return false;
        bs.wo = wo;
        bs.pdf = fr * tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
        bs.f = b.albedo * (fr * tr_D_G(d, wo, wm, wi) / (4 * CosTheta(wi) * CosTheta(wo)));
      } else {
        float eta = 1.0f;
        if (!Refract(wi, wm, b.ior, bs.wo, &eta)) return false;
        const f3 wo = bs.wo;
        const float denom = sqr(dot(wo, wm) + dot(wi, wm) / eta);
        bs.pdf = (1 - fr) * tr_pdf(d, wi, wm) * absdot(wo, wm) / denom;
        bs.f = b.albedo * ((1 - fr) * tr_D(d, wm) * tr_G(d, wi, wo) *
                           pabs(dot(wo, wm) * dot(wi, wm) / (denom * CosTheta(wi) * CosTheta(wo))));
      }
      return true;
    }
    case BX_DIFF_DIEL: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:250-287 (diffuse lobe is NOT flipped to wi's side, Appendix A5)
      const float fr = FrDielectric(CosTheta(wi), b.ior);
      const float alpha = sqr(b.roughness);
      if (alpha < 1e-4f) {
        if (sampler_get1d<LDS>(T, sampler) < fr) {
          bs.wo = Reflect(wi);
          bs.f = mk3(fr);
          bs.pdf = fr * AbsCosTheta(bs.wo);
          bs.is_delta = true;
        } else {
          bs.wo = cosine_weighted_hemisphere(sampler_get2d<LDS>(T, sampler));
          bs.f = b.albedo * ((1 - fr) / kPi);
          bs.pdf = (1 - fr) * AbsCosTheta(bs.wo) / kPi;
        }
        return true;
      }
      const TRDist d{alpha, alpha};
      const f3 wm = tr_SampleWm(d, wi, sampler_get2d<LDS>(T, sampler));
      if (sampler_get1d<LDS>(T, sampler) < fr) {
        const f3 wo = Reflect(wi, wm);
        if (!SameHemisphere(wi, wo)) return false;
        bs.wo = wo;
        bs.f = mk3(fr * tr_D_G(d, wi, wm, wo) / (4 * CosTheta(wi) * CosTheta(wo)));
        bs.pdf = fr * tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
      } else {
        bs.wo = cosine_weighted_hemisphere(sampler_get2d<LDS>(T, sampler));
        bs.f = b.albedo * ((1 - fr) / kPi);
        bs.pdf = AbsCosTheta(bs.wo) * (1 - fr) / kPi;
      }
      return true;
    }
    case BX_BSSRDF: if constexpr (!(F & F_SSS)) __builtin_unreachable(); else {  // bxdf.cpp:356-367
      f3 wo = cosine_weighted_hemisphere(sampler_get2d<LDS>(T, sampler));
      if (CosTheta(wi) > 0) wo = -wo;
      bs.wo = wo;
      bs.pdf = AbsCosTheta(wo) / kPi;
      bs.f = b.albedo_over_pi;
      return true;
    }
  }
  return false;
}
template <unsigned F = F_ALL>
PINE_HD f3 bxdf_f(const DBxdf& b, f3 wo) {
  const f3 wi = b.wi;
  switch (b.kind) {
    case BX_DIFFUSE:  // bxdf.cpp:24-28
      if (!SameHemisphere(wi, wo)) return mk3(0.0f);
      return b.albedo_over_pi;
    case BX_CONDUCTOR: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:65-79
      if (!SameHemisphere(wi, wo)) return mk3(0.0f);
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      const f3 wm = normalize(wi + wo);
      if (is_zero(wm)) return mk3(0.0f);
      const f3 fr = FrSchlick(b.albedo, absdot(wi, wm));
      return fr * (tr_D_G(d, wo, wm, wi) / (4 * AbsCosTheta(wo) * AbsCosTheta(wi)));
    }
    case BX_REFRACTIVE: if constexpr (!(F & F_SSS)) __builtin_unreachable(); else {  // bxdf.cpp:125-140
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      const float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
      if (!(cosThetaI * cosThetaO > 0)) return mk3(0.0f);
      const f3 wm = FaceNormal(normalize(wo + wi));
      if (dot(wm, wo) * cosThetaO <= 0 || dot(wm, wi) * cosThetaI <= 0) return mk3(0.0f);
      return b.albedo * (tr_D_G(d, wi, wm, wo) / pabs(4 * cosThetaI * cosThetaO));
    }
    case BX_REFR_DIEL: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:209-230 (`auto eta = 1` is an int in the reference)
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      const float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
      const bool reflect = cosThetaI * cosThetaO > 0;
      int eta = 1;
      if (!reflect) eta = int(cosThetaI > 0 ? b.ior : 1 / b.ior);
      const f3 wm = FaceNormal(normalize(wo * float(eta) + wi));
      if (dot(wm, wo) * cosThetaO <= 0 || dot(wm, wi) * cosThetaI <= 0) return mk3(0.0f);
      const float fr = FrDielectric(dot(wi, wm), b.ior);
      if (reflect) return b.albedo * (fr * tr_D_G(d, wi, wm, wo) / pabs(4 * cosThetaI * cosThetaO));
      const float denom = sqr(dot(wo, wm) + dot(wi, wm) / float(eta)) * cosThetaI * cosThetaO;
      return b.albedo * ((1 - fr) * tr_D(d, wm) * tr_G(d, wi, wo) * pabs(dot(wo, wm) * dot(wi, wm) / denom));
    }
    case BX_DIFF_DIEL: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:288-306
      if (!SameHemisphere(wi, wo)) return mk3(0.0f);
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      const float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
      const f3 wm = FaceNormal(normalize(wo + wi));
      if (dot(wm, wo) * cosThetaO <= 0 || dot(wm, wi) * cosThetaI <= 0) return mk3(0.0f);
      const float fr = FrDielectric(dot(wi, wm), b.ior);
      const f3 diffused = b.albedo * (1 - fr) / kPi;
      if (alpha < 1e-4f) return diffused;
      const float reflected = fr * tr_D_G(d, wo, wm, wi) / pabs(4 * cosThetaI * cosThetaO);
      return mk3(reflected) + diffused;
    }
    case BX_BSSRDF: if constexpr (!(F & F_SSS)) __builtin_unreachable(); else { return b.albedo_over_pi; }  // bxdf.cpp:368-370
  }
  return mk3(0.0f);
}
template <unsigned F = F_ALL>
PINE_HD float bxdf_pdf(const DBxdf& b, f3 wo) {
  const f3 wi = b.wi;
  switch (b.kind) {
    case BX_DIFFUSE:  // bxdf.cpp:29-33
      if (!SameHemisphere(wi, wo)) return 0.0f;
      return AbsCosTheta(wo) / kPi;
    case BX_CONDUCTOR: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:80-95
      if (!SameHemisphere(wi, wo)) return 0.0f;
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      f3 wm = normalize(wi + wo);
      if (is_zero(wm)) return 0.0f;
      wm = FaceNormal(wm);
      return tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
    }
    case BX_REFRACTIVE: if constexpr (!(F & F_SSS)) __builtin_unreachable(); else {  // bxdf.cpp:141-157
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      const float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
      if (!(cosThetaI * cosThetaO > 0)) return 0.0f;
      const f3 wm = FaceNormal(normalize(wo + wi));
      if (dot(wm, wo) * cosThetaO <= 0 || dot(wm, wi) * cosThetaI <= 0) return 0.0f;
      return tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
    }
    case BX_REFR_DIEL: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:231-245
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      const float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
      const bool reflect = cosThetaI * cosThetaO > 0;
      int eta = 1;
      if (!reflect) eta = int(cosThetaI > 0 ? b.ior : 1 / b.ior);
      const f3 wm = FaceNormal(normalize(wo * float(eta) + wi));
      if (dot(wm, wo) * cosThetaO <= 0 || dot(wm, wi) * cosThetaI <= 0) return 0.0f;
      const float fr = FrDielectric(dot(wi, wm), b.ior);
      if (reflect) return fr * tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
      const float denom = sqr(dot(wo, wm) + dot(wi, wm) / float(eta));
      const float dwm_dwo = absdot(wo, wm) / denom;
      return (1 - fr) * tr_pdf(d, wi, wm) * dwm_dwo;
    }
    case BX_DIFF_DIEL: if constexpr (!(F & F_UBER)) __builtin_unreachable(); else {  // bxdf.cpp:307-324
      if (!SameHemisphere(wi, wo)) return 0.0f;
      const float alpha = sqr(b.roughness);
      const TRDist d{alpha, alpha};
      const float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
      const f3 wm = FaceNormal(normalize(wo + wi));
      if (dot(wm, wo) * cosThetaO <= 0 || dot(wm, wi) * cosThetaI <= 0) return 0.0f;
      const float fr = FrDielectric(dot(wi, wm), b.ior);
      const float pt = (1 - fr) * AbsCosTheta(wo) / kPi;
      if (alpha < 1e-4f) return pt;
      const float pr = fr * tr_pdf(d, wi, wm) / (4 * absdot(wi, wm));
      return pr + pt;
    }
    case BX_BSSRDF: if constexpr (!(F & F_SSS)) __builtin_unreachable(); else { return AbsCosTheta(wo) / kPi; }  // bxdf.cpp:371-373
  }
  return 0.0f;
}

// ThinLenCamera::gen_ray (src/pine/core/camera.cpp:17-33)
PINE_HD DRay camera_gen_ray(const DCamera& c, f2 p_film, f2 u2) {
  p_film = (p_film - f2{0.5f, 0.5f}) * 2.0f;
  const f2 pc = p_film * f2{c.fov2d[0], c.fov2d[1]};
  const m3 c2w{ld3(c.c2w), ld3(c.c2w + 3), ld3(c.c2w + 6)};
  DRay r;
  r.tmin = 0.0f;
  r.tmax = kFloatMax;
  if (c.len_radius == 0.0f) {
    r.o = ld3(c.position);
    r.d = normalize(mul(c2w, f3{pc.x, pc.y, 1.0f}));
  } else {
    const f3 dir = normalize(f3{pc.x, pc.y, 1.0f});
    const f3 p_focus = c.focus_distance * dir / dir.z;
    const f2 d = c.len_radius * sample_disk_polar(u2);
    const f3 p_len{d.x, d.y, 0.0f};
    r.o = ld3(c.position) + p_len;
    r.d = mul(c2w, normalize(p_focus - p_len));
  }
  return r;
}

}  // namespace pine_gpu
