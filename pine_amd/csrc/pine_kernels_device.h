// pine_amd/csrc/pine_kernels_device.h -- the device side of the PathIntegrator hot path: types shared by host and
// kernels (DeviceScene, WorkParams, Counters, PackedState), BVH traversal, the lane-owns-a-path kernel and -- through
// pine_trav.h / pine_queue_kernel.h -- the stage-queued kernel.  Included by pine_kernels.hip (exact arithmetic: the
// parity build) and by pine_kernels_fast.hip (declared-tolerance arithmetic, under another namespace).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pine_device.h"

namespace pine_gpu {

constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kTile = 8;         // 8x8 pixel tiles = 64 pixels = one wave's worth of items
#ifndef PINE_LDS_FOLD_LEVELS
#define PINE_LDS_FOLD_LEVELS 1
#endif
constexpr int kLdsFoldLevels = PINE_LDS_FOLD_LEVELS;  // fold-stack levels kept in LDS (deeper levels spill to global memory)
constexpr int kPoolItems = 128;  // items a wave claims from the global queue per atomic
constexpr int kMaxDepth = 32;    // max_path_length supported (2 beta bits per level in one u64)
// feature sets that several compiled variants share (pine_variants.h)
constexpr unsigned kFBoxes = F_AABB | F_OBB;
constexpr unsigned kFAnalytic = F_AABB | F_OBB | F_SPHERE | F_DISK | F_CONE | F_UBER;

// Diagnostic section timing: per-wave s_memtime deltas summed per section.  Never compiled into the
// product build; the stamps only go to Counters::section_cycles, which nothing else reads.
#ifdef PINE_PROFILE_SECTIONS
static __device__ unsigned long long g_region_lanes[16], g_region_hits[16];  // (one copy per translation unit: pine_kernels_part.hip reads its own)
// REGION(id): average number of active lanes at a code region's entry (divergence probe;
// -DPINE_PROFILE_REGIONS on top, as its global atomics distort the section times)
#ifndef PINE_PROFILE_REGIONS
#define REGION(id)
#else
#define REGION(id)                                                                  \
  do {                                                                              \
    const unsigned long long m_ = __ballot(1);                                      \
    if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)m_) - 1) {               \
      atomicAdd(&g_region_lanes[id], (unsigned long long)__popcll(m_));             \
      atomicAdd(&g_region_hits[id], 1ull);                                          \
    }                                                                               \
  } while (0)
#endif
#define SEC_DECL unsigned long long sec_t = __builtin_readcyclecounter(), sec_acc[16] = {0}
#define SEC_MARK(id)                                              \
  do {                                                            \
    const unsigned long long t_ = __builtin_readcyclecounter();   \
    sec_acc[id] += t_ - sec_t;                                    \
    sec_t = t_;                                                   \
  } while (0)
#define SEC_FLUSH()                                                                     \
  do {                                                                                  \
    if ((threadIdx.x & 63) == 0)                                                        \
      for (int i_ = 0; i_ < 16; i_++) atomicAdd(&counters->section_cycles[i_], sec_acc[i_]); \
  } while (0)
#else
#define REGION(id)
#define SEC_DECL
#define SEC_MARK(id)
#define SEC_FLUSH()
#endif


struct DeviceScene {
  const DShape* shapes;
  const DMaterial* materials;
  const DNode* nodes;
  const int* prims;
  const DBvh* bvhs;
  const float* tri_verts;
  const DLight* lights;
  const DNodeOp* node_ops;  // shading-node programs (F_NODES variants)
  const DShape* leaf;       // leaf[i] = the shape record of top-level primitive entry prims[i], see SceneView
  int num_lights;           // entries of `lights` (the light sampler's N)
  int env_light;            // index of the environment light in `lights`, or -1
  int num_shapes;
  DCamera cam;
  DTables tables;
  int spp;              // effective
  int max_path_length;
  int stack_top;        // traversal stack entries needed by the top-level BVH
  int stack_total;      // top + deepest mesh BVH
  // the small scene records packed in one 16-byte-aligned blob (for LDS staging):
  const uint4* blob;
  int blob_bytes;
  int off_nodes, off_shapes, off_materials, off_bvhs, off_prims, off_lights, off_node_ops, off_leaf;  // byte offsets in the blob
  int top_prim_begin;  // prims[top_prim_begin ..) are the top-level BVH's entries
  const float4* tri_leaf;  // mesh triangles in leaf order, 3 float4 per entry of `prims` (FlatAccel::tri_leaf)
  const float* tri_attrs;  // per-vertex normals / texcoords per triangle, 16 floats each (FlatAccel::tri_attrs), or null
  int lds_nodes;           // F_LDS_TOP variants: nodes[0 .. lds_nodes) are copied to LDS by every workgroup
  // F_LDS_TOP + F_MESH variants: the mesh triangles as LDS-sized packets (plan_build decides whether they are staged):
  // one 8-byte entry per leaf-ordered triangle (three 16-bit vertex numbers, the 16-bit triangle index) and the
  // scene's DISTINCT vertices as float4 -- the same floats as tri_leaf's 48-byte records, a third of the bytes
  const uint4* tri_packets;   // entries (tri_packet_entries x 8 bytes, padded to 16), then vertices (x 16 bytes)
  int tri_packet_entries, tri_packet_verts;
  int lds_tris;               // 1: every workgroup copies the packets to LDS and the traversal reads them there
  // PINE_GPU_FLAG_ORDER_EMBREE (F_EMBREE variants): the BVH8 the reference's EmbreeAccel walks over the non-mesh shapes
  // (pine_embree_order.h) -- EmbreeNode records at off_etree of the blob, the root's child word (kEmbreeNoChild: no such
  // shape) -- the places in `leaf` of the meshes (num_emesh ints at off_emesh; tested first) and the 2048 RCPPS estimates
  // (off_rcpps: BEHIND blob_bytes, global memory only)
  int off_etree, etree_root, off_emesh, num_emesh, off_rcpps;
};

// (EmbreeNode, kEmbreeNoChild, kEmbreeStackEntries: pine_types.h)

// What the traversal and shading code reads.  In the F_LDS_SCENE specialisation every pointer is
// derived from the workgroup's LDS copy of the blob (so the loads are ds_read, ~64-cycle latency,
// instead of L1/L2 round trips); otherwise they point into HBM-backed global memory.
struct SceneView {
  // leaf[i]: a COPY of the shape record of top-level primitive entry i, in BVH leaf order, whose `kind`
  // field holds the packed primitive word (index | emissive | kind).  The leaf loop then needs one memory
  // round trip per primitive (record address = base + 128 i) instead of two dependent ones (word, then
  // shapes[word & mask]).  `leaf` is biased by -top_prim_begin so that the BVH's own indices address it.
  const DShape* leaf;
  const DShape* shapes;
  const DMaterial* materials;
  const DNode* nodes;
  const int* prims;
  const DBvh* bvhs;
  const DLight* lights;
  const float* tri_verts;
  const DNodeOp* node_ops;
  int stack_top;
  int num_shapes;
  const float4* tri_leaf;
  const float* tri_attrs;
  const DNode* lds_nodes;  // F_LDS_TOP: the workgroup's LDS copy of nodes[0 .. lds_node_count)
  int lds_node_count;
  const uint2* lds_tri_entries;   // DeviceScene::tri_packets in LDS (null: triangles are read from tri_leaf)
  const float4* lds_tri_verts;
  // F_EMBREE variants (PINE_GPU_FLAG_ORDER_EMBREE): DeviceScene::off_etree / etree_root / off_emesh / off_rcpps
  const EmbreeNode* etree;
  int etree_root;
  const int* emesh;
  int num_emesh;
  const unsigned* rcpps;
};

// One BVH node into registers.  F_LDS_TOP: from the workgroup's LDS copy when the index is below the cached
// count (four ds_read_b128), else from global memory (four global_load_dwordx4).  The LDS arm goes through an
// address_space(3) pointer: with two generic pointers the compiler folds the branch into a pointer select and
// emits flat loads, which occupy both memory pipes.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
template <unsigned F>
__device__ __forceinline__ DNode fetch_node(const SceneView& S, int index) {
  union {
    DNode n;
    u32x4 q[4];
  } b;
  if constexpr (F & F_LDS_TOP) {
    if (index < S.lds_node_count) {
      lds_u32x4* p = (lds_u32x4*)(S.lds_nodes) + size_t(index) * 4;
      b.q[0] = p[0], b.q[1] = p[1], b.q[2] = p[2], b.q[3] = p[3];
      return b.n;
    }
  }
  const u32x4* g = reinterpret_cast<const u32x4*>(S.nodes + index);
  b.q[0] = g[0], b.q[1] = g[1], b.q[2] = g[2], b.q[3] = g[3];
  return b.n;
}

// One leaf-ordered triangle: from the workgroup's LDS packets when they are staged (an 8-byte entry, then three 16-byte
// vertices: ds_read_b64 + 3 ds_read_b128), else its 48-byte record in global memory.  The floats are the same.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 lds_u32x2;
template <unsigned F>
__device__ __forceinline__ void fetch_triangle(const SceneView& S, int i, float (&v)[9], int& tri) {
  if constexpr ((F & F_LDS_TOP) != 0 && (F & F_MESH) != 0) {
    if (S.lds_tri_entries != nullptr) {
      const u32x2 e = *((lds_u32x2*)(S.lds_tri_entries) + i);
      lds_u32x4* vb = (lds_u32x4*)(S.lds_tri_verts);
      const u32x4 a = vb[e.x & 0xffffu], b = vb[e.x >> 16], c = vb[e.y & 0xffffu];
      v[0] = __uint_as_float(a.x), v[1] = __uint_as_float(a.y), v[2] = __uint_as_float(a.z);
      v[3] = __uint_as_float(b.x), v[4] = __uint_as_float(b.y), v[5] = __uint_as_float(b.z);
      v[6] = __uint_as_float(c.x), v[7] = __uint_as_float(c.y), v[8] = __uint_as_float(c.z);
      tri = int(e.y >> 16);
      return;
    }
  }
  const float4* rec = S.tri_leaf + size_t(i) * 3;
  const float4 a = rec[0], b = rec[1], c = rec[2];
  v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w, v[8] = c.x;
  tri = __float_as_int(c.y);
}

struct WorkParams {
  int tiles_x, tiles_y;
  int num_local_tiles;   // tiles owned by this shard
  int shard_rank, shard_world;
  int samples_per_item;  // k
  int items_per_pixel;   // spp / k (a power of two)
  int log2_items_per_pixel;
  unsigned tiles_x_magic;  // ceil(2^32 / tiles_x): see decode_item
  unsigned long long total_items;  // num_local_tiles * items_per_pixel * 64 (tile classes: see serial_tiles)
  // Tile classes (Subsurface variants of the stage-queued kernel, plan_build): a scene whose materials draw from the
  // pixel's RNG inside radiance() makes a pixel's samples sequentially dependent (one item = the whole pixel) -- but only
  // in pixels whose camera rays can reach such a material.  The host lists the shard's tiles in `tile_order`, those
  // that can first (`serial_tiles` of them, items [0, serial_tiles * 64): one per pixel, all spp samples); the others
  // follow as independent items of samples_per_item samples with RNG checkpoints, like a scene without in-path draws.
  // serial_tiles == 0: one class, as described by samples_per_item / items_per_pixel.
  const int* tile_order;  // local tile -> tile of the film, or null (local tile * shard_world + shard_rank)
  int serial_tiles;
  unsigned long long idle_budget_ticks;  // stage-queued kernel: a wave that finds no work for this long (100 MHz wall clock) bails out
  int debug_force_bail;  // test hook (PINE_GPU_FLAG_DEBUG_FORCE_BAIL): the first wave bails out at once
  int trav_min_lanes, trav_min_trips;  // traversal stages (pine_queue_kernel.h): retire / refill when fewer lanes than this still travel, at the earliest after this many trips
  int fair_period;  // stage-queued kernel, variants with more than two stage queues: every this-many-th pick of a wave serves the shortest non-empty queue (0: never)
  int pick_spins;   // stage-queued kernel: idle polls after which a wave takes a queue's entries although they are fewer than 64
  int pool_items;   // stage-queued kernel: work items a workgroup claims from the global counter at a time
  int max_pixels;   // Subsurface variants: pixels a workgroup has in flight at most
  int fork_sealed;  // Subsurface variants: a path that can make no further RNG draw hands its pixel's next sample to another context
  unsigned long long* progress;  // host-mapped word (or null): work items claimed so far, stored now and then (get_progress)
  // Test hook (pine_gpu_plan_vertex_log; null in every ordinary launch): 16 floats per radiance() invocation at
  // [((py * film_w + px) * spp + sample) * max_path_length + level] -- the layout of `pine_ref vertices` (oracle/ref_driver.cpp):
  // kind | length | direct term (3) | bs.f (3) | cosine | bs.pdf | is_delta | mis | returned light pdf (-1: none) | returned Lo (3)
  float* vertex_log;
};
constexpr int kVertexLogFloats = 16;
// get_progress() (integrator.cpp:17-19): every 16th / 64th pool claim posts the claimed-item count to host memory
__device__ __forceinline__ void post_progress(const WorkParams& W, unsigned long long claimed, unsigned shift) {
  if (W.progress && ((claimed >> shift) & 15ull) == 0ull)
    __hip_atomic_store(W.progress, claimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// BVH traversal -- pine's ordered stack traversal (src/pine/impl/accel/bvh.cpp:321-451), with the
// node's two child boxes tested against the tmax captured when the node is visited, leaf children
// tested inline in stored primitive order, nearer-exit child first.  The stack lives in LDS,
// lane-interleaved ([slot][thread]) so pushes/pops are bank-conflict free.
// ------------------------------------------------------------------------------------------------
// Test hook (pine_gpu_test_traverse): the primitives a traversal tests, in order -- a top-level primitive's geometry
// index, or 0x40000000 | triangle index inside the mesh entered last.  Null in every kernel (the calls fold away).
struct TravLog {
  unsigned* words;
  int n, cap;
  __device__ __forceinline__ void put(unsigned w) {
    if (n < cap) words[n] = w;
    n++;
  }
};
// ---- PINE_GPU_FLAG_ORDER_EMBREE: arithmetic of the vendored Embree 4.3.1 as an AVX2 x86 host runs it (see scene_traverse_embree) ----
// rcp(a) (common/math/vec3fa.h:122-144, common/simd/vfloat4_sse2.h:304-318): one fused Newton step on the RCPPS estimate, which is
// the table's entry for the operand's top 11 mantissa bits scaled by its exponent (infinity for zero / denormal operands and
// results beyond the range, zero for results below the normal range)
__device__ __forceinline__ float embree_rcp(const unsigned* table, float a) {
  const unsigned u = __float_as_uint(a);
  const int e = int((u >> 23) & 0xffu);
  const unsigned t = table[(u >> 12) & 0x7ffu];  // the estimate for the mantissa in [1, 2): in (0.5, 1]
  const int re = int((t >> 23) & 0xffu) + 127 - e;
  const unsigned sign = u & 0x80000000u;
  const float r = __uint_as_float(e == 0 || re >= 255 ? (sign | 0x7f800000u) : re <= 0 ? sign : (sign | (unsigned(re) << 23) | (t & 0x7fffffu)));
  return __fmaf_rn(r, __fmaf_rn(-a, r, 1.0f), r);
}
__device__ __forceinline__ float embree_rcp_safe(const unsigned* table, float a) {
  return embree_rcp(table, fabsf(a) < 1e-18f ? 1e-18f : a);  // zero_fix: min_rcp_input (vec3fa.h:167-172)
}
// One lane of TriangleMIntersector1Moeller<4, true> (kernels/geometry/triangle_intersector_moeller.h:66-140, :29-37): meshes
// are Embree TRIANGLE geometry under EmbreeAccel (embree.cpp:76-87), Triangle4 blocks keeping v0, e1 = v0 - v1, e2 = v2 - v0
// (geometry/triangle.h); cross and dot products fused as common/math/vec3.h does.  -> t, the barycentrics and the geometric
// normal EmbreeAccel::intersect builds the surface point from (embree.cpp:233-247).  Which triangles Embree's own hierarchy
// hands to the test (spatial splits: not restated) cannot change the closest hit; only an exact tie in t is decided by it.
__device__ __forceinline__ bool embree_tri_test(const unsigned* table, const float* v, f3 o, f3 d, float tnear, float tfar, float& t, f2& uv, f3& ng) {
  const f3 v0 = ld3(v), e1 = v0 - ld3(v + 3), e2 = ld3(v + 6) - v0;
  auto crossf = [](f3 a, f3 b) { return f3{__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x))}; };
  auto dotf = [](f3 a, f3 b) { return __fmaf_rn(a.x, b.x, __fmaf_rn(a.y, b.y, a.z * b.z)); };
  ng = crossf(e2, e1);
  const f3 C = v0 - o, R = crossf(C, d);
  const float den = dotf(ng, d), abs_den = fabsf(den);
  const unsigned sgn = __float_as_uint(den) & 0x80000000u;
  const float U = __uint_as_float(__float_as_uint(dotf(R, e2)) ^ sgn), V = __uint_as_float(__float_as_uint(dotf(R, e1)) ^ sgn);
  if (!(den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= abs_den)) return false;
  const float T = __uint_as_float(__float_as_uint(dotf(ng, C)) ^ sgn);
  if (!(abs_den * tnear < T && T <= abs_den * tfar)) return false;
  const float r = embree_rcp(table, abs_den);
  t = T * r;
  uv = f2{U * r, V * r};
  return true;
}
// EmbreeAccel::intersect's surface point on a mesh (embree.cpp:233-247): position, normal and texcoord from Embree's barycentrics
// and geometric normal of the winning triangle -- recomputed here from the ray (they do not depend on tfar)
__device__ __forceinline__ void mesh_surface_info_embree(const unsigned* table, const float* tri_verts, const float* tri_attrs, int flags, int prim, f3 o, f3 d, DSurface& it) {
  const float* v = tri_verts + size_t(prim) * 9;
  float t = 0.0f;
  f2 bary{0.0f, 0.0f};
  f3 ng = mk3(0.0f);
  (void)embree_tri_test(table, v, o, d, -1.0f, __uint_as_float(0x7f800000u), t, bary, ng);  // (the hit exists: every t passes)
  it.p = lerp3(bary.x, bary.y, ld3(v), ld3(v + 3), ld3(v + 6));
  it.n = normalize(ng);
  it.uv = bary;
  if (flags != 0) {
    const float* a = tri_attrs + size_t(prim) * 16;
    if (flags & 1) it.n = normalize(lerp3(bary.x, bary.y, ld3(a), ld3(a + 3), ld3(a + 6)));
    if (flags & 2) it.uv = (1.0f - bary.x - bary.y) * f2{a[9], a[10]} + bary.x * f2{a[11], a[12]} + bary.y * f2{a[13], a[14]};
  }
}

// (EMB: Embree's triangle test instead of pine's -- the scene queries of the F_EMBREE variants; the BSSRDF walk's queries against
//  the mesh's own ShapeBVH are pine's whatever the accel)
template <bool ANY, int STRIDE = kBlock, unsigned F = 0, class StackT = int, bool EMB = false>
__device__ __forceinline__ bool mesh_traverse(const SceneView& S, const DBvh bvh, DRay& ray,
                                              const DRayOct& oct, StackT* stack, int sp0, int& prim_out, TravLog* log = nullptr) {
  bool hit = false;
  auto leaf = [&](int start, int count) -> bool {
    for (int i = start; i < start + count; i++) {
      // leaf-ordered record: v0 v1 v2 | triangle index (FlatAccel::tri_leaf, or its LDS packets)
      float v[9];
      int tri;
      fetch_triangle<F>(S, i, v, tri);
      if (log) log->put(0x40000000u | unsigned(tri - bvh.prim_base));  // (the index within its mesh, as the reference counts)
      if constexpr (EMB) {
        float t;
        f2 uv;
        f3 ng;
        if (embree_tri_test(S.rcpps, v, ray.o, ray.d, fmaxf(ray.tmin, 0.0f), ray.tmax, t, uv, ng)) {
          if (ANY) return true;
          ray.tmax = t;
          hit = true;
          prim_out = tri;
        }
      } else if (ANY) {
        if (tri_hit(v, ray)) return true;
      } else if (tri_intersect(v, ray)) {
        hit = true;
        prim_out = tri;
      }
    }
    return false;
  };
  if (bvh.root_count > 0) {
    if (leaf(bvh.root_start, bvh.root_count)) return true;
    return hit;
  }
  int sp = sp0;
  int next = bvh.root;
  while (true) {
    DNode nd_mesh;
    const DNode* node = &S.nodes[next];
    if constexpr (F & F_LDS_TOP) {
      nd_mesh = fetch_node<F>(S, next);
      node = &nd_mesh;
    }
    int l = -1, r = -1;
    float t0 = ray.tmax, t1 = ray.tmax;
    if (box_hit_oct(node->lo0, node->hi0, oct, ray.tmin, t0)) {
      if (node->count[0] == 0) l = node->child[0];
      else if (leaf(node->child[0], node->count[0])) return true;
    }
    if (box_hit_oct(node->lo1, node->hi1, oct, ray.tmin, t1)) {
      if (node->count[1] == 0) r = node->child[1];
      else if (leaf(node->child[1], node->count[1])) return true;
    }
    if (l != -1) {
      if (r != -1) {
        if (t0 > t1) {
          stack[sp * STRIDE] = StackT(l);
          next = r;
        } else {
          stack[sp * STRIDE] = StackT(r);
          next = l;
        }
        sp++;
      } else next = l;
    } else if (r != -1) next = r;
    else {
      if (sp == sp0) break;
      next = int(stack[(--sp) * STRIDE]);
    }
  }
  return hit;
}

// PINE_GPU_FLAG_ORDER_EMBREE (F_EMBREE variants): closest-hit queries hand the non-mesh shapes to their tests in the order the
// reference's DEFAULT accel does -- BVHNIntersector1<8, BVH_AN1, false, ...>::intersect of the vendored Embree 4.3.1
// (src/contrib/embree/kernels/bvh/bvh_intersector1.cpp:30-107) over the BVH8 of pine_embree_order.h, as an AVX2 x86 host runs it:
//   * the ray: rdir = rcp_safe(dir) -- one fused Newton step on the RCPPS estimate (common/math/vec3fa.h:122-172; the estimates
//     are the table of pine_amd/data/rcpps_table.h) -- and org_rdir = org * rdir (kernels/bvh/node_intersector1.h:26-60);
//   * a node: per child fused plane * rdir - org_rdir, maximum / minimum over the float words compared as INTEGERS, entered when
//     not tNear > tFar (node_intersector1.h:484-530);
//   * its hit children, in slot order: one -> descend; two -> the nearer first (equal: the second); three / four -> the sorting
//     networks of common/stack_item.h:54-84; more -> the stable descending insertion sort (:88-104); the nearest is descended
//     into, the others wait on the stack with their distances (kernels/bvh/bvh_traverser1.h:310-385);
//   * a popped entry whose distance lies beyond the closest hit so far is dropped (bvh_intersector1.cpp:77-79).
// Meshes are Embree triangle geometry: they are asked FIRST (the triangle accel precedes the user-geometry accel,
// kernels/common/scene.cpp:741-755), through Embree's own triangle test (embree_tri_test above; pine's per-mesh BVH only decides
// which triangles are looked at).  tests/test_embree_order.py, tests/test_gpu_parity.py: the films of the real reference built
// with EmbreeAccel, bit for bit.
template <unsigned F, int STRIDE, class StackT>
__device__ __forceinline__ bool scene_traverse_embree(const SceneView& S, DRay& ray, StackT* stack, int& geom_out, int& prim_out, TravLog* log) {
  bool hit = false;
  auto test_leaf = [&](int place) {
    const DShape* sh = &S.leaf[place];
    DShape rec;
    {
      const uint4* src = reinterpret_cast<const uint4*>(sh);
      uint4* dst = reinterpret_cast<uint4*>(&rec);
#pragma unroll
      for (int q = 0; q < 8; q++) dst[q] = src[q];
      sh = &rec;
    }
    const int word = sh->kind;  // (the packed word rides in the copy's kind field)
    const int kind = word >> kPrimKindShift;
    bool is_mesh = false;
    if constexpr (F & F_MESH) is_mesh = kind == SHAPE_MESH;
    if (log) log->put(unsigned(word & kPrimIndexMask));
    if (is_mesh) {
      if constexpr (F & F_MESH) {
        const DRayOct oct = make_oct(ray);
        int prim = 0;
        if (mesh_traverse<false, STRIDE, F, StackT, true>(S, S.bvhs[as_int(sh->f[2])], ray, oct, stack, S.stack_top, prim, log)) {
          hit = true;
          geom_out = word;
          prim_out = prim;
        }
      }
    } else if (shape_intersect<F>(kind, sh, ray)) {
      hit = true;
      geom_out = word;
    }
  };
  if constexpr (F & F_MESH)
    for (int k = 0; k < S.num_emesh; k++) test_leaf(S.emesh[k]);
  if (S.etree_root == kEmbreeNoChild) return hit;
  const float rdx = embree_rcp_safe(S.rcpps, ray.d.x), rdy = embree_rcp_safe(S.rcpps, ray.d.y), rdz = embree_rcp_safe(S.rcpps, ray.d.z);
  const float ordx = ray.o.x * rdx, ordy = ray.o.y * rdy, ordz = ray.o.z * rdz;
  // (near plane: the lower one where rdir >= 0)
  const int nx = rdx >= 0.0f ? 0 : 24, ny = rdy >= 0.0f ? 0 : 24, nz = rdz >= 0.0f ? 0 : 24;
  const int tnear = __float_as_int(fmaxf(ray.tmin, 0.0f));
  int tfar = __float_as_int(fmaxf(ray.tmax, 0.0f));
  int2 items[kEmbreeStackEntries];  // (child word, distance word)
  int sp = 1;
  items[0] = int2{S.etree_root, int(0xff800000u)};  // (distance -inf)
  while (sp > 0) {
    sp--;
    int cur = items[sp].x;
    if (__int_as_float(items[sp].y) > ray.tmax) continue;
    bool dropped = false;
    while (cur >= 0) {
      const float* nd = reinterpret_cast<const float*>(&S.etree[cur]);
      const int count = reinterpret_cast<const int*>(nd)[56];
      const int first = sp;
      // four children at a time: the six planes of each as quads (near / far side chosen by the quad's address), the child words
      // as one more; unused slots are never entered -- their planes are infinite -- and a second half without children is skipped
      for (int half = 0; half < 8 && half < count; half += 4) {
        const float4 nxq = *reinterpret_cast<const float4*>(nd + nx + half), nyq = *reinterpret_cast<const float4*>(nd + 8 + ny + half),
                     nzq = *reinterpret_cast<const float4*>(nd + 16 + nz + half);
        const float4 fxq = *reinterpret_cast<const float4*>(nd + (nx ^ 24) + half), fyq = *reinterpret_cast<const float4*>(nd + 8 + (ny ^ 24) + half),
                     fzq = *reinterpret_cast<const float4*>(nd + 16 + (nz ^ 24) + half);
        const int4 ch = *reinterpret_cast<const int4*>(nd + 48 + half);
        auto one = [&](float px, float py, float pz, float qx, float qy, float qz, int child) {
          const int tn = max(max(__float_as_int(__fmaf_rn(px, rdx, -ordx)), __float_as_int(__fmaf_rn(py, rdy, -ordy))),
                             max(__float_as_int(__fmaf_rn(pz, rdz, -ordz)), tnear));
          const int tf = min(min(__float_as_int(__fmaf_rn(qx, rdx, -ordx)), __float_as_int(__fmaf_rn(qy, rdy, -ordy))),
                             min(__float_as_int(__fmaf_rn(qz, rdz, -ordz)), tfar));
          if (!(tn > tf)) items[sp++] = int2{child, tn};
        };
        one(nxq.x, nyq.x, nzq.x, fxq.x, fyq.x, fzq.x, ch.x);
        one(nxq.y, nyq.y, nzq.y, fxq.y, fyq.y, fzq.y, ch.y);
        one(nxq.z, nyq.z, nzq.z, fxq.z, fyq.z, fzq.z, ch.z);
        one(nxq.w, nyq.w, nzq.w, fxq.w, fyq.w, fzq.w, ch.w);
      }
      const int hits = sp - first;
      if (hits == 0) {
        dropped = true;
        break;
      }
      auto order = [&](int a, int b) {  // cmp_xchg: items[a] <= items[b] afterwards
        if (items[b].y < items[a].y) {
          const int2 t = items[a];
          items[a] = items[b], items[b] = t;
        }
      };
      int2* h = items + first;
      if (hits == 2) {
        if (unsigned(h[0].y) < unsigned(h[1].y)) {
          const int2 t = h[0];
          h[0] = h[1], h[1] = t;
        }
      } else if (hits == 3) {
        order(first + 1, first), order(first + 2, first + 1), order(first + 1, first);
      } else if (hits == 4) {
        order(first + 1, first), order(first + 3, first + 2), order(first + 2, first), order(first + 3, first + 1), order(first + 2, first + 1);
      } else if (hits > 4) {
        for (int i = 1; i < hits; i++) {
          const int2 item = h[i];
          int j = i;
          while (j > 0 && unsigned(h[j - 1].y) < unsigned(item.y)) h[j] = h[j - 1], j--;
          h[j] = item;
        }
      }
      cur = items[--sp].x;  // the nearest; the others wait
    }
    if (dropped) continue;
    test_leaf(~cur);  // an Object leaf: the user callback (embree.cpp:24-40)
    tfar = __float_as_int(ray.tmax);
  }
  return hit;
}

// ... and its any-hit query, EmbreeAccel::hit (embree.cpp:143-165) = BVHNIntersector1<8, ...>::occluded
// (bvh_intersector1.cpp:117-195).  The order cannot change an any-hit answer; WHICH shapes are asked can: a shape is asked
// exactly when the ray enters its own box (and its ancestors') within [tnear, tfar], where pine's BVH asks every shape of a
// leaf whose UNION box is entered -- a Plane beyond its +-100 bounds (geometry.cpp:52) is found by the one and not by the other.
// And a query that starts with a negative tfar is answered "occluded" (rtcOccluded1 leaves such a ray alone, and
// embree.cpp:164 returns `tfar < 0`), where pine's BVH finds nothing: light samples with a negative distance reach this.
template <unsigned F, int STRIDE, class StackT>
__device__ __forceinline__ bool scene_occluded_embree(const SceneView& S, const DRay& ray_in, StackT* stack, TravLog* log) {
  if (ray_in.tmax < 0.0f) return true;
  DRay ray = ray_in;
  if constexpr (F & F_MESH)
    for (int k = 0; k < S.num_emesh; k++) {
      const DShape* sh = &S.leaf[S.emesh[k]];
      if (log) log->put(unsigned(sh->kind & kPrimIndexMask));
      const DRayOct oct = make_oct(ray);
      int prim = 0;
      if (mesh_traverse<true, STRIDE, F, StackT, true>(S, S.bvhs[as_int(sh->f[2])], ray, oct, stack, S.stack_top, prim, log)) return true;
    }
  if (S.etree_root == kEmbreeNoChild) return false;
  const float rdx = embree_rcp_safe(S.rcpps, ray.d.x), rdy = embree_rcp_safe(S.rcpps, ray.d.y), rdz = embree_rcp_safe(S.rcpps, ray.d.z);
  const float ordx = ray.o.x * rdx, ordy = ray.o.y * rdy, ordz = ray.o.z * rdz;
  const int nx = rdx >= 0.0f ? 0 : 24, ny = rdy >= 0.0f ? 0 : 24, nz = rdz >= 0.0f ? 0 : 24;
  const int tnear = __float_as_int(fmaxf(ray.tmin, 0.0f)), tfar = __float_as_int(fmaxf(ray.tmax, 0.0f));
  int items[kEmbreeStackEntries];
  int sp = 1;
  items[0] = S.etree_root;
  while (sp > 0) {
    const int cur = items[--sp];
    if (cur < 0) {
      const DShape* sh = &S.leaf[~cur];
      DShape rec;
      {
        const uint4* src = reinterpret_cast<const uint4*>(sh);
        uint4* dst = reinterpret_cast<uint4*>(&rec);
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = src[q];
        sh = &rec;
      }
      const int word = sh->kind;
      if (log) log->put(unsigned(word & kPrimIndexMask));
      if (shape_hit<F>(word >> kPrimKindShift, sh, ray)) return true;
      continue;
    }
    const float* nd = reinterpret_cast<const float*>(&S.etree[cur]);
    const int count = reinterpret_cast<const int*>(nd)[56];
    for (int half = 0; half < 8 && half < count; half += 4) {  // (quads of planes, as in the closest-hit query)
      const float4 nxq = *reinterpret_cast<const float4*>(nd + nx + half), nyq = *reinterpret_cast<const float4*>(nd + 8 + ny + half),
                   nzq = *reinterpret_cast<const float4*>(nd + 16 + nz + half);
      const float4 fxq = *reinterpret_cast<const float4*>(nd + (nx ^ 24) + half), fyq = *reinterpret_cast<const float4*>(nd + 8 + (ny ^ 24) + half),
                   fzq = *reinterpret_cast<const float4*>(nd + 16 + (nz ^ 24) + half);
      const int4 ch = *reinterpret_cast<const int4*>(nd + 48 + half);
      auto one = [&](float px, float py, float pz, float qx, float qy, float qz, int child) {
        const int tn = max(max(__float_as_int(__fmaf_rn(px, rdx, -ordx)), __float_as_int(__fmaf_rn(py, rdy, -ordy))),
                           max(__float_as_int(__fmaf_rn(pz, rdz, -ordz)), tnear));
        const int tf = min(min(__float_as_int(__fmaf_rn(qx, rdx, -ordx)), __float_as_int(__fmaf_rn(qy, rdy, -ordy))),
                           min(__float_as_int(__fmaf_rn(qz, rdz, -ordz)), tfar));
        if (!(tn > tf)) items[sp++] = child;
      };
      one(nxq.x, nyq.x, nzq.x, fxq.x, fyq.x, fzq.x, ch.x);
      one(nxq.y, nyq.y, nzq.y, fxq.y, fyq.y, fzq.y, ch.y);
      one(nxq.z, nyq.z, nzq.z, fxq.z, fyq.z, fzq.z, ch.z);
      one(nxq.w, nyq.w, nzq.w, fxq.w, fyq.w, fzq.w, ch.w);
    }
  }
  return false;
}

// ANY: BVH::hit (bvh.cpp:497-511).  !ANY: BVH::intersect (bvh.cpp:513-548) minus the final
// compute_surface_info, which the caller does once for the winning primitive.
// geom_out receives the winning primitive's PACKED word (index | emissive bit | kind).
template <bool ANY, unsigned F, int STRIDE = kBlock, class StackT = int>
__device__ __forceinline__ bool scene_traverse(const SceneView& S, DRay& ray, StackT* stack, int& geom_out,
                                               int& prim_out, TravLog* log = nullptr) {
  if (S.num_shapes == 0) return false;
  if constexpr ((F & F_EMBREE) != 0) {
    if constexpr (ANY) return scene_occluded_embree<F, STRIDE>(S, ray, stack, log);
    else return scene_traverse_embree<F, STRIDE>(S, ray, stack, geom_out, prim_out, log);
  }
  const DRayOct oct = make_oct(ray);
  const DBvh top = S.bvhs[0];
  bool hit = false;
  auto leaf = [&](int start, int count) -> bool {
    for (int i = start; i < start + count; i++) {
      REGION(ANY ? 5 : 2);  // leaf primitive test
      const DShape* sh = &S.leaf[i];
      // the whole record in one batch of loads before the kind is looked at (one round trip instead of kind, then the
      // kind's fields: C2 13.20 -> 13.06 ms; see pine_trav.h)
      DShape rec;
      {
        const uint4* src = reinterpret_cast<const uint4*>(sh);
        uint4* dst = reinterpret_cast<uint4*>(&rec);
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = src[q];
        sh = &rec;
      }
      const int word = sh->kind;  // (the packed word rides in the copy's kind field)
      const int kind = word >> kPrimKindShift;
      bool is_mesh = false;
      if constexpr (F & F_MESH) is_mesh = kind == SHAPE_MESH;
      if (log) log->put(unsigned(word & kPrimIndexMask));
      if (is_mesh) {
        if constexpr (F & F_MESH) {
          const DBvh mb = S.bvhs[as_int(sh->f[2])];
          int prim = 0;
          const bool h = mesh_traverse<ANY, STRIDE, F>(S, mb, ray, oct, stack, S.stack_top, prim, log);
          if (ANY) {
            if (h) return true;
          } else if (h) {
            hit = true;
            geom_out = word;
            prim_out = prim;
          }
        }
      } else if (ANY) {
        if (shape_hit<F>(kind, sh, ray)) return true;
      } else if (shape_intersect<F>(kind, sh, ray)) {
        hit = true;
        geom_out = word;
      }
#ifdef PINE_DUP_SHAPES  /* cost-measurement builds only: run the selected shape tests a second time on an opaque copy of the ray */
      {
        const bool sel = PINE_DUP_SHAPES == 0 ? kind == SHAPE_RECT : kind == SHAPE_OBB;
        if (sel) {
          DRay rr = ray;
          asm volatile("" : "+v"(rr.tmin));
          const bool h2 = ANY ? shape_hit<F>(kind, sh, rr) : shape_intersect<F>(kind, sh, rr);
          float sink = h2 ? rr.tmax : 0.0f;
          asm volatile("" : : "v"(sink));
        }
      }
#endif
    }
    return false;
  };
  if (top.root_count > 0) {
    if (leaf(top.root_start, top.root_count)) return true;
    return hit;
  }
  if (top.root < 0) return false;  // geometries exist but none has primitives (only empty meshes): nothing to visit
  int sp = 0;
  int next = top.root;
  while (true) {
    REGION(ANY ? 4 : 1);  // top-level node visit
    DNode nd_top;
    const DNode* node = &S.nodes[next];
    if constexpr (F & F_LDS_TOP) {
      nd_top = fetch_node<F>(S, next);
      node = &nd_top;
    }
    int l = -1, r = -1;
    float t0 = ray.tmax, t1 = ray.tmax;
#ifdef PINE_DUP_NODES
    {
      float q0 = ray.tmax, q1 = ray.tmax, tm = ray.tmin;
      asm volatile("" : "+v"(tm));
      const bool b0 = box_hit_oct(node->lo0, node->hi0, oct, tm, q0);
      const bool b1 = box_hit_oct(node->lo1, node->hi1, oct, tm, q1);
      float sink = (b0 ? q0 : 0.0f) + (b1 ? q1 : 0.0f);
      asm volatile("" : : "v"(sink));
    }
#endif
    if (box_hit_oct(node->lo0, node->hi0, oct, ray.tmin, t0)) {
      if (node->count[0] == 0) l = node->child[0];
      else if (leaf(node->child[0], node->count[0])) return true;
    }
    if (box_hit_oct(node->lo1, node->hi1, oct, ray.tmin, t1)) {
      if (node->count[1] == 0) r = node->child[1];
      else if (leaf(node->child[1], node->count[1])) return true;
    }
    if (l != -1) {
      if (r != -1) {
        if (t0 > t1) {
          stack[sp * STRIDE] = StackT(l);
          next = r;
        } else {
          stack[sp * STRIDE] = StackT(r);
          next = l;
        }
        sp++;
      } else next = l;
    } else if (r != -1) next = r;
    else {
      if (sp == 0) break;
      next = int(stack[(--sp) * STRIDE]);
    }
  }
  return hit;
}

// ------------------------------------------------------------------------------------------------
// Item <-> pixel mapping.  Items are ordered [local tile][chunk][pixel in tile] so that the 64
// consecutive items a fresh wave pulls are one 8x8 tile at one sample range: coherent rays,
// contiguous sampler-tile bytes, contiguous sample-buffer rows.
// ------------------------------------------------------------------------------------------------
struct ItemInfo {
  int px, py;
  int chunk;
  unsigned long long sample_base;  // index of sample 0 of this pixel row in the samples buffer / 64-strided
  unsigned long long ckpt_index;   // the item's RNG checkpoint (items of the independent class)
  bool serial;                     // an item of the whole-pixel class of a launch with tile classes
  bool valid;
};
__device__ __forceinline__ int film_tile_of(const WorkParams& W, int ltile) {
  return W.tile_order ? W.tile_order[ltile] : ltile * W.shard_world + W.shard_rank;
}
__device__ __forceinline__ ItemInfo decode_item(const WorkParams& W, int film_w, int film_h, int spp,
                                                unsigned long long item) {
  ItemInfo it;
  const unsigned long long serial_items = (unsigned long long)W.serial_tiles * 64ull;
  it.serial = item < serial_items;
  const unsigned long long rel = it.serial ? item : item - serial_items;
  const int p = int(rel & 63);
  const unsigned long long tc = rel >> 6;
  // spp and k are powers of two: shifts instead of 64-bit divisions (this runs once per camera sample)
  const int chunk = it.serial ? 0 : int(tc & (unsigned long long)(W.items_per_pixel - 1));
  const int ltile = it.serial ? int(tc) : W.serial_tiles + int(tc >> W.log2_items_per_pixel);
  const int tile = film_tile_of(W, ltile);
  it.ckpt_index = rel;
  // tile / tiles_x by multiplication with the rounded-up reciprocal + one fix-up step (exact for any
  // 32-bit tile: the estimate is never more than one too large)
  unsigned ty = unsigned((uint64_t(unsigned(tile)) * W.tiles_x_magic) >> 32);
  if (ty * unsigned(W.tiles_x) > unsigned(tile)) ty--;
  if (W.tiles_x == 1) ty = unsigned(tile);  // (2^32 / 1 does not fit the 32-bit magic)
  const int tx = tile - int(ty) * W.tiles_x;
  it.px = tx * kTile + (p & 7);
  it.py = ty * kTile + (p >> 3);
  it.chunk = chunk;
  it.sample_base = (unsigned long long)ltile * (unsigned)spp * 64ull + (unsigned)p;
  it.valid = it.px < film_w && it.py < film_h;
  return it;
}

// RNG state at the start of every item: the reference reseeds per pixel (sampler.h:286-290) and
// then draws 4 floats per camera sample (path.cpp:35); when nothing inside radiance() touches the
// RNG the state at sample s is the seed advanced 4*s steps.
static __global__ void __launch_bounds__(kBlock) rng_checkpoint_kernel(WorkParams W, int film_w, int film_h, int spp,
                                                               ulonglong2* ckpt) {
  // one thread per (local tile of the independent class, pixel in tile); walks the whole pixel, storing at chunk starts
  const unsigned long long t = blockIdx.x * (unsigned long long)kBlock + threadIdx.x;
  const unsigned long long n = (unsigned long long)(W.num_local_tiles - W.serial_tiles) * 64ull;
  if (t >= n) return;
  const int p = int(t & 63);
  const int ptile = int(t >> 6);  // (position among the independent tiles: ItemInfo::ckpt_index counts from there)
  const int tile = film_tile_of(W, W.serial_tiles + ptile);
  const int px = (tile % W.tiles_x) * kTile + (p & 7), py = (tile / W.tiles_x) * kTile + (p >> 3);
  DRng g = rng_seed(hash_pixel(px, py, 0));
  for (int c = 0; c < W.items_per_pixel; c++) {
    const unsigned long long item = ((unsigned long long)ptile * W.items_per_pixel + c) * 64ull + p;
    ckpt[item] = make_ulonglong2(g.s0, g.s1);
    for (int i = 0; i < 4 * W.samples_per_item; i++) rng_next64(g);
  }
  (void)film_w, (void)film_h, (void)spp;
}

// ------------------------------------------------------------------------------------------------
// The path kernel
// ------------------------------------------------------------------------------------------------
struct Counters {
  unsigned long long next_item;
  unsigned long long vertices;
  unsigned long long shadow_rays;
  // protocol failure of the stage-queued kernel (a bounded spin or the idle budget ran out): number of
  // bail-outs of the launch, and the code / operands of one of them.  Read by every host entry point
  // that synchronises (plan_check): a launch with bail_count != 0 has an incomplete film and FAILS.
  unsigned long long bail_count;
  unsigned long long bail_code, bail_a, bail_b;
  unsigned long long walk_steps;  // BSSRDF random-walk steps (stage-queued kernel, F_SSS variants)
  unsigned long long section_cycles[16];  // diagnostic builds (-DPINE_PROFILE_SECTIONS) only
  unsigned long long t_start, t_pool_dry, t_end;  // ... 100 MHz wall clock: first workgroup in, the work-item pool found empty, last workgroup out
#ifdef PINE_PROFILE_SECTIONS
  unsigned long long wg_t[1024][4];  // per workgroup: its last whole-pixel item sealed | pool found dry | out | whole-pixel items claimed
#endif
};



__device__ __forceinline__ f3 material_le(const DMaterial* m, f3 n, f3 wo) {  // material.h:22-25
  if (m->kind != MAT_EMISSIVE) return mk3(0.0f);
  if (dot(wo, n) < 0.0f) return mk3(0.0f);
  return ld3(m->color);
}

// LDS layout of the path kernel (dword offsets; everything per-lane is [slot][thread], so a wave's
// accesses are bank-conflict free and one VGPR (thread id) + an immediate offset addresses all of it):
//   fold level(s)   kLdsFoldLevels * 8 x 256      FoldEntry fields of the shallowest level(s)
//   sampler slices  12 x 256                      40 ranking + 8 scrambling bytes of the lane's pixel
//   RNG state       4 x 256                       per-pixel xoroshiro state (only touched at sample start)
//   Sobol rows      40 x 256 bytes                transposed table, dimensions < 40
//   traversal stack stack_total x 256             (runtime depth)
//   scene blob      blob_bytes                    nodes | shapes | materials | bvhs | prims | lights
constexpr int kOffFold = 0;
constexpr int kOffTile = kLdsFoldLevels * 8 * kBlock;
constexpr int kOffRng = kOffTile + kLdsTileDwords * kBlock;
constexpr int kOffSobol = kOffRng + 4 * kBlock;
constexpr int kOffStack = kOffSobol + kLdsSamplerDims * 256 / 4;
constexpr size_t kLdsFixedBytes = size_t(kOffStack) * 4;

// Packed per-lane path bookkeeping (one VGPR):
//   bits 0-11 sample index within the pixel (BlueSobolSampler::index / the low part of SobolSampler's index),
//   12-20 sampler dimension, 21-26 Vertex::length, 27 Vertex::diffuse_length > 0 (all the path reads of it,
//   path.cpp:93), 28 Vertex::is_delta, 29-30 the stage-queued kernel's BSSRDF walk status of the vertex being shaded
//   (kWalk*), 31 "sealed" (stage-queued kernel, Subsurface variants: the path has released its pixel's sample token, see
//   pine_queue_kernel.h).  0xffffffff marks an empty context (a length of 63 cannot occur: kMaxDepth is 32).
constexpr int kMaxDeviceSpp = 4096;      // 12 bits of sample index
constexpr int kMaxSamplerDimension = 511;  // 9 bits: BlueSampler wraps at 256; SobolSampler counts up to 8 draws per vertex
enum : unsigned { kWalkNone = 0, kWalkRunning = 1, kWalkExited = 2, kWalkFailed = 3 };
struct PackedState {
  unsigned v;
  __device__ __forceinline__ unsigned walk() const { return (v >> 29) & 3u; }
  __device__ __forceinline__ void set_walk(unsigned w) { v = (v & ~(3u << 29)) | (w << 29); }
  __device__ __forceinline__ int s_cur() const { return int(v & 0xfffu); }
  __device__ __forceinline__ int dim() const { return int((v >> 12) & 0x1ffu); }
  __device__ __forceinline__ int length() const { return int((v >> 21) & 0x3fu); }
  __device__ __forceinline__ int diffuse_length() const { return int((v >> 27) & 1u); }  // 0 or "at least 1"
  __device__ __forceinline__ bool is_delta() const { return (v >> 28) & 1u; }
  __device__ __forceinline__ void set_dim(int d) { v = (v & ~(0x1ffu << 12)) | (unsigned(d) << 12); }
  __device__ __forceinline__ void start_sample(int s) { v = unsigned(s) | (1u << 28); }  // dim 0, first_vertex()
  __device__ __forceinline__ void next_vertex(bool delta) {  // Vertex(pv, pdf, is_delta) path.cpp:18-19
    v = (v & 0x801fffffu) + ((unsigned(length()) + 1u) << 21) + (((v >> 27) & 1u) | (delta ? 0u : 1u)) * (1u << 27) +
        (delta ? (1u << 28) : 0u);
  }
  __device__ __forceinline__ bool sealed() const { return (v >> 31) != 0u; }
  __device__ __forceinline__ void set_sealed() { v |= 0x80000000u; }
};

template <unsigned F, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(kBlock, WAVES_PER_SIMD)
path_trace_kernel(DeviceScene S, WorkParams W, const ulonglong2* __restrict__ ckpt, float4* __restrict__ samples,
                  float* __restrict__ fold, Counters* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  constexpr int kSM = kSmLds | ((F & F_SOBOL) ? kSmSobol : 0);  // sampler front mode (pine_device.h)
  const unsigned tid = threadIdx.x;
  float* const lds_f = reinterpret_cast<float*>(lds_raw);
  uint32_t* const lds_u = reinterpret_cast<uint32_t*>(lds_raw);
  int* const stack = lds_raw + kOffStack + tid;
  {
    const uint4* src = reinterpret_cast<const uint4*>(S.tables.sobol);
    uint4* dst = reinterpret_cast<uint4*>(lds_raw + kOffSobol);
    for (int i = tid; i < kLdsSamplerDims * 256 / 16; i += kBlock) dst[i] = src[i];
  }
  DTables T = S.tables;
  T.lds_sobol = reinterpret_cast<const uint8_t*>(lds_raw + kOffSobol);
  T.lds_tile = lds_u + kOffTile + tid;
  T.lds_scr = lds_u + kOffTile + tid + 10 * kLdsLaneStride;
  T.tile_stride = kLdsLaneStride;
  T.win_lo = 0;
  T.win_len = kLdsSamplerDims;
  SceneView V;
  V.tri_verts = S.tri_verts;
  V.tri_leaf = S.tri_leaf;
  V.tri_attrs = S.tri_attrs;
  V.lds_nodes = nullptr;
  V.lds_node_count = 0;
  V.lds_tri_entries = nullptr;
  V.lds_tri_verts = nullptr;
  V.stack_top = S.stack_top;
  V.num_shapes = S.num_shapes;
  V.etree_root = S.etree_root;
  V.num_emesh = S.num_emesh;
  V.rcpps = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(S.blob) + S.off_rcpps);
  if constexpr (F & F_LDS_SCENE) {
    uint4* dst = reinterpret_cast<uint4*>(lds_raw + kOffStack + S.stack_total * kBlock);
    const int n16 = S.blob_bytes >> 4;
    for (int i = tid; i < n16; i += kBlock) dst[i] = S.blob[i];
    __syncthreads();
    const char* base = reinterpret_cast<const char*>(dst);
    V.nodes = reinterpret_cast<const DNode*>(base + S.off_nodes);
    V.shapes = reinterpret_cast<const DShape*>(base + S.off_shapes);
    V.materials = reinterpret_cast<const DMaterial*>(base + S.off_materials);
    V.bvhs = reinterpret_cast<const DBvh*>(base + S.off_bvhs);
    V.prims = nullptr;
    V.lights = reinterpret_cast<const DLight*>(base + S.off_lights);
    V.node_ops = reinterpret_cast<const DNodeOp*>(base + S.off_node_ops);
    V.leaf = reinterpret_cast<const DShape*>(base + S.off_leaf) - S.top_prim_begin;
    V.etree = reinterpret_cast<const EmbreeNode*>(base + S.off_etree);
    V.emesh = reinterpret_cast<const int*>(base + S.off_emesh);
    if (S.off_rcpps < S.blob_bytes) V.rcpps = reinterpret_cast<const unsigned*>(base + S.off_rcpps);
  } else {
    __syncthreads();  // Sobol rows staged above
    V.etree = reinterpret_cast<const EmbreeNode*>(reinterpret_cast<const char*>(S.blob) + S.off_etree);
    V.emesh = reinterpret_cast<const int*>(reinterpret_cast<const char*>(S.blob) + S.off_emesh);
    V.leaf = S.leaf;
    V.nodes = S.nodes;
    V.shapes = S.shapes;
    V.materials = S.materials;
    V.bvhs = S.bvhs;
    V.prims = nullptr;
    V.lights = S.lights;
    V.node_ops = S.node_ops;
  }
  // Global part of the fold stack: lane-major, one 32-byte entry (two float4) per level, so the
  // bytes a lane touches are only the levels its paths really reach -- the hot set (~2.6 levels x
  // 32 B x resident lanes ~ 22 MB chip-wide, 2.7 MB per XCD) stays in the XCD's 4 MB L2, whereas a
  // [level][field][lane] layout touches all levels of all lanes (58 MB) and thrashes it.
  auto fold_entry = [&](int level) -> float4* {
    return reinterpret_cast<float4*>(fold) + (size_t(blockIdx.x * kBlock + tid) * size_t(S.max_path_length) + size_t(level)) * 2;
  };
  auto fold_store = [&](int level, const float (&e)[8]) {
    if (level < kLdsFoldLevels) {
#pragma unroll
      for (int i = 0; i < 8; i++) lds_f[kOffFold + (level * 8 + i) * kBlock + tid] = e[i];
    } else {
      float4* q = fold_entry(level);
      q[0] = make_float4(e[0], e[1], e[2], e[3]);
      q[1] = make_float4(e[4], e[5], e[6], e[7]);
    }
  };
  auto fold_load = [&](int level, float (&e)[8]) {
    if (level < kLdsFoldLevels) {
#pragma unroll
      for (int i = 0; i < 8; i++) e[i] = lds_f[kOffFold + (level * 8 + i) * kBlock + tid];
    } else {
      const float4* q = fold_entry(level);
      const float4 a = q[0], b = q[1];
      e[0] = a.x, e[1] = a.y, e[2] = a.z, e[3] = a.w, e[4] = b.x, e[5] = b.y, e[6] = b.z, e[7] = b.w;
    }
  };
  // the per-pixel RNG lives in LDS: it is only touched when a sample starts (4 draws, path.cpp:35)
  // and by the few material branches that draw from it inside radiance()
  auto rng_load = [&]() -> DRng {
    const uint32_t a = lds_u[kOffRng + tid], b = lds_u[kOffRng + kBlock + tid], c = lds_u[kOffRng + 2 * kBlock + tid],
                   d = lds_u[kOffRng + 3 * kBlock + tid];
    return DRng{uint64_t(a) | (uint64_t(b) << 32), uint64_t(c) | (uint64_t(d) << 32)};
  };
  auto rng_store = [&](const DRng& g) {
    lds_u[kOffRng + tid] = uint32_t(g.s0);
    lds_u[kOffRng + kBlock + tid] = uint32_t(g.s0 >> 32);
    lds_u[kOffRng + 2 * kBlock + tid] = uint32_t(g.s1);
    lds_u[kOffRng + 3 * kBlock + tid] = uint32_t(g.s1 >> 32);
  };

  // ---- lane state (kept small on purpose: the kernel sits at the 128-VGPR / 4-waves-per-SIMD edge) ----
  bool lane_done = false;  // queue exhausted for this lane
  bool have_item = false;
  bool alive = false;      // a path is in flight
  f3 ray_o = mk3(0.0f), ray_d = mk3(0.0f);
  float ray_tmax = 0.0f;   // every ray on this path has tmin == 0
  unsigned pxy = 0;        // px | py << 16
  unsigned sample_base = 0;
  PackedState st{0};
  unsigned shadow_count = 0;
  unsigned long long beta_flags = 0;  // 2 bits per level (BSSRDF beta channel); dead code without F_SSS
  // SobolSampler / HaltonSampler in a scene with Subsurface: a BSSRDF walk draws three dimensions per step and has no bound on
  // its steps, SobolSampler's dimension counter does not wrap (sampler.h:143-155) and HaltonSampler's wraps at 1000 -- more
  // than the nine bits of the packed state hold.  These variants keep the counter in a register of its own.
  constexpr bool kBigDim = (F & F_SSS) != 0 && (F & F_SOBOL) != 0;
  int big_dim = 0;
  (void)big_dim;
  // wave-uniform private item pool [pool_next, pool_end)
  unsigned long long pool_next = 0, pool_end = 0;
  bool queue_empty = false;
  const int kspi = W.samples_per_item;

  SEC_DECL;
  while (true) {
    SEC_MARK(0);  // loop overhead
    // ---------------- regeneration ----------------
    // Lanes whose item is exhausted take the next items of the wave's private pool (a range of
    // kPoolItems consecutive items claimed from the global queue with ONE atomic by one lane);
    // ranks inside the wave come from a ballot prefix count, so there is no per-lane atomic.
    // (A per-iteration wave-aggregated atomic on one word saturates at ~90 M dequeues/s chip-wide,
    // MI355X_MICROARCH.md "dequeue" -- that was the first bottleneck measured.)
    {
      bool need_item = !alive && !lane_done && !have_item;
      while (true) {
        const unsigned long long mask = __ballot(need_item);
        if (mask == 0) break;
        if (pool_next == pool_end) {
          if (queue_empty) {
            if (need_item) lane_done = true;
            break;
          }
          unsigned long long base = 0;
          if ((tid & 63) == 0) base = atomicAdd(&counters->next_item, (unsigned long long)kPoolItems);
          base = __shfl(base, 0);
          if (base >= W.total_items) {
            queue_empty = true;
          } else {
            if ((tid & 63) == 0) post_progress(W, base, 9);
            pool_next = base;
            pool_end = base + kPoolItems < W.total_items ? base + kPoolItems : W.total_items;
          }
          continue;
        }
        const unsigned lane = tid & 63;
        const unsigned rank = __popcll(mask & ((1ull << lane) - 1ull));
        const unsigned long long avail = pool_end - pool_next;
        const unsigned want = __popcll(mask);
        const unsigned take = want < avail ? want : unsigned(avail);
        if (need_item && rank < take) {
          const unsigned long long item = pool_next + rank;
          need_item = false;
          const ItemInfo it = decode_item(W, S.cam.W, S.cam.H, S.spp, item);
          if (it.valid) {
            have_item = true;
            pxy = unsigned(it.px) | (unsigned(it.py) << 16);
            st.start_sample(it.chunk * kspi);
            sample_base = unsigned(it.sample_base);
            {
              // refresh this lane's sampler slice: 40 ranking bytes + 8 scrambling bytes of the pixel
              const int pix = (it.px & 127) + (it.py & 127) * 128;
              const uint2* rsrc = reinterpret_cast<const uint2*>(S.tables.rank + size_t(pix) * 8);
              const uint2 sc = *reinterpret_cast<const uint2*>(S.tables.scramble + size_t(pix) * 8);
#pragma unroll
              for (int j = 0; j < 5; j++) {
                const uint2 r = rsrc[j];
                lds_u[kOffTile + (2 * j) * kBlock + tid] = r.x;
                lds_u[kOffTile + (2 * j + 1) * kBlock + tid] = r.y;
              }
              lds_u[kOffTile + 10 * kBlock + tid] = sc.x;
              lds_u[kOffTile + 11 * kBlock + tid] = sc.y;
            }
            if (W.items_per_pixel == 1) {
              rng_store(rng_seed(hash_pixel(it.px, it.py, 0)));  // Sampler::start_pixel
            } else {
              const ulonglong2 c = ckpt[it.ckpt_index];
              rng_store(DRng{c.x, c.y});
            }
          }  // else: pixel outside the film (partial border tile): ask again next trip
        }
        pool_next += take;
      }
      if (!alive && !lane_done && have_item) {
        REGION(0);  // camera ray generation
        // start sample s_cur: BlueSobolSampler index = s, dimension = 0 (sampler.h:174-181)
        st.start_sample(st.s_cur());
        if constexpr (F & F_SOBOL)
          if (S.tables.kind == 2) st.set_dim(2);  // HaltonSampler::start_pixel / start_next_sample: dimension = 2
        if constexpr (kBigDim) big_dim = S.tables.kind == 2 ? 2 : 0;
        const int px = int(pxy & 0xffffu), py = int(pxy >> 16);
        // g++ evaluates gen_ray's arguments right to left (path.cpp:35): lens first, then jitter
        DRng g = rng_load();
        const float lx = rng_nextf(g);
        const float ly = rng_nextf(g);
        const float jx = rng_nextf(g);
        const float jy = rng_nextf(g);
        rng_store(g);
        const f2 pf{(float(px) + jx) / float(S.cam.W), (float(py) + jy) / float(S.cam.H)};
        const DRay r = camera_gen_ray(S.cam, pf, f2{lx, ly});
        ray_o = r.o;
        ray_d = r.d;
        ray_tmax = r.tmax;
        if constexpr (F & F_SSS) beta_flags = 0;
        alive = true;
      }
    }
    SEC_MARK(1);  // regeneration
    if (__all(lane_done && !alive)) break;
    if (!alive) continue;

    // ---------------- one radiance() invocation (path.cpp:42-124) ----------------
    DSampler sampler;
    sampler.px = int(pxy & 0xffffu);
    sampler.py = int(pxy >> 16);
    sampler.index = st.s_cur();
    sampler.dimension = st.dim();
    if constexpr (kBigDim)
      if (S.tables.kind != 0) sampler.dimension = big_dim;
    const int pv_length = st.length();
    int geom = -1, prim = 0;
    bool hit;
    {
      DRay ray{ray_o, ray_d, 0.0f, ray_tmax};
      hit = scene_traverse<false, F>(V, ray, stack, geom, prim);
      if (hit) geom &= kPrimIndexMask;  // (the packed word's flag bits are used by the queue kernel only)
      ray_tmax = ray.tmax;
    }
    SEC_MARK(2);  // closest-hit traversal

    // terminal result of this vertex, if it terminates
    bool terminal = false;
    f3 Lo = mk3(0.0f);
    bool has_light_pdf = false;
    float light_pdf = 0.0f;

    DSurface it;
    it.p = it.n = mk3(0.0f);
    it.uv = f2{0, 0};
    const DShape* shape = nullptr;
    const DMaterial* mat = nullptr;
    if (!hit) {
      terminal = true;  // path.cpp:75-81
      if constexpr (F & F_LIGHTS)
        if (S.env_light >= 0) {
          Lo = mk3(1.0f) * sky_color_of(ld3(V.lights[S.env_light].color), ray_d);
          if (!st.is_delta()) {
            has_light_pdf = true;
            light_pdf = 1 / (4 * kPi);  // Sky::pdf -- not divided by the light count
          }
        }
    } else {
      REGION(3);  // surface info
      shape = &V.shapes[geom];
      mat = &V.materials[shape->material];
      const f3 ph = ray_o + ray_tmax * ray_d;
      bool on_mesh = false;
      if constexpr (F & F_MESH) on_mesh = shape->kind == SHAPE_MESH;
      if (on_mesh) {
        if constexpr (F & F_EMBREE) mesh_surface_info_embree(V.rcpps, V.tri_verts, V.tri_attrs, as_int(shape->f[4]), prim, ray_o, ray_d, it);
        else mesh_surface_info(V.tri_verts, V.tri_attrs, as_int(shape->f[4]), prim, ph, it);
      } else shape_surface_info<F>(shape, ph, it);
      if (mat->kind == MAT_EMISSIVE) {  // path.cpp:83-87
        Lo = mk3(1.0f) * material_le(mat, it.n, -ray_d);
        if (!st.is_delta()) {
          has_light_pdf = true;
          const DRay ray{ray_o, ray_d, 0.0f, ray_tmax};
          light_pdf = shape_pdf<F>(shape, ray, it.n);  // lightsampler.cpp:27-29: / lights.size()
          if (S.num_lights != 1) light_pdf = light_pdf / float(size_t(S.num_lights));  // x / 1.0f == x exactly
        }
        terminal = true;
      } else if (pv_length + 1 >= S.max_path_length) {  // path.cpp:89
        terminal = true;
      }
    }

    SEC_MARK(3);  // surface info + emissive/terminal test
    if (!terminal) {
      REGION(6);  // non-terminal shading
      const f3 wi = -ray_d;
      m3 l2w = coordinate_system(it.n);  // interaction.h:14-17
      m3 w2l = transpose(l2w);
      // ---- material.sample_bxdf (material.h:30-131, material.cpp:9-28) ----
      const bool diffused = st.diffuse_length() > 0;
      const float min_roughness = diffused ? 0.6f : 0.0f;  // bxdf.h:15
      DBxdf bx;
      bx.kind = BX_DIFFUSE;
      bx.roughness = 0.0f;
      bx.ior = 1.0f;
      bool is_uber = false, is_sss = false, is_lobe = false;
      if constexpr (F & F_UBER) is_uber = mat->kind == MAT_UBER;
      if constexpr (F & F_UBER) is_lobe = mat->kind >= MAT_METAL;  // Metal / Glossy / Glass: one fixed lobe
      if constexpr (F & F_SSS) is_sss = mat->kind == MAT_SUBSURFACE;
      const MatParams mp = material_params<F>(mat, V.node_ops, it.p, it.n, it.uv);
      if (is_uber) {
        DRng g = rng_load();
        if (with_probability(mp.metallic, g)) {
          bx.kind = BX_CONDUCTOR;
          bx.roughness = mp.roughness;
        } else if (with_probability(mp.transmission, g)) {
          bx.kind = BX_REFR_DIEL;
          bx.roughness = mp.roughness;
          bx.ior = mp.ior;
        } else {
          bx.kind = BX_DIFF_DIEL;
          bx.roughness = mp.roughness;
          bx.ior = mp.ior;
        }
        rng_store(g);
      } else if (is_lobe) {  // material.h:39-78
        bx.kind = mat->kind == MAT_METAL ? BX_CONDUCTOR : mat->kind == MAT_GLOSSY ? BX_DIFF_DIEL : BX_REFR_DIEL;
        bx.roughness = pmax(mp.roughness, min_roughness);
        bx.ior = mp.ior;
      } else if (is_sss) {
        const float fr = FrDielectric(dot(wi, it.n), mat->ior);
        if (sampler_get1d<kSM>(T, sampler) < fr) {
          bx.kind = BX_REFRACTIVE;
          bx.roughness = pmax(mp.roughness, min_roughness);
          bx.ior = mat->ior;
        } else if (diffused) {
          bx.kind = BX_DIFFUSE;
        } else {
          bx.kind = BX_BSSRDF;
          bx.ior = mat->ior;
        }
      }
      bx.wi = mul(w2l, wi);  // material.h:119

      // ---- BSSRDF random walk inside the same shape (bxdf.cpp:329-353, :375-382) ----
      int beta_channel = 0;
      bool do_walk = false;
      if constexpr (F & F_SSS) do_walk = bx.kind == BX_BSSRDF;
      if (do_walk) {
        f3 p = it.p;
        f3 w = -wi;
        if (Refract(wi, it.n, bx.ior, w, nullptr)) {
          DRng g = rng_load();
          const int channel = int(rng_nextf(g) * 3);
          rng_store(g);
          const float sigma_t_inv = 1 / mat->sigma_s[channel];
          const f3 n0 = it.n;
          for (int i = 0;; i++) {
            DRay wr = i == 0 ? spawn_ray_raw(p, n0, w) : DRay{p, w, 0.0f, kFloatMax};
            DSurface sit;
            sit.p = sit.n = mk3(0.0f);  // non-mesh shapes leave them zero (Appendix A5)
            bool h;
            bool walk_mesh = false;
            if constexpr (F & F_MESH) walk_mesh = shape->kind == SHAPE_MESH;
            if (walk_mesh) {
              const DRayOct oct = make_oct(wr);
              int wprim = 0;
              h = mesh_traverse<false>(V, V.bvhs[as_int(shape->f[2])], wr, oct, stack, 0, wprim);
              if (h) mesh_surface_info(V.tri_verts, V.tri_attrs, as_int(shape->f[4]), wprim, ray_at(wr, wr.tmax), sit);
            } else {
              h = shape_intersect<F>(shape, wr);
            }
            if (!h) break;  // sample_p returns nullopt: nothing changes
            const float t = -plog(1 - sampler_get1d<kSM>(T, sampler)) * sigma_t_inv;
            if (wr.tmax < t) {
              beta_channel = channel + 1;
              it.p = sit.p;
              it.n = sit.n;
              l2w = coordinate_system(it.n);
              w2l = transpose(l2w);
              bx.wi = mul(w2l, -w);
              break;
            }
            p = ray_at(wr, t);
            w = uniform_sphere(sampler_get2d<kSM>(T, sampler));
          }
        }
      }

      SEC_MARK(4);  // sample_bxdf (+ BSSRDF walk)
      // ---- next-event estimation (path.cpp:98-113) ----
      f3 nee = mk3(0.0f);
      if (!bxdf_is_delta<F>(bx)) {
        // g++ order for LightSampler::sample's arguments (lightsampler.h:27): get2d, then get1d
        const f2 u2 = sampler_get2d<kSM>(T, sampler);
        float u1 = sampler_get1d<kSM>(T, sampler);
        if (S.num_lights > 0) {  // UniformLightSampler::sample lightsampler.cpp:12-26
          if (S.num_lights != 1) u1 *= float(S.num_lights);  // x * 1.0f == x exactly
          const int index = int(u1);
          const DLight* L = &V.lights[index];
          int lkind = LIGHT_AREA;
          if constexpr (F & F_LIGHTS) lkind = L->kind;
          bool lvalid = false;
          f3 lw = mk3(0.0f), lle = mk3(0.0f);
          float ldist = 0.0f, lpdf = 0.0f;
          if (lkind == LIGHT_AREA) {  // AreaLight::sample light.cpp:55-69
            const DShape* lshape = &V.shapes[L->geom];
            DShapeSample gs;
            if (shape_sample<F>(lshape, V.tri_verts, it.p, u2, u1 - float(index), gs)) {
              lle = material_le(&V.materials[lshape->material], gs.n, -gs.w);
              lvalid = !is_zero(lle);
              lw = gs.w;
              ldist = gs.distance;
              lpdf = gs.pdf;
            }
          } else {
            if constexpr (F & F_LIGHTS) lvalid = light_sample_other(L, it.p, u2, lw, ldist, lpdf, lle);
          }
          const bool ldelta = lkind == LIGHT_POINT || lkind == LIGHT_SPOT || lkind == LIGHT_DIRECTIONAL;  // light.h:111-113
          if (lvalid) {
            const float ls_pdf = S.num_lights != 1 ? lpdf / float(S.num_lights) : lpdf;
            REGION(7);  // shadow ray cast
            shadow_count++;
            DRay sr = spawn_ray(it.p, it.n, lw, ldist);
            int g2, p2;
            SEC_MARK(5);  // light sampling
            const bool occluded = scene_traverse<true, F>(V, sr, stack, g2, p2);
            SEC_MARK(6);  // shadow traversal
            if (!occluded) {
              bx.albedo = mp.albedo;
              bx.albedo_over_pi = mp.albedo_over_pi;
              const float cosine = absdot(lw, it.n);
              const f3 wo = mul(w2l, lw);
              const f3 f = bxdf_f<F>(bx, wo);
              if (ldelta) {  // path.cpp:104-106: no MIS against a delta light
                nee = mk3(0.0f) + lle * mk3(1.0f) * cosine * f / ls_pdf;
              } else {
                const float mis = balance_heuristic(ls_pdf, bxdf_pdf<F>(bx, wo));
                nee = mk3(0.0f) + lle * mk3(1.0f) * cosine * f / ls_pdf * mis;
              }
            }
          }
        }
      }

      SEC_MARK(7);  // NEE evaluation (and light sampling of lanes without a shadow ray)
      // ---- BSDF sampling + continuation (path.cpp:114-120) ----
      bx.albedo = mp.albedo;
      bx.albedo_over_pi = mp.albedo_over_pi;
      DBsdfSample bs;
      if (bxdf_sample<F, kSM>(bx, T, sampler, bs)) {
        const f3 wo_world = mul(l2w, bs.wo);
        const float cosine = absdot(wo_world, it.n);
        const int level = pv_length;
        const float entry[8] = {nee.x, nee.y, nee.z, bs.f.x, bs.f.y, bs.f.z, cosine / bs.pdf, bs.pdf};
        fold_store(level, entry);
        if constexpr (F & F_SSS)
          beta_flags = (beta_flags & ~(3ull << (2 * level))) | ((unsigned long long)beta_channel << (2 * level));
        const DRay nr = spawn_ray(it.p, it.n, wo_world, kFloatMax);
        ray_o = nr.o;
        ray_d = nr.d;
        ray_tmax = nr.tmax;
        st.set_dim(sampler.dimension & 0x1ff);
        if constexpr (kBigDim) big_dim = sampler.dimension;
        st.next_vertex(bs.is_delta);
      } else {
        // no continuation: this vertex resolves now with lo = nee (path.cpp:121)
        f3 beta = mk3(1.0f);
        if (beta_channel) {
          beta = mk3(0.0f);
          set(beta, beta_channel - 1, 3.0f);
        }
        Lo = mk3(0.0f) + vmin(mk3(1.0f) * beta * nee, mk3(8.0f));
        terminal = true;
      }
    }

    SEC_MARK(8);  // BSDF sample + push
    if (terminal) {
      // ---- backward fold through the pending levels (path.cpp:114-121, Appendix A1) ----
      f3 Li = Lo;
      bool lp_valid = has_light_pdf;
      float lp = light_pdf;
      REGION(8);  // terminal fold entry
      for (int level = pv_length - 1; level >= 0; level--) {
        REGION(9);  // fold level
        float e[8];
        fold_load(level, e);
        const f3 e_nee{e[0], e[1], e[2]};
        const f3 e_f{e[3], e[4], e[5]};
        const float e_cp = e[6], e_pdf = e[7];
        const float mis = lp_valid ? balance_heuristic(e_pdf, lp) : 1.0f;
        const f3 lo = e_nee + Li * e_f * (e_cp * mis);
        f3 beta = mk3(1.0f);
        if constexpr (F & F_SSS) {
          const unsigned bc = unsigned(beta_flags >> (2 * level)) & 3u;
          if (bc) {
            beta = mk3(0.0f);
            set(beta, int(bc) - 1, 3.0f);
          }
        }
        Li = mk3(0.0f) + vmin(mk3(1.0f) * beta * lo, mk3(8.0f));
        lp_valid = false;
      }
      // .w = radiance() invocations of this sample (= depth reached + 1); resolve_kernel sums them
      const int s_now = st.s_cur();
      samples[size_t(sample_base) + size_t(s_now) * 64u] = make_float4(Li.x, Li.y, Li.z, float(pv_length + 1));
      st.v = unsigned(s_now + 1);
      // item = kspi consecutive samples: a power of two that divides spp, or (another SobolSampler / HaltonSampler count) the whole pixel
      if ((kspi & (kspi - 1)) == 0 ? ((s_now + 1) & (kspi - 1)) == 0 : s_now + 1 == kspi) have_item = false;
      alive = false;
    }
    SEC_MARK(9);  // fold + sample store
  }

  SEC_FLUSH();
  // per-wave reduction of the shadow-ray counter, one atomic per wave
  unsigned long long sc = shadow_count;
  for (int off = 32; off > 0; off >>= 1) sc += __shfl_down(sc, off);
  if ((tid & 63) == 0) atomicAdd(&counters->shadow_rays, sc);
}

}  // namespace pine_gpu
#ifdef PINE_BAKED_SCENE
// generated at plan creation (pine_specialize.h) and found on the include path of that compile: namespace-less text
// that defines scene_traverse_baked<ANY, F>(ray, geom_out)
namespace pine_gpu {
#include "pine_baked_scene.inc"
}
#endif
#include "pine_trav.h"
#include "pine_queue_kernel.h"
