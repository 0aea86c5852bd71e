// pine_amd/csrc/pine_kernels.hip -- the PathIntegrator hot path on gfx950 (MI355X).
//
// Formulation (DESIGN.md has the long version):
//  * A persistent grid of 64-lane waves.  Each lane owns one *work item* = `samples_per_item`
//    consecutive camera samples of one pixel, and runs the reference's radiance() recursion as an
//    iterative state machine: one radiance() invocation ("path vertex") per loop trip.  A lane
//    whose path ends regenerates immediately (next sample of its item, or a new item pulled from a
//    global queue with one wave-aggregated atomic), so all 64 lanes stay busy until the queue runs
//    dry -- the exit condition every wave reaches.
//  * The per-level firefly clamp of the reference (path.cpp:121) forces a backward fold of
//    per-vertex terms; each non-terminal vertex spills a 32-byte FoldEntry to a lane-interleaved
//    global stack and the terminal vertex folds it back.
//  * Per-sample radiance goes to a [tile][sample][pixel-in-tile] buffer; a second kernel sums each
//    pixel's samples in sample order (the reference's `L += ...` order, path.cpp:34-37) so the
//    film is independent of scheduling and of the number of GPUs.
//  * The per-pixel RNG stream (pixel jitter) is sequential across a pixel's samples; a prepass
//    computes its state at every item boundary when the scene has no in-path RNG consumer,
//    otherwise one item = the whole pixel.
//
// No MFMA (no dense contraction here); the kernel is VALU/latency bound on cbox-class scenes.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include <dlfcn.h>

#include "../../include/pine_gpu.h"
#include "pine_device.h"
#include "pine_host.h"

struct pine_gpu_scene;
namespace pine_gpu {
SceneHost& scene_host(pine_gpu_scene* s);

constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kTile = 8;         // 8x8 pixel tiles = 64 pixels = one wave's worth of items
#ifndef PINE_LDS_FOLD_LEVELS
#define PINE_LDS_FOLD_LEVELS 1
#endif
constexpr int kLdsFoldLevels = PINE_LDS_FOLD_LEVELS;  // fold-stack levels kept in LDS (deeper levels spill to global memory)
constexpr int kPoolItems = 128;  // items a wave claims from the global queue per atomic
constexpr int kMaxDepth = 32;    // max_path_length supported (2 beta bits per level in one u64)

// Diagnostic section timing: per-wave s_memtime deltas summed per section.  Never compiled into the
// product build; the stamps only go to Counters::section_cycles, which nothing else reads.
#ifdef PINE_PROFILE_SECTIONS
__device__ unsigned long long g_region_lanes[16], g_region_hits[16];
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
  int lds_nodes;           // F_LDS_TOP variants: nodes[0 .. lds_nodes) are copied to LDS by every workgroup
};

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
  const DNode* lds_nodes;  // F_LDS_TOP: the workgroup's LDS copy of nodes[0 .. lds_node_count)
  int lds_node_count;
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

struct WorkParams {
  int tiles_x, tiles_y;
  int num_local_tiles;   // tiles owned by this shard
  int shard_rank, shard_world;
  int samples_per_item;  // k
  int items_per_pixel;   // spp / k (a power of two)
  int log2_items_per_pixel;
  unsigned tiles_x_magic;  // ceil(2^32 / tiles_x): see decode_item
  unsigned long long total_items;  // num_local_tiles * items_per_pixel * 64
  unsigned long long idle_budget_ticks;  // stage-queued kernel: a wave that finds no work for this long (100 MHz wall clock) bails out
  int debug_force_bail;  // test hook (PINE_GPU_FLAG_DEBUG_FORCE_BAIL): the first wave bails out at once
  int trav_min_lanes, trav_min_trips;  // resumable traversal (pine_trav.h): park when fewer lanes than this still travel after this many trips
  unsigned long long* progress;  // host-mapped word (or null): work items claimed so far, stored now and then (get_progress)
};
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
template <bool ANY, int STRIDE = kBlock, unsigned F = 0, class StackT = int>
__device__ __forceinline__ bool mesh_traverse(const SceneView& S, const DBvh bvh, DRay& ray,
                                              const DRayOct& oct, StackT* stack, int sp0, int& prim_out) {
  bool hit = false;
  auto leaf = [&](int start, int count) -> bool {
    for (int i = start; i < start + count; i++) {
      // leaf-ordered 48-byte record: v0 v1 v2 | triangle index (FlatAccel::tri_leaf)
      const float4* rec = S.tri_leaf + size_t(i) * 3;
      const float4 a = rec[0], b = rec[1], c = rec[2];
      const float v[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x};
      if (ANY) {
        if (tri_hit(v, ray)) return true;
      } else if (tri_intersect(v, ray)) {
        hit = true;
        prim_out = __float_as_int(c.y);
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

// ANY: BVH::hit (bvh.cpp:497-511).  !ANY: BVH::intersect (bvh.cpp:513-548) minus the final
// compute_surface_info, which the caller does once for the winning primitive.
// geom_out receives the winning primitive's PACKED word (index | emissive bit | kind).
template <bool ANY, unsigned F, int STRIDE = kBlock, class StackT = int>
__device__ __forceinline__ bool scene_traverse(const SceneView& S, DRay& ray, StackT* stack, int& geom_out,
                                               int& prim_out) {
  if (S.num_shapes == 0) return false;
  const DRayOct oct = make_oct(ray);
  const DBvh top = S.bvhs[0];
  bool hit = false;
  auto leaf = [&](int start, int count) -> bool {
    for (int i = start; i < start + count; i++) {
      REGION(ANY ? 5 : 2);  // leaf primitive test
      const DShape* sh = &S.leaf[i];
      const int word = sh->kind;  // (the packed word rides in the copy's kind field)
      const int kind = word >> kPrimKindShift;
      bool is_mesh = false;
      if constexpr (F & F_MESH) is_mesh = kind == SHAPE_MESH;
      if (is_mesh) {
        if constexpr (F & F_MESH) {
          const DBvh mb = S.bvhs[as_int(sh->f[2])];
          int prim = 0;
          const bool h = mesh_traverse<ANY, STRIDE, F>(S, mb, ray, oct, stack, S.stack_top, prim);
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
  bool valid;
};
__device__ __forceinline__ ItemInfo decode_item(const WorkParams& W, int film_w, int film_h, int spp,
                                                unsigned long long item) {
  ItemInfo it;
  const int p = int(item & 63);
  const unsigned long long tc = item >> 6;
  // spp and k are powers of two: shifts instead of 64-bit divisions (this runs once per camera sample)
  const int chunk = int(tc & (unsigned long long)(W.items_per_pixel - 1));
  const int ltile = int(tc >> W.log2_items_per_pixel);
  const int tile = ltile * W.shard_world + W.shard_rank;
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
__global__ void __launch_bounds__(kBlock) rng_checkpoint_kernel(WorkParams W, int film_w, int film_h, int spp,
                                                               ulonglong2* ckpt) {
  // one thread per (local tile, pixel in tile); walks the whole pixel, storing at chunk starts
  const unsigned long long t = blockIdx.x * (unsigned long long)kBlock + threadIdx.x;
  const unsigned long long n = (unsigned long long)W.num_local_tiles * 64ull;
  if (t >= n) return;
  const int p = int(t & 63);
  const int ltile = int(t >> 6);
  const int tile = ltile * W.shard_world + W.shard_rank;
  const int px = (tile % W.tiles_x) * kTile + (p & 7), py = (tile / W.tiles_x) * kTile + (p >> 3);
  DRng g = rng_seed(hash_pixel(px, py, 0));
  for (int c = 0; c < W.items_per_pixel; c++) {
    const unsigned long long item = ((unsigned long long)ltile * W.items_per_pixel + c) * 64ull + p;
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
//   (kWalk*).  Bit 31 stays clear (0xffffffff marks an empty context).
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
    v = (v & 0x1fffffu) + ((unsigned(length()) + 1u) << 21) + (((v >> 27) & 1u) | (delta ? 0u : 1u)) * (1u << 27) +
        (delta ? (1u << 28) : 0u);
  }
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
  V.lds_nodes = nullptr;
  V.lds_node_count = 0;
  V.stack_top = S.stack_top;
  V.num_shapes = S.num_shapes;
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
  } else {
    __syncthreads();  // Sobol rows staged above
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
              const ulonglong2 c = ckpt[item];
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
      if (on_mesh) tri_surface_info(V.tri_verts + size_t(prim) * 9, ph, it);
      else shape_surface_info<F>(shape, ph, it);
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
              if (h) tri_surface_info(V.tri_verts + size_t(wprim) * 9, ray_at(wr, wr.tmax), sit);
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
        st.set_dim(sampler.dimension);
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
      if (((s_now + 1) & (kspi - 1)) == 0) have_item = false;  // item = kspi consecutive samples, kspi a power of two
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
#include "pine_trav.h"
#include "pine_queue_kernel.h"
namespace pine_gpu {

// Compiled specialisations, most specific first.
using PathKernelFn = void (*)(DeviceScene, WorkParams, const ulonglong2*, float4*, float*, Counters*);
struct KernelVariant {
  unsigned features;
  int waves_per_simd;
  PathKernelFn fn;
  const char* name;
};
constexpr unsigned kFBoxes = F_AABB | F_OBB;
constexpr unsigned kFAnalytic = F_AABB | F_OBB | F_SPHERE | F_DISK | F_CONE | F_UBER;
static const KernelVariant kVariants[] = {
    {kFBoxes | F_LDS_SCENE, 4, path_trace_kernel<kFBoxes | F_LDS_SCENE, 4>, "rect+box/diffuse, scene in LDS"},
#ifndef PINE_ONLY_CBOX_VARIANT  /* experiment builds compile just the first variant */
    {kFAnalytic | F_LDS_SCENE, 2, path_trace_kernel<kFAnalytic | F_LDS_SCENE, 2>, "analytic shapes/uber, scene in LDS"},
    {kFAnalytic, 2, path_trace_kernel<kFAnalytic, 2>, "analytic shapes/uber"},
    {F_ALL | F_LDS_SCENE, 2, path_trace_kernel<F_ALL | F_LDS_SCENE, 2>, "all features, scene in LDS"},
    {F_ALL, 2, path_trace_kernel<F_ALL, 2>, "all features"},
#endif
};
constexpr int kNumVariants = int(sizeof(kVariants) / sizeof(kVariants[0]));

// stage-queued kernel (pine_queue_kernel.h): same feature lattice
using QueueKernelFn = void (*)(DeviceScene, WorkParams, const ulonglong2*, float4*, float*, uint32_t*, Counters*);
struct QueueVariant {
  unsigned features;
  int ctx;             // path contexts per workgroup
  size_t fixed_lds;    // LDS bytes before the traversal stack
  size_t min_stack;    // least size of the stack region (it also holds the sampler window in the F_LDS_TOP variants)
  QueueKernelFn fn;
  const char* name;
};
#define PINE_QV(F, CTX, NAME) \
  {F, CTX, QLayout<CTX, q_num_queues(F), ((F) & F_LDS_TOP) != 0>::fixed_bytes, QLayout<CTX, q_num_queues(F), ((F) & F_LDS_TOP) != 0>::min_stack_bytes, path_queue_kernel<F, CTX>, NAME}
static const QueueVariant kQueueVariants[] = {
    PINE_QV(F_OBB | F_LDS_SCENE, PINE_QCTX, "queue: rect+transformed box/diffuse, scene in LDS"),  // cbox exactly
    PINE_QV(kFBoxes | F_LDS_SCENE, PINE_QCTX, "queue: rect+box/diffuse, scene in LDS"),
#ifndef PINE_ONLY_CBOX_VARIANT
    PINE_QV(kFAnalytic | F_LDS_SCENE, PINE_QCTX, "queue: analytic shapes/uber, scene in LDS"),
    // scenes that do not fit LDS whole (F_LDS_TOP): 1024 contexts; the top of the BVH (breadth-first numbering) is cached
    // in whatever LDS the contexts and the 16-bit traversal stack (2 KB per slot) leave; traversals are resumable and
    // regrouped through the XS / XC queues (pine_trav.h)
    PINE_QV(F_SPHERE | F_DISK | F_CONE | F_UBER | F_LDS_TOP, 1024, "queue: rect+sphere+disk+cone/uber, 1024 contexts, BVH top in LDS (classic.pine's kinds exactly)"),
    PINE_QV(kFAnalytic | F_LDS_TOP, 1024, "queue: analytic shapes/uber, 1024 contexts, BVH top in LDS"),
    // everything except Subsurface (meshes, node-graph materials, every light kind); F_LDS_REST: few geometries (big
    // meshes or not): their shape / leaf / material / light records are staged in LDS too
    PINE_QV((F_ALL & ~F_SSS) | F_LDS_TOP | F_LDS_REST, 1024, "queue: all but SSS, 1024 contexts, BVH top + scene records in LDS"),
    PINE_QV((F_ALL & ~F_SSS) | F_LDS_TOP, 1024, "queue: all but SSS, 1024 contexts, BVH top in LDS"),
    // Subsurface: the BSSRDF random walk is a third stage (W) with its own queue
    PINE_QV(F_MESH | F_SSS | F_LDS_TOP | F_LDS_REST, 1024, "queue: rect+mesh/diffuse+subsurface, walk stage, 1024 contexts, BVH top + scene records in LDS"),
    PINE_QV(F_ALL | F_LDS_TOP | F_LDS_REST, 1024, "queue: all features, walk stage, 1024 contexts, BVH top + scene records in LDS"),
    PINE_QV(F_ALL | F_LDS_TOP, 1024, "queue: all features, walk stage, 1024 contexts, BVH top in LDS"),
    // BVHs of 65 536 nodes and more: 32-bit traversal stack, no node cache
    PINE_QV(F_ALL, 1024, "queue: all features, walk stage, 1024 contexts"),
#endif
};
constexpr int kNumQueueVariants = int(sizeof(kQueueVariants) / sizeof(kQueueVariants[0]));

// Ordered per-pixel sum: film[p] = (sum_{s=0..spp-1, in order} L_s) / spp  (path.cpp:34-38).
// One wave per tile, lane = pixel in tile: every sample row is one coalesced 1 KiB read.
// `packed` != 0: the output is this rank's tile-major slab [local tile][pixel in tile] (multi-GPU gather)
// instead of the row-major film.
__global__ void __launch_bounds__(kBlock) resolve_kernel(WorkParams W, int film_w, int film_h, int spp,
                                                        const float4* __restrict__ samples, float4* __restrict__ film,
                                                        Counters* __restrict__ counters, int packed) {
  const unsigned long long t = blockIdx.x * (unsigned long long)kBlock + threadIdx.x;
  const int ltile = int(t >> 6);
  if (ltile >= W.num_local_tiles) return;
  const int p = int(t & 63);
  const int tile = ltile * W.shard_world + W.shard_rank;
  const int px = (tile % W.tiles_x) * kTile + (p & 7), py = (tile / W.tiles_x) * kTile + (p >> 3);
  if (px >= film_w || py >= film_h) return;
  const float4* row = samples + (unsigned long long)ltile * (unsigned)spp * 64ull + p;
  f3 L = mk3(0.0f);
  unsigned long long verts = 0;
  // the sum is sequential in s (path.cpp:34-37), the loads need not be: 8 rows in flight per lane
  int s = 0;
  for (; s + 8 <= spp; s += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = row[(unsigned long long)(s + j) * 64ull];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      L = L + f3{v[j].x, v[j].y, v[j].z};
      verts += (unsigned long long)v[j].w;
    }
  }
  for (; s < spp; s++) {
    const float4 v = row[(unsigned long long)s * 64ull];
    L = L + f3{v.x, v.y, v.z};
    verts += (unsigned long long)v.w;
  }
  const f3 m = L / float(spp);
  const size_t out_index = packed ? size_t(ltile) * 64u + size_t(p) : size_t(py) * film_w + px;
  film[out_index] = make_float4(m.x, m.y, m.z, 1.0f);
  // radiance() invocation count of the launch (the unit of the roofline's algorithmic bytes)
  for (int off = 32; off > 0; off >>= 1) verts += __shfl_down(verts, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(&counters->vertices, verts);
}

// Multi-GPU: scatter the gathered per-rank slabs [rank][local tile][pixel in tile] into the row-major film.
__global__ void __launch_bounds__(kBlock) unpack_film_kernel(int film_w, int film_h, int tiles_x, int total_tiles, int world,
                                                            int tiles_per_rank, const float4* __restrict__ slabs,
                                                            float4* __restrict__ film) {
  const unsigned long long t = blockIdx.x * (unsigned long long)kBlock + threadIdx.x;
  const int tile = int(t >> 6);
  if (tile >= total_tiles) return;
  const int p = int(t & 63);
  const int px = (tile % tiles_x) * kTile + (p & 7), py = (tile / tiles_x) * kTile + (p >> 3);
  if (px >= film_w || py >= film_h) return;
  const int rank = tile % world, ltile = tile / world;
  film[size_t(py) * film_w + px] = slabs[(size_t(rank) * tiles_per_rank + ltile) * 64u + p];
}

// ------------------------------------------------------------------------------------------------
// Device-side unit-test kernels (parity of the building blocks against the oracle)
// ------------------------------------------------------------------------------------------------
__global__ void test_sincos_kernel(const float* x, long long n, float* s, float* c) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) {
    // the branch-free shared-reduction form the kernels call, cross-checked against the two single functions
    float sn, cs;
    psincos(x[i], sn, cs);
    const float s1 = psin(x[i]), c1 = pcos(x[i]);
    const bool same = __float_as_uint(s1) == __float_as_uint(sn) && __float_as_uint(c1) == __float_as_uint(cs);
    s[i] = same ? sn : __uint_as_float(0x7fc00001u);
    c[i] = same ? cs : __uint_as_float(0x7fc00001u);
  }
}
__global__ void test_powlog_kernel(const float* x, const float* y, long long n, float* p, float* l) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) {
    p[i] = ppow(x[i], y[i]);
    l[i] = plog(x[i]);
  }
}
__constant__ int kTestPixels[6][2] = {{0, 0}, {1, 0}, {3, 5}, {127, 127}, {128, 5}, {639, 639}};
__global__ void test_sampler_kernel(DTables T, int spp, float* out) {
  // one thread per (pixel, pass); layout identical to oracle_sampler_stream
  const int pix = blockIdx.x;
  if (threadIdx.x != 0) return;
  float* o = out + size_t(pix) * spp * (260 + 270);
  DSampler s;
  s.px = kTestPixels[pix][0];
  s.py = kTestPixels[pix][1];
  s.dimension = 0;
  s.index = 0;
  size_t k = 0;
  for (int i = 0; i < spp; i++) {
    for (int d = 0; d < 130; d++) {
      const f2 v = sampler_get2d(T, s);
      o[k++] = v.x;
      o[k++] = v.y;
    }
    s.dimension = 0;
    s.index++;
  }
  s.index = 0;
  for (int i = 0; i < spp; i++) {
    for (int d = 0; d < 90; d++) {
      o[k++] = sampler_get1d(T, s);
      const f2 v = sampler_get2d(T, s);
      o[k++] = v.x;
      o[k++] = v.y;
    }
    s.dimension = 0;
    s.index++;
  }
}
__global__ void test_rng_kernel(unsigned long long* out) {
  const int pix = threadIdx.x;
  if (pix >= 6) return;
  unsigned long long* o = out + pix * 19;
  const uint64_t h = hash_pixel(kTestPixels[pix][0], kTestPixels[pix][1], 0);
  o[0] = h;
  DRng g = rng_seed(h);
  o[1] = g.s0;
  o[2] = g.s1;
  for (int i = 0; i < 16; i++) o[3 + i] = (unsigned long long)(uint32_t)as_int(rng_nextf(g));
}
__global__ void test_shapes_kernel(const DShape* shapes, int num_shapes, const float* rays, long long nrays,
                                   float* out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= nrays * num_shapes) return;
  const int g = int(i / nrays);
  const long long r = i % nrays;
  const float* q = rays + r * 8;
  DRay ray{f3{q[0], q[1], q[2]}, f3{q[3], q[4], q[5]}, q[6], q[7]};
  float* o = out + i * 11;
  const DShape* S = &shapes[g];
  o[0] = shape_hit(S, ray) ? 1.0f : 0.0f;
  DRay r2 = ray;
  const bool h = shape_intersect(S, r2);
  o[1] = h ? 1.0f : 0.0f;
  o[2] = r2.tmax;
  DSurface it;
  it.p = it.n = mk3(0.0f);
  it.uv = f2{0, 0};
  if (h) shape_surface_info(S, ray_at(r2, r2.tmax), it);
  o[3] = it.p.x, o[4] = it.p.y, o[5] = it.p.z;
  o[6] = it.n.x, o[7] = it.n.y, o[8] = it.n.z;
  o[9] = it.uv.x, o[10] = it.uv.y;
}

// ================================================================================================
// Host side: plans, launches
// ================================================================================================
#define HIP_OK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                           \
      return -1;                                                                              \
    }                                                                                         \
  } while (0)

static std::string g_table_path;
// The packed tables are immutable once read; users take a shared snapshot, so a concurrent
// pine_gpu_set_table_path (which only drops the library's own reference) cannot free them under a reader.
using TableBlob = std::shared_ptr<const std::vector<uint8_t>>;
static TableBlob g_tables;
static std::mutex g_table_mutex;
static std::atomic<float> g_progress{0.0f};
static std::atomic<const volatile unsigned long long*> g_progress_src{nullptr};
static std::atomic<unsigned long long> g_progress_total{0};

static int load_tables(TableBlob& out) {
  std::lock_guard<std::mutex> lock(g_table_mutex);
  if (g_tables) {
    out = g_tables;
    return 0;
  }
  if (g_table_path.empty()) {
    // not set by the host: $PINE_GPU_TABLES, else data/bluesobol_u8.bin next to the directory this library sits in
    // (pine_amd/lib/libpine_gpu.so -> pine_amd/data/), wherever the process was started from
    if (const char* env = getenv("PINE_GPU_TABLES")) g_table_path = env;
    else {
      Dl_info info;
      if (dladdr(reinterpret_cast<const void*>(&load_tables), &info) && info.dli_fname) {
        std::string lib = info.dli_fname;
        const size_t slash = lib.rfind('/');
        g_table_path = (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/../data/bluesobol_u8.bin";
      }
    }
  }
  if (g_table_path.empty()) {
    set_error("BlueSobol table path not set (pine_gpu_set_table_path)");
    return -1;
  }
  FILE* f = fopen(g_table_path.c_str(), "rb");
  if (!f) {
    set_error("cannot open " + g_table_path);
    return -1;
  }
  std::vector<uint8_t> buf(65536 + 9 * 262144);
  size_t n = fread(buf.data(), 1, buf.size(), f);
  fclose(f);
  if (n != buf.size()) {
    set_error("short read of " + g_table_path);
    return -1;
  }
  g_tables = std::make_shared<const std::vector<uint8_t>>(std::move(buf));
  out = g_tables;
  return 0;
}
// The device keeps sobol_256spp_256d transposed ([dimension][sample] instead of [sample][dimension]):
// lanes of a wave usually ask for the same dimension at 64 different (ranked) sample rows, which is
// one 256-byte row here instead of 64 cache lines 256 bytes apart.
static std::vector<uint8_t> transposed_sobol(const std::vector<uint8_t>& tables) {
  std::vector<uint8_t> t(65536);
  for (int s = 0; s < 256; s++)
    for (int d = 0; d < 256; d++) t[d * 256 + s] = tables[s * 256 + d];
  return t;
}
static int effective_spp(int spp) {  // BlueSobolSampler ctor sampler.cpp:115-121
  if (spp > 256) spp = 256;
  if (spp <= 0) return 0;
  int x = spp - 1;
  for (unsigned i = 1; i < 32; i <<= 1) x |= x >> i;
  return x + 1;
}

template <class T>
static int upload(T*& dptr, const std::vector<T>& v) {
  dptr = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIP_OK(hipMalloc((void**)&dptr, bytes));
  if (!v.empty()) HIP_OK(hipMemcpy(dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// depth of the inner-node tree below `node` (= traversal stack entries that can be live)
static int bvh_depth(const std::vector<DNode>& nodes, int node) {
  const DNode& n = nodes[node];
  int d = 0;
  for (int c = 0; c < 2; c++)
    if (n.count[c] == 0) d = std::max(d, bvh_depth(nodes, n.child[c]));
  return d + 1;
}

// Everything the kernels index with comes from these host arrays: check every index against its array BEFORE anything
// is uploaded or launched (an out-of-range access on the device can take the whole GPU down, for everyone on it).
static int validate_device_scene(const FlatAccel& A, const std::vector<DShape>& shapes, size_t num_materials,
                                 const std::vector<DLight>& lights, int stack_top, int stack_total) {
  const int n_nodes = int(A.nodes.size()), n_prims = int(A.prims.size()), n_tris = int(A.tri_verts.size() / 9);
  auto bad = [](const std::string& what) {
    set_error("internal: inconsistent acceleration structure (" + what + ")");
    return -1;
  };
  if (A.bvhs.empty()) return bad("no top-level BVH");
  if (A.top_prim_begin < 0 || A.top_prim_begin > n_prims) return bad("top_prim_begin");
  if (A.tri_leaf.size() != size_t(A.top_prim_begin) * 12) return bad("tri_leaf size");
  for (size_t b = 0; b < A.bvhs.size(); b++) {
    const DBvh& v = A.bvhs[b];
    const int lo = b == 0 ? A.top_prim_begin : 0, hi = b == 0 ? n_prims : A.top_prim_begin;
    if (v.root_count > 0) {
      if (v.root_start < lo || v.root_start + v.root_count > hi) return bad("root leaf range");
    } else if (v.root >= n_nodes || (v.root < 0 && b != 0)) return bad("root node");
    // depth-first walk of this BVH: child indices, leaf ranges, depth against the stack the kernels get
    std::vector<std::pair<int, int>> todo;
    if (v.root_count == 0 && v.root >= 0) todo.push_back({v.root, 1});
    size_t visited = 0;
    while (!todo.empty()) {
      const auto [node, depth] = todo.back();
      todo.pop_back();
      if (++visited > size_t(n_nodes)) return bad("cycle in the node graph");
      if (depth > (b == 0 ? stack_top : stack_total - stack_top) + 1) return bad("tree deeper than the traversal stack");
      for (int c = 0; c < 2; c++) {
        const int ch = A.nodes[size_t(node)].child[c], cnt = A.nodes[size_t(node)].count[c];
        if (cnt > 0) {
          if (ch < lo || ch + cnt > hi) return bad("leaf range");
        } else if (cnt < 0 || ch < 0 || ch >= n_nodes) return bad("child index");
        else todo.push_back({ch, depth + 1});
      }
    }
  }
  for (int i = 0; i < A.top_prim_begin; i++) {
    int tri;
    memcpy(&tri, &A.tri_leaf[size_t(i) * 12 + 9], 4);
    if (tri < 0 || tri >= n_tris) return bad("triangle index of a leaf record");
  }
  for (int i = A.top_prim_begin; i < n_prims; i++)
    if (A.prims[size_t(i)] < 0 || A.prims[size_t(i)] >= int(shapes.size())) return bad("geometry index of a top-level primitive");
  for (const DShape& sh : shapes) {
    if (sh.material < 0 || size_t(sh.material) >= num_materials) return bad("material index");
    if (sh.kind == SHAPE_MESH) {
      int first, count, bvh;
      memcpy(&first, &sh.f[0], 4), memcpy(&count, &sh.f[1], 4), memcpy(&bvh, &sh.f[2], 4);
      if (count > 0 && (bvh < 1 || bvh >= int(A.bvhs.size()) || first < 0 || first + count > n_tris)) return bad("mesh record");
    }
  }
  for (const DLight& L : lights)
    if (L.kind == LIGHT_AREA && (L.geom < 0 || L.geom >= int(shapes.size()))) return bad("area light geometry");
  return 0;
}

}  // namespace pine_gpu

using namespace pine_gpu;

struct pine_gpu_plan {
  int device = 0;
  pine_gpu_render_params params{};
  DeviceScene S{};
  WorkParams W{};
  int film_w = 0, film_h = 0;
  // device buffers
  char* d_blob = nullptr;  // nodes | shapes | materials | bvhs | prims | lights
  float* d_tri = nullptr;
  float* d_tri_leaf = nullptr;
  uint8_t* d_tables = nullptr;
  int variant = -1;
  int queue_variant = -1;   // >= 0: the stage-queued kernel is used instead of path_trace_kernel
  uint32_t* d_ctxg = nullptr;
  ulonglong2* d_ckpt = nullptr;
  float4* d_samples = nullptr;
  float* d_fold = nullptr;
  Counters* d_counters = nullptr;
  int grid = 0;
  size_t lds_bytes = 0;
  bool serial_rng = false;
  // per-launch HIP events (prepass start / path kernel start / resolve start / end) for the last
  // kEvRing launches: reading them (stats_get) averages over the launches since the previous read,
  // so a timed loop never has to synchronise inside
  static constexpr int kEvRing = 64;
  hipEvent_t ev[kEvRing][4] = {};
  unsigned long long launch_count = 0, stats_read_upto = 0;
  bool timed = false;
  bool launched = false;
  hipStream_t last_stream = nullptr;
  unsigned long long* h_progress = nullptr;  // host-mapped progress word (PINE_GPU_FLAG_PROGRESS)
  float accel_build_ms = 0.0f, upload_ms = 0.0f;  // host-side cost of plan creation (reported by stats_get)
};

static int plan_check_counters(const Counters& c) {
  if (c.bail_count == 0) return 0;
  static const char* const kWhat[] = {"?", "idle budget exhausted with work outstanding", "ring slot never filled", "item-pool lock never released",
                                      "work-item hand-out did not converge", "?", "?", "forced by PINE_GPU_FLAG_DEBUG_FORCE_BAIL"};
  char msg[256];
  snprintf(msg, sizeof msg, "path kernel bailed out (%llu wave(s)): code %llu (%s), operands 0x%llx 0x%llx -- the film of this launch is incomplete",
           c.bail_count, c.bail_code, c.bail_code < 8 ? kWhat[c.bail_code] : "?", c.bail_a, c.bail_b);
  set_error(msg);
  return -1;
}

extern "C" {

float pine_gpu_progress(void) {
  // while a one-shot render is in flight: items claimed by the device / items of the launch
  const volatile unsigned long long* src = g_progress_src.load();
  if (src) {
    const unsigned long long total = g_progress_total.load();
    const float f = total ? float(double(*src) / double(total)) : 0.0f;
    return f < 1.0f ? f : 1.0f;
  }
  return g_progress.load();
}

/* Synchronise with the plan's last launch and report a protocol failure of its path kernel (a bounded
 * spin or the idle budget ran out: the film of that launch is incomplete).  0 = the launch completed. */
int pine_gpu_plan_check(pine_gpu_plan* p) {
  if (!p) {
    set_error("null argument");
    return -1;
  }
  if (!p->launched) return 0;
  HIP_OK(hipSetDevice(p->device));
  HIP_OK(hipStreamSynchronize(p->last_stream));
  Counters c;
  HIP_OK(hipMemcpy(&c, p->d_counters, sizeof c, hipMemcpyDeviceToHost));
  return plan_check_counters(c);
}

int pine_gpu_set_table_path(const char* path) {
  if (!path) {
    set_error("null path");
    return -1;
  }
  std::lock_guard<std::mutex> lock(g_table_mutex);
  g_table_path = path;
  g_tables.reset();  // (plans being built keep their own snapshot)
  return 0;
}

void pine_gpu_plan_destroy(pine_gpu_plan* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipFree(p->d_blob);
  (void)hipFree(p->d_tri);
  (void)hipFree(p->d_tri_leaf);
  (void)hipFree(p->d_tables);
  (void)hipFree(p->d_ctxg);
  (void)hipFree(p->d_ckpt);
  (void)hipFree(p->d_samples);
  (void)hipFree(p->d_fold);
  (void)hipFree(p->d_counters);
  if (p->h_progress) (void)hipHostFree(p->h_progress);
  for (auto& slot : p->ev)
    for (auto& e : slot)
      if (e) (void)hipEventDestroy(e);
  delete p;
}

static int plan_build(pine_gpu_plan* p, pine_gpu_scene* scene, const pine_gpu_render_params* prm) {
  SceneHost& H = scene_host(scene);
  if (!H.has_camera) {
    set_error("scene has no camera");
    return -1;
  }
  if (prm->max_path_length <= 0) {  // path.cpp:12-13
    set_error("`PathIntegrator` expect `max_path_length` to be positive");
    return -1;
  }
  if (prm->max_path_length > kMaxDepth) {
    set_error("max_path_length above the supported fold-stack depth (32)");
    return -1;
  }
  // BlueSampler(n): n rounded up to a power of two, clamped to 256 (sampler.cpp:115-121).  SobolSampler(n):
  // n as given (sampler.h:127-131); the work decomposition here needs a power of two.
  const bool sobol = prm->sampler == PINE_GPU_SAMPLER_SOBOL;
  if (prm->sampler != PINE_GPU_SAMPLER_BLUE && !sobol) {
    set_error("unknown sampler kind");
    return -1;
  }
  const int spp = sobol ? prm->spp : effective_spp(prm->spp);
  if (spp <= 0) {
    set_error(sobol ? "`SobolSampler` should have positive samples per pixel" : "samples per pixel must be positive");
    return -1;
  }
  if (sobol && ((spp & (spp - 1)) != 0 || spp > kMaxDeviceSpp)) {
    set_error("SobolSampler on the device: samples per pixel must be a power of two, at most 4096");
    return -1;
  }
  if (prm->shard_world < 1 || prm->shard_rank < 0 || prm->shard_rank >= prm->shard_world) {
    set_error("bad shard rank/world");
    return -1;
  }
  TableBlob tables_blob;
  if (load_tables(tables_blob)) return -1;
  const std::vector<uint8_t>& g_tables = *tables_blob;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("no HIP device available: the PathIntegrator hot path requires an AMD GPU (no CPU fallback)");
    return -1;
  }
  HIP_OK(hipSetDevice(prm->device));
  p->device = prm->device;
  p->params = *prm;
  const auto t_build0 = std::chrono::steady_clock::now();
  if (!H.accel.built) H.build_accel();
  const auto t_build1 = std::chrono::steady_clock::now();
  p->accel_build_ms = std::chrono::duration<float, std::milli>(t_build1 - t_build0).count();
  const FlatAccel& A = H.accel;

  std::vector<DShape> shapes;
  for (auto& g : H.geometries) shapes.push_back(g.shape);
  // one blob for the small records (16-byte aligned sections), so a workgroup can stage it in LDS
  std::vector<char> blob;
  auto put = [&](const void* src, size_t bytes) {
    size_t off = (blob.size() + 15) & ~size_t(15);
    blob.resize(off + std::max<size_t>(bytes, 16));
    if (bytes) memcpy(blob.data() + off, src, bytes);
    return int(off);
  };
  DeviceScene& S = p->S;
  S.off_nodes = put(A.nodes.data(), A.nodes.size() * sizeof(DNode));
  S.off_shapes = put(shapes.data(), shapes.size() * sizeof(DShape));
  std::vector<DMaterial> dev_materials;  // literals folded, node programs attached
  std::vector<DNodeOp> node_ops;
  if (!H.compile_node_programs(dev_materials, node_ops)) return -1;
  S.off_materials = put(dev_materials.data(), dev_materials.size() * sizeof(DMaterial));
  S.off_node_ops = put(node_ops.data(), node_ops.size() * sizeof(DNodeOp));
  S.off_bvhs = put(A.bvhs.data(), A.bvhs.size() * sizeof(DBvh));
  std::vector<int> packed_prims = A.prims;  // top-level entries: geometry | emissive | kind (pine_types.h)
  for (size_t i = size_t(A.top_prim_begin); i < packed_prims.size(); i++) {
    const int g = packed_prims[i];
    if (g > kPrimIndexMask) {
      set_error("too many geometries for the packed primitive word");
      return -1;
    }
    const DShape& sh = shapes[size_t(g)];
    packed_prims[i] = g | (H.materials[size_t(sh.material)].kind == MAT_EMISSIVE ? kPrimEmissiveBit : 0) | (sh.kind << kPrimKindShift);
  }
  S.off_prims = 0;  // (the primitive index list stays on the host: the leaf-ordered record copies below replace it)
  std::vector<DShape> leaf_shapes;  // SceneView::leaf
  for (size_t i = size_t(A.top_prim_begin); i < packed_prims.size(); i++) {
    DShape c = shapes[size_t(packed_prims[i] & kPrimIndexMask)];
    c.kind = packed_prims[i];
    leaf_shapes.push_back(c);
  }
  S.off_leaf = put(leaf_shapes.data(), leaf_shapes.size() * sizeof(DShape));
  S.top_prim_begin = A.top_prim_begin;
  std::vector<DLight> light_list = H.lights;  // + the environment light last (lightsampler.cpp:6-10)
  if (H.has_env) light_list.push_back(H.env);
  S.off_lights = put(light_list.data(), light_list.size() * sizeof(DLight));
  blob.resize((blob.size() + 15) & ~size_t(15));
  S.blob_bytes = int(blob.size());
  HIP_OK(hipMalloc((void**)&p->d_blob, blob.size()));
  HIP_OK(hipMemcpy(p->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
  if (upload(p->d_tri, A.tri_verts)) return -1;
  if (upload(p->d_tri_leaf, A.tri_leaf)) return -1;
  // tables: sobol + the selected spp variant
  int k = 0;
  while ((1 << k) < spp) k++;
  if (sobol) k = 0;  // (SobolSampler reads no table; any variant keeps the BlueSampler window loads in bounds)
  // device layout: sobolT 64 KiB | scramble 128 KiB | rank 128 KiB | 64 bytes = rank[0..63] again, so
  // a pixel's 40 consecutive ranking bytes never need the reference's modulo wrap
  HIP_OK(hipMalloc((void**)&p->d_tables, 65536 + 262144 + 64));
  {
    const std::vector<uint8_t> st = transposed_sobol(g_tables);
    HIP_OK(hipMemcpy(p->d_tables, st.data(), 65536, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMemcpy(p->d_tables + 65536, g_tables.data() + 65536 + size_t(k) * 262144, 262144, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(p->d_tables + 65536 + 262144, g_tables.data() + 65536 + size_t(k) * 262144 + 131072, 64,
                   hipMemcpyHostToDevice));

  S.blob = reinterpret_cast<const uint4*>(p->d_blob);
  S.nodes = reinterpret_cast<const DNode*>(p->d_blob + S.off_nodes);
  S.shapes = reinterpret_cast<const DShape*>(p->d_blob + S.off_shapes);
  S.materials = reinterpret_cast<const DMaterial*>(p->d_blob + S.off_materials);
  S.bvhs = reinterpret_cast<const DBvh*>(p->d_blob + S.off_bvhs);
  S.prims = nullptr;
  S.leaf = reinterpret_cast<const DShape*>(p->d_blob + S.off_leaf) - S.top_prim_begin;
  S.lights = reinterpret_cast<const DLight*>(p->d_blob + S.off_lights);
  S.node_ops = reinterpret_cast<const DNodeOp*>(p->d_blob + S.off_node_ops);
  S.tri_verts = p->d_tri;
  S.tri_leaf = reinterpret_cast<const float4*>(p->d_tri_leaf);
  S.lds_nodes = 0;
  S.num_lights = int(light_list.size());
  S.env_light = H.has_env ? int(light_list.size()) - 1 : -1;
  S.num_shapes = int(shapes.size());
  S.cam = H.camera;
  S.tables.sobol = p->d_tables;
  S.tables.scramble = p->d_tables + 65536;
  S.tables.rank = p->d_tables + 65536 + 131072;
  S.tables.lds_sobol = nullptr;
  S.tables.lds_tile = nullptr;
  S.tables.lds_scr = nullptr;
  S.tables.tile_stride = 0;
  S.tables.win_lo = 0;
  S.tables.win_len = 0;
  S.tables.kind = sobol ? 1 : 0;
  {
    // SobolSampler(spp): log2_spp = psl::log2i(spp); init(image_size): nbase4_digits =
    // log2i(roundup2(max(w, h))) + (log2_spp + 1) / 2   (sampler.h:127-129, sampler.cpp:81-84)
    int l2 = 0;
    while ((2 << l2) <= spp) l2++;
    int res = 1;
    while (res < std::max(H.camera.W, H.camera.H)) res *= 2;
    int lr = 0;
    while ((2 << lr) <= res) lr++;
    S.tables.sobol_log2_spp = l2;
    S.tables.sobol_digits = lr + (l2 + 1) / 2;
  }
  S.spp = spp;
  S.max_path_length = prm->max_path_length;
  int d_top = 0, d_mesh = 0;
  for (size_t b = 0; b < A.bvhs.size(); b++) {
    if (A.bvhs[b].root_count > 0 || A.bvhs[b].root < 0) continue;
    int d = bvh_depth(A.nodes, A.bvhs[b].root);
    if (b == 0) d_top = d;
    else d_mesh = std::max(d_mesh, d);
  }
  S.stack_top = d_top;
  S.stack_total = std::max(1, d_top + d_mesh);
  if (validate_device_scene(A, shapes, dev_materials.size(), light_list, S.stack_top, S.stack_total)) return -1;
  p->lds_bytes = kLdsFixedBytes + size_t(S.stack_total) * kBlock * sizeof(int);
  if (p->lds_bytes > 64 * 1024) {
    set_error("BVH too deep for the LDS traversal stack");
    return -1;
  }
  // kernel specialisation: the smallest compiled feature set that covers the scene
  unsigned need = 0;
  for (auto& sh : shapes) {
    switch (sh.kind) {
      case SHAPE_AABB: need |= F_AABB; break;
      case SHAPE_OBB: need |= F_OBB; break;
      case SHAPE_SPHERE: need |= F_SPHERE; break;
      case SHAPE_DISK: need |= F_DISK; break;
      case SHAPE_CONE: need |= F_CONE; break;
      case SHAPE_MESH: need |= F_MESH; break;
      case SHAPE_PLANE: case SHAPE_LINE: case SHAPE_CYLINDER: case SHAPE_TRIANGLE: need |= F_XSHAPES; break;
      default: break;
    }
  }
  for (auto& m : dev_materials) {
    if (m.kind == MAT_UBER || m.kind >= MAT_METAL) need |= F_UBER;  // the microfacet lobes
    if (m.kind == MAT_SUBSURFACE) need |= F_SSS;
  }
  if (!node_ops.empty()) need |= F_NODES;
  if (sobol) need |= F_SOBOL;
  if (sobol && (need & F_SSS)) {
    // a BSSRDF walk draws from the sampler at every step: SobolSampler's dimension counter (no wrap-around,
    // unlike BlueSampler's) is unbounded there and does not fit the packed path state
    set_error("SobolSampler with Subsurface materials is not supported on the device");
    return -1;
  }
  for (auto& L : light_list)
    if (L.kind != LIGHT_AREA) need |= F_LIGHTS;
  const bool lds_ok = size_t(S.blob_bytes) <= 32 * 1024 && getenv("PINE_GPU_NO_LDS_SCENE") == nullptr;
  p->variant = -1;
  for (int v = 0; v < kNumVariants; v++) {
    const unsigned F = kVariants[v].features;
    if ((F & need) != need) continue;
    if (((F & F_LDS_SCENE) != 0) != lds_ok) continue;
    if (getenv("PINE_GPU_WPS") && atoi(getenv("PINE_GPU_WPS")) != kVariants[v].waves_per_simd) continue;
    p->variant = v;
    break;
  }
  if (p->variant < 0) {
    set_error("no kernel variant covers this scene");
    return -1;
  }
  if (kVariants[p->variant].features & F_LDS_SCENE) p->lds_bytes += size_t(S.blob_bytes);
  // The stage-queued kernel is the default whenever a variant covers the scene and its LDS fits;
  // PINE_GPU_KERNEL=mega forces the lane-owns-a-path kernel, which covers every scene.
  p->queue_variant = -1;
  {
    const char* ksel = getenv("PINE_GPU_KERNEL");
    const bool want_queue = !(ksel && std::string(ksel) == "mega");
    if (want_queue) {
      const char* no_top = getenv("PINE_GPU_NO_LDS_TOP");  // (measurement aid: keep every node in global memory)
      for (int v = 0; v < kNumQueueVariants; v++) {
        const unsigned F = kQueueVariants[v].features;
        if ((F & need) != need) continue;
        if ((F & F_LDS_SCENE) && !lds_ok) continue;  // (a scene-in-global variant later in the table is the fallback when LDS is short)
        if ((F & F_LDS_TOP) && (A.nodes.size() > 65535 || S.stack_total > kTravMaxStack)) continue;  // 16-bit stack entries; a parked record holds 32
        const size_t rest_bytes = size_t(S.blob_bytes - S.off_shapes);
        if ((F & F_LDS_REST) && rest_bytes > 12 * 1024) continue;
        const size_t stack_bytes = std::max(kQueueVariants[v].min_stack,
                                            size_t(S.stack_total) * kQBlock * ((F & F_LDS_TOP) ? sizeof(unsigned short) : sizeof(int)));
        const size_t lds = kQueueVariants[v].fixed_lds + stack_bytes + ((F & F_LDS_SCENE) ? size_t(S.blob_bytes) : 0) +
                           ((F & F_LDS_REST) ? rest_bytes : 0);
        if (lds > 160 * 1024) continue;
        p->queue_variant = v;
        p->lds_bytes = lds;
        S.lds_nodes = 0;
        if (F & F_LDS_TOP) {
          S.lds_nodes = int(std::min<size_t>(A.nodes.size(), (160 * 1024 - lds) / sizeof(DNode)));
          if (no_top) S.lds_nodes = 0;
          p->lds_bytes += size_t(S.lds_nodes) * sizeof(DNode);
        }
        break;
      }
    }
  }

  // scenes whose materials draw from the per-pixel RNG inside radiance() (Uber with fractional
  // metallic/transmission: sampler.h:317-324; BSSRDF channel pick: bxdf.cpp:335) make a pixel's
  // samples sequentially dependent: one item = the whole pixel.
  bool in_path_rng = false;
  for (auto& m : dev_materials) {
    if (m.kind == MAT_SUBSURFACE) in_path_rng = true;
    if (m.kind == MAT_UBER && (m.prog[2] >= 0 || m.prog[3] >= 0)) in_path_rng = true;  // value known only at the surface
    if (m.kind == MAT_UBER) {
      if (m.metallic != 0 && m.metallic != 1) in_path_rng = true;
      if (m.metallic != 1 && m.transmission != 0 && m.transmission != 1) in_path_rng = true;
    }
  }
  p->serial_rng = in_path_rng;
  int kspi = prm->samples_per_item;
  if (in_path_rng) kspi = spp;
  else if (kspi <= 0) kspi = p->queue_variant >= 0 ? 1 : std::min(spp, 4);  // (queue kernel: one sample per item balances small shards best)
  if (kspi > spp) kspi = spp;
  {  // spp is a power of two; k must be a power of two dividing it
    int k2 = 1;
    while (k2 * 2 <= kspi) k2 *= 2;
    kspi = k2;
  }

  WorkParams& W = p->W;
  p->film_w = H.camera.W;
  p->film_h = H.camera.H;
  W.tiles_x = (p->film_w + kTile - 1) / kTile;
  W.tiles_y = (p->film_h + kTile - 1) / kTile;
  const int total_tiles = W.tiles_x * W.tiles_y;
  W.shard_rank = prm->shard_rank;
  W.shard_world = prm->shard_world;
  W.num_local_tiles = (total_tiles - prm->shard_rank + prm->shard_world - 1) / prm->shard_world;
  W.samples_per_item = kspi;
  W.items_per_pixel = spp / kspi;
  W.log2_items_per_pixel = 0;
  while ((1 << W.log2_items_per_pixel) < W.items_per_pixel) W.log2_items_per_pixel++;
  W.tiles_x_magic = unsigned(((1ull << 32) + unsigned(W.tiles_x) - 1) / unsigned(W.tiles_x));  // tiles_x >= 1
  W.total_items = (unsigned long long)W.num_local_tiles * W.items_per_pixel * 64ull;
  // packing limits of the kernels: pixel coordinates travel as 16 + 16 bits, the sample-buffer index of a
  // context as 32 bits
  if (p->film_w > 65535 || p->film_h > 65535) {
    set_error("film sides above 65535 are not supported");
    return -1;
  }
  if ((unsigned long long)W.num_local_tiles * 64ull * (unsigned long long)spp >= (1ull << 32)) {
    set_error("film pixels x samples per pixel of one shard must stay below 2^32 (render in several shards)");
    return -1;
  }
  {
    double budget_s = 30.0;
    if (const char* e = getenv("PINE_GPU_IDLE_BUDGET_S")) budget_s = atof(e) > 0 ? atof(e) : budget_s;
    W.idle_budget_ticks = (unsigned long long)(budget_s * 100e6);  // wall_clock64(): 100 MHz
  }
  W.debug_force_bail = (prm->flags & PINE_GPU_FLAG_DEBUG_FORCE_BAIL) ? 1 : 0;
  // Parking is OFF by default: measured on the MI355X it removes a quarter of the VALU instructions of the
  // mesh scene but the kernels wait on memory, and a parked traversal costs extra round trips.
  W.trav_min_lanes = 0;
  W.trav_min_trips = 8;
  if (const char* e = getenv("PINE_GPU_TRAV_MIN_LANES")) W.trav_min_lanes = atoi(e);  // (0 = never park: measurement aid)
  if (const char* e = getenv("PINE_GPU_TRAV_MIN_TRIPS")) W.trav_min_trips = atoi(e) > 0 ? atoi(e) : 1;
  W.progress = nullptr;
  if (prm->flags & PINE_GPU_FLAG_PROGRESS) {
    HIP_OK(hipHostMalloc((void**)&p->h_progress, sizeof(unsigned long long), hipHostMallocMapped));
    *p->h_progress = 0;
    HIP_OK(hipHostGetDevicePointer((void**)&W.progress, p->h_progress, 0));
  }

  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, prm->device));
  int blocks_per_cu = 0;
  if (p->queue_variant >= 0) {
    blocks_per_cu = 1;  // one 1024-thread workgroup per CU owns the CU's LDS
    HIP_OK(hipFuncSetAttribute((const void*)kQueueVariants[p->queue_variant].fn,
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(p->lds_bytes)));
  } else {
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, (const void*)kVariants[p->variant].fn, kBlock, p->lds_bytes));
  }
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  if (blocks_per_cu > 8) blocks_per_cu = 8;
  const char* env_bpc = getenv("PINE_GPU_BLOCKS_PER_CU");
  if (env_bpc && atoi(env_bpc) > 0) blocks_per_cu = atoi(env_bpc);
  unsigned long long want = (W.total_items + kBlock - 1) / kBlock;
  const int qctx = p->queue_variant >= 0 ? kQueueVariants[p->queue_variant].ctx : 0;
  if (p->queue_variant >= 0) want = (W.total_items + qctx - 1) / qctx;
  p->grid = int(std::min<unsigned long long>(want, (unsigned long long)prop.multiProcessorCount * blocks_per_cu));
  if (p->grid < 1) p->grid = 1;

  if (W.items_per_pixel > 1) HIP_OK(hipMalloc((void**)&p->d_ckpt, W.total_items * sizeof(ulonglong2)));
  HIP_OK(hipMalloc((void**)&p->d_samples, (size_t)W.num_local_tiles * spp * 64 * sizeof(float4)));
  const size_t fold_slots = p->queue_variant >= 0 ? size_t(p->grid) * qctx : size_t(p->grid) * kBlock;
  HIP_OK(hipMalloc((void**)&p->d_fold, size_t(prm->max_path_length) * 8 * fold_slots * sizeof(float)));
  if (p->queue_variant >= 0)
    HIP_OK(hipMalloc((void**)&p->d_ctxg, size_t(p->grid) * qctx * q_ctx_global_dwords(kQueueVariants[p->queue_variant].features) * sizeof(uint32_t)));
  HIP_OK(hipMalloc((void**)&p->d_counters, sizeof(Counters)));
  p->timed = (prm->flags & PINE_GPU_FLAG_TIMING) != 0;
  if (p->timed)
    for (auto& slot : p->ev)
      for (auto& e : slot) HIP_OK(hipEventCreate(&e));
  HIP_OK(hipDeviceSynchronize());
  p->upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_build1).count();
  return 0;
}

pine_gpu_plan* pine_gpu_plan_create(pine_gpu_scene* scene, const pine_gpu_render_params* prm) {
  if (!scene || !prm) {
    set_error("null argument");
    return nullptr;
  }
  pine_gpu_plan* p = new pine_gpu_plan();
  if (plan_build(p, scene, prm)) {
    std::string keep = pine_gpu_last_error();
    pine_gpu_plan_destroy(p);
    set_error(keep);
    return nullptr;
  }
  return p;
}

static int plan_launch(pine_gpu_plan* p, void* film_dev, void* stream_, bool packed) {
  if (!p || !film_dev) {
    set_error("null argument");
    return -1;
  }
  hipStream_t stream = (hipStream_t)stream_;
  HIP_OK(hipSetDevice(p->device));
  (void)hipGetLastError();  // (HIP's last error is sticky: what the check at the end reports must come from THIS launch's calls)
  g_progress.store(0.0f);
  const size_t film_bytes = size_t(p->film_w) * p->film_h * sizeof(float4);
  if (p->W.shard_world > 1 && !packed) HIP_OK(hipMemsetAsync(film_dev, 0, film_bytes, stream));
  HIP_OK(hipMemsetAsync(p->d_counters, 0, sizeof(Counters), stream));
  hipEvent_t* ev = p->ev[p->launch_count % pine_gpu_plan::kEvRing];
  if (p->timed) HIP_OK(hipEventRecord(ev[0], stream));
  // (a shard can own no tile at all -- more ranks than 8x8 tiles: nothing to launch, the film / slab stays zero)
  const bool has_work = p->W.num_local_tiles > 0;
  if (has_work && p->W.items_per_pixel > 1) {
    const unsigned long long n = (unsigned long long)p->W.num_local_tiles * 64ull;
    hipLaunchKernelGGL(rng_checkpoint_kernel, dim3(unsigned((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       p->W, p->film_w, p->film_h, p->S.spp, p->d_ckpt);
  }
  if (p->timed) HIP_OK(hipEventRecord(ev[1], stream));
  if (!has_work) {
  } else if (p->queue_variant >= 0)
    hipLaunchKernelGGL(kQueueVariants[p->queue_variant].fn, dim3(p->grid), dim3(kQBlock), p->lds_bytes, stream, p->S,
                       p->W, (const ulonglong2*)p->d_ckpt, p->d_samples, p->d_fold, p->d_ctxg, p->d_counters);
  else
    hipLaunchKernelGGL(kVariants[p->variant].fn, dim3(p->grid), dim3(kBlock), p->lds_bytes, stream, p->S, p->W,
                       (const ulonglong2*)p->d_ckpt, p->d_samples, p->d_fold, p->d_counters);
  if (p->timed) HIP_OK(hipEventRecord(ev[2], stream));
  if (has_work) {
    const unsigned long long n = (unsigned long long)p->W.num_local_tiles * 64ull;
    hipLaunchKernelGGL(resolve_kernel, dim3(unsigned((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, p->W,
                       p->film_w, p->film_h, p->S.spp, p->d_samples, (float4*)film_dev, p->d_counters, packed ? 1 : 0);
  }
  if (p->timed) HIP_OK(hipEventRecord(ev[3], stream));
  HIP_OK(hipGetLastError());
  p->launched = true;
  p->launch_count++;
  p->last_stream = stream;
  return 0;
}

int pine_gpu_plan_launch(pine_gpu_plan* p, void* film_dev, void* stream) { return plan_launch(p, film_dev, stream, false); }
int pine_gpu_plan_launch_packed(pine_gpu_plan* p, void* slab_dev, void* stream) { return plan_launch(p, slab_dev, stream, true); }

int64_t pine_gpu_packed_slab_floats(int film_w, int film_h, int world) {
  if (film_w <= 0 || film_h <= 0 || world < 1) return -1;
  const int64_t tiles = int64_t((film_w + kTile - 1) / kTile) * ((film_h + kTile - 1) / kTile);
  return (tiles + world - 1) / world * 64 * 4;
}

int pine_gpu_packed_offset(int film_w, int film_h, int world, int x, int y, int* rank_out, int64_t* float4_index_out) {
  if (film_w <= 0 || film_h <= 0 || world < 1 || x < 0 || y < 0 || x >= film_w || y >= film_h) {
    set_error("bad argument");
    return -1;
  }
  const int tiles_x = (film_w + kTile - 1) / kTile;
  const int tile = (y / kTile) * tiles_x + x / kTile;
  if (rank_out) *rank_out = tile % world;
  if (float4_index_out) *float4_index_out = int64_t(tile / world) * 64 + (y % kTile) * kTile + x % kTile;
  return 0;
}

int pine_gpu_film_unpack(int film_w, int film_h, int world, int device, const void* slabs_dev, void* film_dev, void* stream_) {
  if (!slabs_dev || !film_dev || film_w <= 0 || film_h <= 0 || world < 1) {
    set_error("bad argument");
    return -1;
  }
  HIP_OK(hipSetDevice(device));
  const int tiles_x = (film_w + kTile - 1) / kTile, tiles_y = (film_h + kTile - 1) / kTile;
  const int total = tiles_x * tiles_y;
  const int per_rank = (total + world - 1) / world;
  const unsigned long long n = (unsigned long long)total * 64ull;
  hipLaunchKernelGGL(unpack_film_kernel, dim3(unsigned((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream_,
                     film_w, film_h, tiles_x, total, world, per_rank, (const float4*)slabs_dev, (float4*)film_dev);
  HIP_OK(hipGetLastError());
  return 0;
}

int pine_gpu_plan_stats_get(pine_gpu_plan* p, pine_gpu_plan_stats* out) {
  if (!p || !out) {
    set_error("null argument");
    return -1;
  }
  memset(out, 0, sizeof *out);
  HIP_OK(hipSetDevice(p->device));
  out->camera_samples = (unsigned long long)p->W.num_local_tiles * 64ull * p->S.spp;
  // tiles on the film border may be partially outside: count real pixels
  {
    unsigned long long px = 0;
    for (int lt = 0; lt < p->W.num_local_tiles; lt++) {
      int tile = lt * p->W.shard_world + p->W.shard_rank;
      int tx = tile % p->W.tiles_x, ty = tile / p->W.tiles_x;
      int w = std::min(kTile, p->film_w - tx * kTile), h = std::min(kTile, p->film_h - ty * kTile);
      px += (unsigned long long)w * h;
    }
    out->camera_samples = px * p->S.spp;
  }
  out->spp_effective = p->S.spp;
  out->samples_per_item = p->W.samples_per_item;
  out->grid_blocks = p->grid;
  out->block_threads = p->queue_variant >= 0 ? kQBlock : kBlock;
  out->lds_bytes = int(p->lds_bytes);
  out->accel_build_ms = p->accel_build_ms;
  out->upload_ms = p->upload_ms;
  if (p->launched) {
    HIP_OK(hipStreamSynchronize(p->last_stream));
    Counters c;
    HIP_OK(hipMemcpy(&c, p->d_counters, sizeof c, hipMemcpyDeviceToHost));
    if (plan_check_counters(c)) return -1;
    out->vertices = c.vertices;
    out->shadow_rays = c.shadow_rays;
    out->walk_steps = c.walk_steps;
    if (p->timed) {
      // mean over the launches since the previous read (at most the last kEvRing of them)
      unsigned long long first = p->stats_read_upto;
      if (p->launch_count - first > (unsigned long long)pine_gpu_plan::kEvRing) first = p->launch_count - pine_gpu_plan::kEvRing;
      if (first == p->launch_count) first = p->launch_count - 1;  // nothing new: report the last launch again
      double a = 0, b = 0, c3 = 0;
      for (unsigned long long i = first; i < p->launch_count; i++) {
        hipEvent_t* ev = p->ev[i % pine_gpu_plan::kEvRing];
        float x = 0, y = 0, z = 0;
        HIP_OK(hipEventElapsedTime(&x, ev[0], ev[1]));
        HIP_OK(hipEventElapsedTime(&y, ev[1], ev[2]));
        HIP_OK(hipEventElapsedTime(&z, ev[2], ev[3]));
        a += x, b += y, c3 += z;
      }
      const double n = double(p->launch_count - first);
      out->prepass_ms = float(a / n);
      out->trace_ms = float(b / n);
      out->resolve_ms = float(c3 / n);
      out->timed_launches = int32_t(p->launch_count - first);
      p->stats_read_upto = p->launch_count;
    }
  }
  return 0;
}

int pine_gpu_plan_debug_sections(pine_gpu_plan* p, uint64_t out[16]) {
  if (!p || !out) {
    set_error("null argument");
    return -1;
  }
  HIP_OK(hipSetDevice(p->device));
  HIP_OK(hipDeviceSynchronize());
  Counters c;
  HIP_OK(hipMemcpy(&c, p->d_counters, sizeof c, hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; i++) out[i] = c.section_cycles[i];
#ifdef PINE_PROFILE_SECTIONS
  unsigned long long rl[16], rh[16];
  HIP_OK(hipMemcpyFromSymbol(rl, HIP_SYMBOL(g_region_lanes), sizeof rl));
  HIP_OK(hipMemcpyFromSymbol(rh, HIP_SYMBOL(g_region_hits), sizeof rh));
  for (int i = 0; i < 16; i++)
    if (rh[i]) fprintf(stderr, "region %2d: entries %llu avg active lanes %.2f\n", i, rh[i], double(rl[i]) / double(rh[i]));
#endif
  return 0;
}

int pine_gpu_plan_read_samples(pine_gpu_plan* p, float* out, int64_t capacity) {
  if (!p || !out) {
    set_error("null argument");
    return -1;
  }
  const int spp = p->S.spp;
  const int64_t need = int64_t(p->film_w) * p->film_h * spp * 4;
  if (capacity < need) {
    set_error("capacity too small");
    return -1;
  }
  HIP_OK(hipSetDevice(p->device));
  HIP_OK(hipDeviceSynchronize());
  std::vector<float> tmp(size_t(p->W.num_local_tiles) * spp * 64 * 4);
  HIP_OK(hipMemcpy(tmp.data(), p->d_samples, tmp.size() * 4, hipMemcpyDeviceToHost));
  memset(out, 0, size_t(need) * 4);
  for (int lt = 0; lt < p->W.num_local_tiles; lt++) {
    int tile = lt * p->W.shard_world + p->W.shard_rank;
    int tx = tile % p->W.tiles_x, ty = tile / p->W.tiles_x;
    for (int q = 0; q < 64; q++) {
      int px = tx * kTile + (q & 7), py = ty * kTile + (q >> 3);
      if (px >= p->film_w || py >= p->film_h) continue;
      for (int s = 0; s < spp; s++)
        memcpy(out + ((size_t(py) * p->film_w + px) * spp + s) * 4,
               tmp.data() + ((size_t(lt) * spp + s) * 64 + q) * 4, 16);
    }
  }
  return 0;
}

int pine_gpu_path_render(pine_gpu_scene* scene, const pine_gpu_render_params* prm, float* film_out) {
  if (!scene || !prm || !film_out) {
    set_error("null argument");
    return -1;
  }
  pine_gpu_render_params prm2 = *prm;
  prm2.flags |= PINE_GPU_FLAG_PROGRESS;  // the reference's CLI polls get_progress() while render() runs (src/cli/pine.cpp:36-40)
  pine_gpu_plan* p = pine_gpu_plan_create(scene, &prm2);
  if (!p) return -1;
  int rc = -1;
  void* d_film = nullptr;
  const size_t bytes = size_t(p->film_w) * p->film_h * 16;
  do {
    if (hipMalloc(&d_film, bytes) != hipSuccess) {
      set_error("hipMalloc(film) failed");
      break;
    }
    g_progress_total.store(p->W.total_items);
    g_progress_src.store(p->h_progress);
    if (pine_gpu_plan_launch(p, d_film, nullptr)) break;
    if (hipMemcpy(film_out, d_film, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
      set_error("film download failed");
      break;
    }
    if (pine_gpu_plan_check(p)) break;  // a bailed-out path kernel leaves an incomplete film: fail, do not return it as a result
    rc = 0;
  } while (0);
  g_progress_src.store(nullptr);
  g_progress.store(rc ? 0.0f : 1.0f);
  std::string keep = rc ? pine_gpu_last_error() : "";
  hipFree(d_film);
  pine_gpu_plan_destroy(p);
  if (rc) set_error(keep);
  return rc;
}

/* One process, several devices: shard r of n (8x8-pixel tiles dealt round-robin) renders on devices[r];
 * every device writes its tiles into a packed slab, the slabs are copied device-to-device (peer copies over xGMI) into
 * one [rank][slab] buffer on devices[0], scattered into the row-major film there and downloaded.  Bit-identical to the
 * one-device film for any list (the same device may appear more than once).  This is what the C++ facade and the PRL
 * command line use to drive a whole node without torch.distributed. */
int pine_gpu_path_render_devices(pine_gpu_scene* scene, const pine_gpu_render_params* prm, const int* devices, int num_devices,
                                 float* film_out) {
  if (!scene || !prm || !devices || !film_out || num_devices < 1 || num_devices > 64) {
    set_error("bad argument");
    return -1;
  }
  if (num_devices == 1) {
    pine_gpu_render_params one = *prm;
    one.device = devices[0];
    one.shard_rank = 0;
    one.shard_world = 1;
    return pine_gpu_path_render(scene, &one, film_out);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("no HIP device available: the PathIntegrator hot path requires an AMD GPU (no CPU fallback)");
    return -1;
  }
  for (int r = 0; r < num_devices; r++)
    if (devices[r] < 0 || devices[r] >= ndev) {
      set_error("device ordinal out of range");
      return -1;
    }
  std::vector<pine_gpu_plan*> plans(size_t(num_devices), nullptr);
  std::vector<void*> slabs(size_t(num_devices), nullptr);
  std::vector<hipStream_t> streams(size_t(num_devices), nullptr);
  void *gathered = nullptr, *d_film = nullptr;
  int rc = -1;
  std::string err;
  do {
    bool ok = true;
    for (int r = 0; r < num_devices && ok; r++) {
      pine_gpu_render_params p = *prm;
      p.device = devices[r];
      p.shard_rank = r;
      p.shard_world = num_devices;
      plans[size_t(r)] = pine_gpu_plan_create(scene, &p);
      ok = plans[size_t(r)] != nullptr;
    }
    if (!ok) break;
    const int w = plans[0]->film_w, h = plans[0]->film_h;
    const int64_t slab_floats = pine_gpu_packed_slab_floats(w, h, num_devices);
    const size_t slab_bytes = size_t(slab_floats) * 4, film_bytes = size_t(w) * h * 16;
    if (hipSetDevice(devices[0]) != hipSuccess || hipMalloc(&gathered, slab_bytes * size_t(num_devices)) != hipSuccess ||
        hipMalloc(&d_film, film_bytes) != hipSuccess) {
      set_error("device allocation failed");
      break;
    }
    for (int r = 0; r < num_devices && ok; r++) {
      ok = hipSetDevice(devices[r]) == hipSuccess && hipStreamCreateWithFlags(&streams[size_t(r)], hipStreamNonBlocking) == hipSuccess &&
           hipMalloc(&slabs[size_t(r)], slab_bytes) == hipSuccess;
      if (ok && devices[r] != devices[0]) {
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, devices[r], devices[0]);
        if (can) (void)hipDeviceEnablePeerAccess(devices[0], 0);  // (already enabled is fine; without peer access the copy is staged)
        (void)hipGetLastError();
      }
    }
    if (!ok) {
      set_error("per-device setup failed");
      break;
    }
    // all devices render concurrently; each slab goes to devices[0] on the rendering device's own stream as soon as it is ready
    for (int r = 0; r < num_devices && ok; r++) {
      char* dst = static_cast<char*>(gathered) + size_t(r) * slab_bytes;
      ok = pine_gpu_plan_launch_packed(plans[size_t(r)], slabs[size_t(r)], streams[size_t(r)]) == 0;
      if (!ok) break;
      // (a peer copy between a device and itself is refused: "invalid device ordinal")
      const hipError_t e = devices[r] == devices[0]
                               ? hipMemcpyAsync(dst, slabs[size_t(r)], slab_bytes, hipMemcpyDeviceToDevice, streams[size_t(r)])
                               : hipMemcpyPeerAsync(dst, devices[0], slabs[size_t(r)], devices[r], slab_bytes, streams[size_t(r)]);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(std::string("slab copy to the first device failed: ") + hipGetErrorString(e));
        err = "copy";
        ok = false;
      }
    }
    if (!ok) {
      if (err.empty()) set_error(std::string("launch failed: ") + pine_gpu_last_error());
      break;
    }
    for (int r = 0; r < num_devices && ok; r++) ok = pine_gpu_plan_check(plans[size_t(r)]) == 0;  // waits for the stream, reports bail-outs
    if (!ok) break;
    if (pine_gpu_film_unpack(w, h, num_devices, devices[0], gathered, d_film, nullptr)) break;
    if (hipMemcpy(film_out, d_film, film_bytes, hipMemcpyDeviceToHost) != hipSuccess) {
      set_error("film download failed");
      break;
    }
    rc = 0;
  } while (0);
  const std::string keep = rc ? pine_gpu_last_error() : "";
  for (int r = 0; r < num_devices; r++) {
    (void)hipSetDevice(devices[r]);
    if (plans[size_t(r)]) pine_gpu_plan_destroy(plans[size_t(r)]);
    if (slabs[size_t(r)]) (void)hipFree(slabs[size_t(r)]);
    if (streams[size_t(r)]) (void)hipStreamDestroy(streams[size_t(r)]);
  }
  (void)hipSetDevice(devices[0]);
  if (gathered) (void)hipFree(gathered);
  if (d_film) (void)hipFree(d_film);
  if (rc) set_error(keep);
  return rc;
}

/* SURVEY.md 8(b)'s form: bit d of device_mask selects HIP device d; shards are dealt to the selected devices in
 * ascending order. */
int pine_gpu_path_render_multi(pine_gpu_scene* scene, const pine_gpu_render_params* prm, uint64_t device_mask, float* film_out) {
  int list[64], n = 0;
  for (int d = 0; d < 64; d++)
    if (device_mask & (1ull << d)) list[n++] = d;
  if (n == 0) {
    set_error("empty device mask");
    return -1;
  }
  return pine_gpu_path_render_devices(scene, prm, list, n, film_out);
}

// ---- device unit-test hooks -------------------------------------------------------------------
static int need_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("no HIP device available");
    return -1;
  }
  HIP_OK(hipSetDevice(device));
  return 0;
}
int pine_gpu_test_sincos(int device, const float* x, int64_t n, float* s, float* c) {
  if (need_device(device)) return -1;
  float *dx, *ds, *dc;
  HIP_OK(hipMalloc((void**)&dx, n * 4));
  HIP_OK(hipMalloc((void**)&ds, n * 4));
  HIP_OK(hipMalloc((void**)&dc, n * 4));
  HIP_OK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(test_sincos_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, dx, (long long)n, ds, dc);
  HIP_OK(hipMemcpy(s, ds, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(c, dc, n * 4, hipMemcpyDeviceToHost));
  hipFree(dx);
  hipFree(ds);
  hipFree(dc);
  return 0;
}
int pine_gpu_test_powlog(int device, const float* x, const float* y, int64_t n, float* pw, float* lg) {
  if (need_device(device)) return -1;
  float *dx, *dy, *dp, *dl;
  HIP_OK(hipMalloc((void**)&dx, n * 4));
  HIP_OK(hipMalloc((void**)&dy, n * 4));
  HIP_OK(hipMalloc((void**)&dp, n * 4));
  HIP_OK(hipMalloc((void**)&dl, n * 4));
  HIP_OK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(test_powlog_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, dx, dy, (long long)n, dp, dl);
  HIP_OK(hipMemcpy(pw, dp, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(lg, dl, n * 4, hipMemcpyDeviceToHost));
  hipFree(dx);
  hipFree(dy);
  hipFree(dp);
  hipFree(dl);
  return 0;
}
int pine_gpu_test_sampler(int device, int spp_req, float* out, int64_t capacity) {
  if (need_device(device)) return -1;
  TableBlob tables_blob;
  if (load_tables(tables_blob)) return -1;
  const std::vector<uint8_t>& g_tables = *tables_blob;
  const int spp = effective_spp(spp_req);
  const int64_t need = int64_t(6) * spp * (260 + 270);
  if (capacity < need) {
    set_error("capacity too small");
    return -1;
  }
  int k = 0;
  while ((1 << k) < spp) k++;
  uint8_t* dt;
  float* dout;
  HIP_OK(hipMalloc((void**)&dt, 65536 + 262144));
  {
    const std::vector<uint8_t> st = transposed_sobol(g_tables);
    HIP_OK(hipMemcpy(dt, st.data(), 65536, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMemcpy(dt + 65536, g_tables.data() + 65536 + size_t(k) * 262144, 262144, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc((void**)&dout, need * 4));
  DTables T{dt, dt + 65536, dt + 65536 + 131072, nullptr, nullptr, 0};
  hipLaunchKernelGGL(test_sampler_kernel, dim3(6), dim3(64), 0, 0, T, spp, dout);
  HIP_OK(hipMemcpy(out, dout, need * 4, hipMemcpyDeviceToHost));
  hipFree(dt);
  hipFree(dout);
  return 0;
}
int pine_gpu_test_rng(int device, uint64_t* out, int64_t capacity) {
  if (need_device(device)) return -1;
  if (capacity < 6 * 19) {
    set_error("capacity too small");
    return -1;
  }
  unsigned long long* d;
  HIP_OK(hipMalloc((void**)&d, 6 * 19 * 8));
  hipLaunchKernelGGL(test_rng_kernel, dim3(1), dim3(64), 0, 0, d);
  HIP_OK(hipMemcpy(out, d, 6 * 19 * 8, hipMemcpyDeviceToHost));
  hipFree(d);
  return 0;
}
int pine_gpu_test_shapes(pine_gpu_scene* scene, int device, const float* rays, int64_t nrays, float* out,
                         int64_t capacity) {
  if (!scene || !rays || !out) {
    set_error("null argument");
    return -1;
  }
  if (need_device(device)) return -1;
  SceneHost& H = scene_host(scene);
  std::vector<DShape> shapes;
  for (auto& g : H.geometries)
    if (g.shape.kind != SHAPE_MESH) shapes.push_back(g.shape);
  const int64_t need = int64_t(shapes.size()) * nrays * 11;
  if (capacity < need) {
    set_error("capacity too small");
    return -1;
  }
  DShape* ds;
  float *dr, *dout;
  if (upload(ds, shapes)) return -1;
  HIP_OK(hipMalloc((void**)&dr, nrays * 32));
  HIP_OK(hipMemcpy(dr, rays, nrays * 32, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc((void**)&dout, std::max<int64_t>(need, 1) * 4));
  const long long total = (long long)shapes.size() * nrays;
  if (total > 0)
    hipLaunchKernelGGL(test_shapes_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, 0, ds,
                       int(shapes.size()), dr, (long long)nrays, dout);
  HIP_OK(hipMemcpy(out, dout, need * 4, hipMemcpyDeviceToHost));
  hipFree(ds);
  hipFree(dr);
  hipFree(dout);
  return 0;
}

}  // extern "C"
