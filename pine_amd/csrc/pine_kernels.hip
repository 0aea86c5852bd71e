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
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <algorithm>
#include <vector>

#include <dlfcn.h>
#include <signal.h>
#include <sys/stat.h>
#include <thread>

#include "../../include/pine_gpu.h"
#include "pine_device.h"
#include "pine_specialize.h"
#include "pine_variants.h"
#include "pine_host.h"

struct pine_gpu_scene;
namespace pine_gpu {
SceneHost& scene_host(pine_gpu_scene* s);
}  // namespace pine_gpu
#include "pine_kernels_device.h"
#include "pine_bvh_build_device.h"
#include "pine_embree_order.h"
#include "../data/rcpps_table.h"
namespace pine_gpu {

// ---- the BVH build on the device: host orchestration (one decide / scan / split triple per level) ----
static int device_build(std::vector<BuildPrim>& prims, const std::vector<BuildTask>& roots, FlatAccel& A, int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    (void)hipGetLastError();
    return -1;
  }
  if (hipSetDevice(device) != hipSuccess) return -1;
  const size_t n = prims.size();
  BuildPrim *d_prims = nullptr, *d_scratch = nullptr;
  BuildTask *d_tasks[2] = {nullptr, nullptr};
  BuildDecision* d_dec = nullptr;
  int *d_rank = nullptr, *d_perm = nullptr, *d_src = nullptr, *d_counts = nullptr;
  unsigned char* d_pred = nullptr;
  DNode* d_nodes = nullptr;
  DBvh* d_bvhs = nullptr;
  int rc = -1;
  auto ok = [](hipError_t e) { return e == hipSuccess; };
  do {
    const size_t max_tasks = std::max<size_t>(n, roots.size()) + 2;
    if (!ok(hipMalloc((void**)&d_prims, n * sizeof(BuildPrim))) || !ok(hipMalloc((void**)&d_scratch, n * sizeof(BuildPrim))) ||
        !ok(hipMalloc((void**)&d_tasks[0], max_tasks * sizeof(BuildTask))) || !ok(hipMalloc((void**)&d_tasks[1], max_tasks * sizeof(BuildTask))) ||
        !ok(hipMalloc((void**)&d_dec, max_tasks * sizeof(BuildDecision))) || !ok(hipMalloc((void**)&d_rank, max_tasks * sizeof(int))) ||
        !ok(hipMalloc((void**)&d_perm, n * sizeof(int))) || !ok(hipMalloc((void**)&d_src, n * sizeof(int))) || !ok(hipMalloc((void**)&d_pred, n)) || !ok(hipMalloc((void**)&d_nodes, (n + 1) * sizeof(DNode))) ||
        !ok(hipMalloc((void**)&d_bvhs, A.bvhs.size() * sizeof(DBvh))) || !ok(hipMalloc((void**)&d_counts, 4 * sizeof(int))))
      break;
    if (!ok(hipMemcpy(d_prims, prims.data(), n * sizeof(BuildPrim), hipMemcpyHostToDevice)) ||
        !ok(hipMemcpy(d_tasks[0], roots.data(), roots.size() * sizeof(BuildTask), hipMemcpyHostToDevice)) ||
        !ok(hipMemcpy(d_bvhs, A.bvhs.data(), A.bvhs.size() * sizeof(DBvh), hipMemcpyHostToDevice)))
      break;
    int counts[4] = {0, 0, 0, 0};  // nodes so far | tasks of the next level | largest range of the next level
    int ntasks = int(roots.size()), node_base = 0, max_n = 0;
    for (const BuildTask& t : roots) max_n = std::max(max_n, t.end - t.begin);
    if (!ok(hipFuncSetAttribute((const void*)bvh_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBuildLdsPrims * 9 + 16))) break;
    bool failed = false;
    for (int cur = 0; ntasks > 0; cur ^= 1) {
      if (!ok(hipMemcpy(d_counts, counts, sizeof counts, hipMemcpyHostToDevice))) { failed = true; break; }
      hipLaunchKernelGGL(bvh_decide_kernel, dim3(ntasks), dim3(kBuildWave), 0, 0, d_prims, d_tasks[cur], ntasks, d_dec);
      hipLaunchKernelGGL(bvh_scan_kernel, dim3(1), dim3(1024), 0, 0, d_tasks[cur], ntasks, d_dec, d_rank, d_nodes, d_bvhs, d_counts);
      const int lds_prims = std::min(max_n, kBuildLdsPrims);
      hipLaunchKernelGGL(bvh_split_kernel, dim3(ntasks), dim3(kSplitBlock), size_t(lds_prims) * 8 + ((size_t(lds_prims) + 15) & ~size_t(15)), 0, d_prims, d_scratch,
                         d_tasks[cur], ntasks, d_dec, d_rank, node_base, d_nodes, d_tasks[cur ^ 1], d_perm, d_src, d_pred, lds_prims, d_counts + 2);
      if (!ok(hipGetLastError()) || !ok(hipMemcpy(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost))) { failed = true; break; }
      node_base = counts[0];
      ntasks = counts[1];
      max_n = counts[2];
      counts[1] = counts[2] = 0;
      if (size_t(node_base) > n || size_t(ntasks) > max_tasks) { failed = true; break; }  // (cannot happen: a binary tree over n leaves)
    }
    if (failed) break;
    A.nodes.resize(size_t(node_base));
    if (node_base && !ok(hipMemcpy(A.nodes.data(), d_nodes, size_t(node_base) * sizeof(DNode), hipMemcpyDeviceToHost))) break;
    if (!ok(hipMemcpy(prims.data(), d_prims, n * sizeof(BuildPrim), hipMemcpyDeviceToHost)) ||
        !ok(hipMemcpy(A.bvhs.data(), d_bvhs, A.bvhs.size() * sizeof(DBvh), hipMemcpyDeviceToHost)))
      break;
    rc = 0;
  } while (0);
  (void)hipGetLastError();
  for (void* q : {(void*)d_prims, (void*)d_scratch, (void*)d_tasks[0], (void*)d_tasks[1], (void*)d_dec, (void*)d_rank, (void*)d_perm, (void*)d_src, (void*)d_pred,
                  (void*)d_nodes, (void*)d_bvhs, (void*)d_counts})
    if (q) (void)hipFree(q);
  return rc;
}
static const bool g_device_builder_installed = (g_device_builder = &device_build, true);

// Compiled specialisations (pine_variants.h): instantiated by the PART translation units (pine_kernels_part.hip, built in
// parallel), merged here in `order`; the host takes the first one that covers a scene.
struct VariantTables {
  std::vector<PineKernelVariant> queue, mega;
  VariantTables() {
    using PartFn = const PineKernelVariant* (*)(int*, int*);
    static const PartFn parts[kPineKernelParts] = {pine_gpu_kernel_part_0, pine_gpu_kernel_part_1, pine_gpu_kernel_part_2, pine_gpu_kernel_part_3,
                                                   pine_gpu_kernel_part_4, pine_gpu_kernel_part_5, pine_gpu_kernel_part_6, pine_gpu_kernel_part_7};
    for (PartFn f : parts) {
      int nq = 0, nm = 0;
      const PineKernelVariant* t = f(&nq, &nm);
      queue.insert(queue.end(), t, t + nq);
      mega.insert(mega.end(), t + nq, t + nq + nm);
    }
    auto by_order = [](const PineKernelVariant& a, const PineKernelVariant& b) { return a.order < b.order; };
    std::sort(queue.begin(), queue.end(), by_order);
    std::sort(mega.begin(), mega.end(), by_order);
  }
};
static const VariantTables& variant_tables() {
  static const VariantTables t;
  return t;
}
#define kVariants (variant_tables().mega)
#define kQueueVariants (variant_tables().queue)
#define kNumVariants (int(variant_tables().mega.size()))
#define kNumQueueVariants (int(variant_tables().queue.size()))

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
  const int tile = film_tile_of(W, ltile);
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
  // (the slab is tile-major in the shard's NATURAL tile order, whatever order the launch works in: tile_order)
  const size_t out_index = packed ? size_t(tile / W.shard_world) * 64u + size_t(p) : size_t(py) * film_w + px;
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
__global__ void test_atan_kernel(const float* y, const float* x, long long n, float* at2, float* ac) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) {
    at2[i] = patan2(y[i], x[i]);
    ac[i] = pacos(x[i]);
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
// The primitives a ray's traversal tests, in order, and its result: the nested loops of the scene-in-LDS variants
// (FLAT = false: scene_traverse / mesh_traverse) or the flat state machine of the F_LDS_TOP variants (pine_trav.h).
// One thread per ray, 64 per block; out: per ray `cap` words closest (count, words...), 4 result words (hit, geometry,
// triangle, tmax bits), `cap` words any-hit, 1 result word.
// (MODE 2: the closest-hit query in EmbreeAccel's order, PINE_GPU_FLAG_ORDER_EMBREE; the any-hit query is the nested loops')
template <int MODE>
__global__ void __launch_bounds__(64) test_traverse_kernel(DeviceScene S, const float* rays, long long nrays, int cap, unsigned* out) {
  constexpr bool FLAT = MODE == 1;
  constexpr unsigned F = FLAT ? (F_ALL | F_LDS_TOP) : MODE == 2 ? (F_ALL | F_EMBREE) : F_ALL;
  using StackT = typename std::conditional<FLAT, unsigned short, int>::type;
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  StackT* const stack = reinterpret_cast<StackT*>(lds_raw) + threadIdx.x;
  SceneView V;
  V.tri_verts = S.tri_verts, V.tri_leaf = S.tri_leaf, V.tri_attrs = S.tri_attrs;
  V.lds_nodes = nullptr, V.lds_node_count = 0, V.lds_tri_entries = nullptr, V.lds_tri_verts = nullptr;
  V.stack_top = S.stack_top, V.num_shapes = S.num_shapes;
  V.leaf = S.leaf, V.nodes = S.nodes, V.shapes = S.shapes, V.materials = S.materials, V.bvhs = S.bvhs, V.prims = nullptr;
  V.lights = S.lights, V.node_ops = S.node_ops;
  V.etree = reinterpret_cast<const EmbreeNode*>(reinterpret_cast<const char*>(S.blob) + S.off_etree), V.etree_root = S.etree_root;
  V.emesh = reinterpret_cast<const int*>(reinterpret_cast<const char*>(S.blob) + S.off_emesh), V.num_emesh = S.num_emesh;
  V.rcpps = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(S.blob) + S.off_rcpps);
  const long long i = blockIdx.x * 64ll + threadIdx.x;
  const bool live = i < nrays;
  const float* q = rays + (live ? i : 0) * 8;
  unsigned* o = out + (live ? i : 0) * (2ll * cap + 5);
  for (int pass = 0; pass < 2; pass++) {
    DRay ray{f3{q[0], q[1], q[2]}, f3{q[3], q[4], q[5]}, q[6], q[7]};
    TravLog log{o + (pass ? cap + 4 : 0) + 1, 0, cap - 1};
    bool hit = false;
    int geom = 0, prim = 0;
    if constexpr (FLAT) {
      TravState ts;
      trav_begin(V, ts);
      if (!live) ts.done = 1;
      const DRayOct oct = make_oct(ray);
      if (pass == 0) trav_trips<false, F, 64>(V, ray, oct, ts, stack, 0, 1 << 30, nullptr, &log);
      else trav_trips<true, F, 64>(V, ray, oct, ts, stack, 0, 1 << 30, nullptr, &log);
      hit = ts.hit_geom >= 0;
      geom = ts.hit_geom, prim = ts.hit_prim;
    } else if (live) {
      hit = pass == 0 ? scene_traverse<false, F, 64>(V, ray, stack, geom, prim, &log) : scene_traverse<true, F, 64>(V, ray, stack, geom, prim, &log);
    }
    if (live) {
      o[pass ? cap + 4 : 0] = unsigned(log.n);
      if (pass == 0) {
        o[cap] = hit ? 1u : 0u;
        o[cap + 1] = hit ? unsigned(geom & kPrimIndexMask) : 0u;
        // (a mesh hit reports the triangle's index within its mesh, as the reference does; elsewhere the word is unused: 0)
        const bool on_mesh = hit && (geom >> kPrimKindShift) == SHAPE_MESH;
        o[cap + 2] = on_mesh ? unsigned(prim - S.bvhs[as_int(S.shapes[geom & kPrimIndexMask].f[2])].prim_base) : 0u;
        o[cap + 3] = __float_as_uint(ray.tmax);
      } else {
        o[2 * cap + 4] = hit ? 1u : 0u;
      }
    }
  }
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
      (void)hipGetLastError(); /* reported here: do not leave it sticky for the caller's next HIP user */ \
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

// Device memory of plans comes from a process-wide pool: hipMalloc / hipFree of the big per-plan buffers (the per-sample
// radiance buffer is 1.7 GB for a 640 x 640 x 256 render, 6.8 GB for 1920 x 1080) cost 30 - 90 ms per plan, which is most of
// what a ONE-SHOT render (pine_gpu_path_render: create, launch, destroy -- what PathIntegrator::render does) spends outside
// its kernels.  A destroyed plan's blocks of 256 KB and more go to a per-device free list instead and the next plan takes
// the smallest one that fits within 25 %; at most $PINE_GPU_POOL_MB (default 16 384; 0: no pool) are kept,
// pine_gpu_release_cached_memory() frees them.  No kernel reads a buffer before writing it (hipMalloc does not clear either).
struct DevicePool {
  struct Block {
    void* p;
    size_t bytes;
    int device;
  };
  std::mutex mu;
  std::vector<Block> free_blocks;
  std::map<void*, Block> live;  // pooled-size allocations handed out
  size_t pooled = 0, cap = size_t(16384) << 20;
  DevicePool() {
    if (const char* e = getenv("PINE_GPU_POOL_MB")) cap = size_t(atoll(e) > 0 ? atoll(e) : 0) << 20;
  }
  static DevicePool& get() {
    static DevicePool* q = new DevicePool();  // (never destroyed: plans may be destroyed during static destruction)
    return *q;
  }
  static constexpr size_t kMinPooled = size_t(256) << 10;
  hipError_t alloc(void** out, size_t bytes) {
    *out = nullptr;
    if (bytes < kMinPooled || cap == 0) return hipMalloc(out, bytes);
    int device = 0;
    (void)hipGetDevice(&device);
    const size_t want = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    {
      std::lock_guard<std::mutex> lock(mu);
      int best = -1;
      for (size_t i = 0; i < free_blocks.size(); i++) {
        const Block& b = free_blocks[i];
        if (b.device != device || b.bytes < want || b.bytes > want + want / 4) continue;
        if (best < 0 || b.bytes < free_blocks[size_t(best)].bytes) best = int(i);
      }
      if (best >= 0) {
        const Block b = free_blocks[size_t(best)];
        free_blocks.erase(free_blocks.begin() + best);
        pooled -= b.bytes;
        live[b.p] = b;
        *out = b.p;
        return hipSuccess;
      }
    }
    hipError_t e = hipMalloc(out, want);
    if (e != hipSuccess) {  // memory is short: give the pool back and try once more
      release_all();
      (void)hipGetLastError();
      e = hipMalloc(out, want);
    }
    if (e == hipSuccess) {
      std::lock_guard<std::mutex> lock(mu);
      live[*out] = Block{*out, want, device};
    }
    return e;
  }
  void free(void* p) {
    if (!p) return;
    Block b{nullptr, 0, 0};
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = live.find(p);
      if (it != live.end()) {
        b = it->second;
        live.erase(it);
        if (pooled + b.bytes <= cap) {
          free_blocks.push_back(b);
          pooled += b.bytes;
          return;
        }
      }
    }
    (void)hipFree(p);  // (a small allocation, or the pool is full)
  }
  void release_all() {
    std::vector<Block> blocks;
    {
      std::lock_guard<std::mutex> lock(mu);
      blocks.swap(free_blocks);
      pooled = 0;
    }
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (const Block& b : blocks) {
      (void)hipSetDevice(b.device);
      (void)hipFree(b.p);
    }
    (void)hipSetDevice(keep);
  }
};
#define POOL_ALLOC(ptr, bytes) DevicePool::get().alloc((void**)&(ptr), (bytes))

template <class T>
static int upload(T*& dptr, const std::vector<T>& v) {
  dptr = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIP_OK(POOL_ALLOC(dptr, bytes));
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
  if (!A.tri_attrs.empty() && A.tri_attrs.size() != size_t(n_tris) * 16) return bad("tri_attrs size");
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
  uint4* d_tri_packets = nullptr;
  uint8_t* d_halton = nullptr;
  float* d_tri_attrs = nullptr;
  uint8_t* d_tables = nullptr;
  int variant = -1;
  int queue_variant = -1;   // >= 0: the stage-queued kernel is used instead of path_trace_kernel
  const PineFastVariant* fast = nullptr;  // PINE_GPU_FLAG_FAST: the declared-tolerance variant that runs instead (pine_kernels_fast.hip)
  uint32_t* d_ctxg = nullptr;
  ulonglong2* d_ckpt = nullptr;
  // The RNG checkpoints are a function of the film partition and the sample counts alone: the plan's FIRST launch computes them,
  // later launches reuse the table (they wait for `ckpt_done` when they run on another stream).  $PINE_GPU_CKPT_EVERY_LAUNCH=1: as
  // before round 4, every launch recomputes it (measurement aid).
  bool ckpt_valid = false, ckpt_every_launch = false;
  hipEvent_t ckpt_done = nullptr;
  hipStream_t ckpt_stream = nullptr;
  float* d_vertex_log = nullptr;        // test hook (pine_gpu_plan_vertex_log)
  int* d_tile_order = nullptr;          // tile classes (WorkParams::tile_order), or null
  std::vector<int> tile_order;          // ... its host copy (empty: local tile t is film tile t * shard_world + shard_rank)
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
  bool accel_on_device = false;
  // PINE_GPU_FLAG_SPECIALIZE: the queue kernel compiled for this scene (pine_specialize.h); null: the precompiled variant
  std::shared_ptr<struct LoadedKernel> spec_loaded;  // (shared with every plan of this geometry on this device: LoadedKernels)
  hipModule_t spec_module = nullptr;
  hipFunction_t spec_fn = nullptr;
  unsigned spec_features = 0;  // ... its feature set (the scene's own), and whether the scene's BVH is baked in
  bool spec_baked = false;
  KernelRequest spec_request;  // what to compile (filled at plan creation)
  std::chrono::steady_clock::time_point spec_t0;
  bool spec_explicit = false;  // the caller asked for the scene's kernel (PINE_GPU_FLAG_SPECIALIZE): failures are errors
  int spec_source = 0;         // kSpecSource*: where the scene's kernel came / comes from
  // background build (the default mode, PINE_GPU_FLAG_SPECIALIZE_ASYNC): a job of the process-wide queue; a launch adopts its result
  std::atomic<int> spec_state{0};  // kSpecNone / kSpecBuilding / kSpecAdopted / kSpecFailed
  std::shared_ptr<struct SpecJob> spec_job;
  std::string spec_async_error;
  float specialize_ms = 0.0f;
};

// Scene-specialised kernels (pine_specialize.h): the stage-queued kernel compiled FOR THIS SCENE.
//  (1) its exact feature set: the precompiled variants are a handful of supersets (pine_variants.h) -- a scene of spheres
//      under a point light runs the everything-but-Subsurface kernel and pays for every shape kind, node programs and the
//      Sobol sampler in registers (38 spilled VGPRs).  `need` is what plan_build found in the scene; the LDS layout flags
//      (and F_SSS, which sizes the per-context records) stay those of the chosen variant, so every buffer size computed
//      from it stays right.
//  (2) if the scene has no meshes and its BVH is small enough to unroll: the BVH and primitive records baked in; one mesh
//      under a small top level: the top level as code.
// Nothing to gain (the variant IS the exact set, nothing to bake): the precompiled kernel runs.
//
// Modes.  DEFAULT (no flag): automatic and never in the caller's way -- a code object already in the cache is loaded at plan
// creation (about a millisecond); otherwise the compiler runs in the BACKGROUND (a process-wide queue of at most
// kSpecWorkers compiler children, keyed by content, shared by every plan that wants the same kernel and outliving the plan
// that asked first) while the precompiled kernel renders; the first launch after the build has finished -- of this plan or of
// any later plan of the same geometry -- runs the scene's own kernel.  Nothing can fail because of it: no compiler, no
// headers, no cache directory, a full queue all leave the precompiled kernel in place (plan stats: specialized 0 or -1).
// PINE_GPU_FLAG_SPECIALIZE (or $PINE_GPU_SPECIALIZE=1): the caller WANTS the scene's kernel -- plan creation waits for the
// compiler and a kernel that cannot be built fails the plan; with _ASYNC the build runs in the background as above but a
// failure is still reported (specialized == -1).  PINE_GPU_FLAG_NO_SPECIALIZE / $PINE_GPU_SPECIALIZE=0: precompiled only.
enum : int { kSpecNone = 0, kSpecBuilding, kSpecBuilt, kSpecAdopted, kSpecFailed };
enum : int { kSpecSourceNone = 0, kSpecSourceCache = 1, kSpecSourceCompiledHere = 2, kSpecSourceBackground = 3 };

namespace pine_gpu {
AbiFingerprint abi_fingerprint() {
  return AbiFingerprint{sizeof(DeviceScene), sizeof(WorkParams), sizeof(Counters), sizeof(DNode), sizeof(DShape), sizeof(DMaterial), sizeof(DLight),
                        offsetof(WorkParams, total_items), offsetof(DeviceScene, cam), kQFields, kQWinDwords, int(QC_WORDS), kQTokenDwords,
                        kTravRecordDwords, kQCtxGlobalDwordsPlain, kQCtxGlobalDwordsSss};
}
}  // namespace pine_gpu

// The background compile queue.  A job is one code object (content key); plans hold a shared_ptr and poll `state`.  Workers
// are started on demand, run jobs in order, skip jobs nobody waits for any more, and exit when the queue is empty.  The
// singleton is never destroyed (worker threads may outlive static destruction); at process exit an atexit handler raises
// `closing`, ends running compilers (their own process groups) and joins the workers, so nothing of ours runs while the
// runtime goes down and no half-written build directory is left behind.
struct SpecJob {
  KernelRequest req;
  std::atomic<int> state{kSpecBuilding};
  std::atomic<int> waiters{0};
  std::string error;  // (written before `state` becomes kSpecFailed)
  float compile_ms = 0.0f;
};
struct SpecQueue {
  static constexpr int kSpecWorkers = 2, kMaxPending = 6;
  std::mutex mu;
  std::map<std::string, std::shared_ptr<SpecJob>> jobs;  // by key: every job ever started in this process
  std::deque<std::shared_ptr<SpecJob>> pending;
  std::vector<std::thread> workers;
  int running = 0;
  std::atomic<int> children[kSpecWorkers];
  std::atomic<bool> closing{false};
  static SpecQueue& get() {
    static SpecQueue* q = [] {
      SpecQueue* x = new SpecQueue();
      for (auto& c : x->children) c.store(0);
      atexit([] { SpecQueue::get().shutdown(); });
      return x;
    }();
    return *q;
  }
  // the job for `req` (already keyed): an existing one, or a new one queued for a worker; null when the queue is full
  // (`retry_failed`: a kernel whose build failed earlier in this process is tried again -- the caller asked for it by flag;
  //  the automatic mode does not spend a compiler run per plan on a kernel that does not build)
  std::shared_ptr<SpecJob> submit(const KernelRequest& req, bool retry_failed) {
    std::lock_guard<std::mutex> lock(mu);
    if (closing.load()) return nullptr;
    auto it = jobs.find(req.key);
    if (it != jobs.end() && (it->second->state.load() != kSpecFailed || !retry_failed)) return it->second;
    if (int(pending.size()) >= kMaxPending) return nullptr;
    auto job = std::make_shared<SpecJob>();
    job->req = req;
    jobs[req.key] = job;
    pending.push_back(job);
    if (running < kSpecWorkers) {
      const int slot = running++;
      workers.emplace_back([this, slot] { work(slot); });
    }
    return job;
  }
  void work(int slot) {
    for (;;) {
      std::shared_ptr<SpecJob> job;
      {
        std::lock_guard<std::mutex> lock(mu);
        while (!pending.empty() && !job) {
          job = pending.front();
          pending.pop_front();
          if (job->waiters.load() == 0 || closing.load()) {  // nobody wants it any more
            job->error = "cancelled";
            job->state.store(kSpecFailed);
            jobs.erase(job->req.key);
            job.reset();
          }
        }
        if (!job) {
          running--;
          return;
        }
      }
      const auto t0 = std::chrono::steady_clock::now();
      std::string err;
      struct stat st;
      const bool ok = (stat(job->req.path.c_str(), &st) == 0 && st.st_size > 0) || kernel_compile(job->req, err, &children[slot]);
      job->compile_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
      job->error = err;
      job->req.baked.clear();  // (the text is no longer needed: waiting plans keep their own copy)
      job->req.baked.shrink_to_fit();
      job->state.store(ok ? kSpecBuilt : kSpecFailed);
      if (ok) {  // later plans find the code object on disk; only failures are remembered
        std::lock_guard<std::mutex> lock(mu);
        jobs.erase(job->req.key);
      }
    }
  }
  void shutdown() {
    closing.store(true);
    for (auto& c : children) {
      const int pid = c.load();
      if (pid > 0) kill(-pid, SIGKILL);  // the compiler's process group
    }
    std::vector<std::thread> w;
    {
      std::lock_guard<std::mutex> lock(mu);
      w.swap(workers);
    }
    for (auto& t : w)
      if (t.joinable()) t.join();
  }
};

// A code object loaded into a device's context, shared by the plans that run it: hipModuleLoadData of a scene's kernel is a
// millisecond or two, which a one-shot render (create, launch, destroy) of the same geometry would pay on every call.  The
// process keeps the last kMaxLoaded of them per (device, content key); a module is unloaded when the table has dropped it and
// the last plan that launches it is gone.
struct LoadedKernel {
  int device = 0;
  std::string key, image;  // (the image is kept for the module's lifetime: the runtime may build the program lazily from it)
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  ~LoadedKernel() {
    if (module) {
      int keep = 0;
      (void)hipGetDevice(&keep);
      (void)hipSetDevice(device);
      (void)hipModuleUnload(module);
      (void)hipSetDevice(keep);
    }
  }
};
struct LoadedKernels {
  static constexpr size_t kMaxLoaded = 16;
  std::mutex mu;
  std::deque<std::shared_ptr<LoadedKernel>> recent;
  static LoadedKernels& get() {
    static LoadedKernels* q = new LoadedKernels();  // (never destroyed: unloading modules while the runtime goes down is not safe)
    return *q;
  }
  std::shared_ptr<LoadedKernel> find(int device, const std::string& key) {
    std::lock_guard<std::mutex> lock(mu);
    for (auto& k : recent)
      if (k->device == device && k->key == key) return k;
    return nullptr;
  }
  void remember(const std::shared_ptr<LoadedKernel>& k) {
    std::lock_guard<std::mutex> lock(mu);
    recent.push_front(k);
    if (recent.size() > kMaxLoaded) recent.pop_back();
  }
  void forget(const std::string& key) {
    std::lock_guard<std::mutex> lock(mu);
    for (auto it = recent.begin(); it != recent.end();)
      it = (*it)->key == key ? recent.erase(it) : it + 1;
  }
};

static int plan_adopt_kernel(pine_gpu_plan* p, bool compile_here);
static int plan_specialize(pine_gpu_plan* p, const FlatAccel& A, const std::vector<DShape>& shapes, const std::vector<int>& packed_prims,
                           const pine_gpu_render_params* prm, unsigned need) {
  // explicit: the caller asked for the scene's kernel (failures are errors); automatic: the default (failures are silent)
  bool explicit_want = (prm->flags & PINE_GPU_FLAG_SPECIALIZE) != 0;
  bool automatic = !explicit_want && !(prm->flags & PINE_GPU_FLAG_NO_SPECIALIZE);
  if (const char* e = getenv("PINE_GPU_SPECIALIZE")) {
    if (atoi(e) != 0) explicit_want = true, automatic = false;
    else explicit_want = automatic = false;
  }
  if ((!explicit_want && !automatic) || p->queue_variant < 0 || (prm->flags & (PINE_GPU_FLAG_FAST | PINE_GPU_FLAG_VERTEX_LOG))) return 0;
  const PineKernelVariant& V = kQueueVariants[p->queue_variant];
  const auto t0 = std::chrono::steady_clock::now();
  const unsigned kLayout = F_LDS_SCENE | F_LDS_TOP | F_LDS_REST | F_XSTAGE | F_SSS;
  unsigned exact = (V.features & kLayout) | need;
  std::string baked;
  // (a baked scene IS pine's visiting order as code: the EmbreeAccel order mode keeps to the feature-set level)
  if (!(prm->flags & (PINE_GPU_FLAG_SPECIALIZE_NO_BAKE | PINE_GPU_FLAG_ORDER_EMBREE)) && getenv("PINE_GPU_SPECIALIZE_NO_BAKE") == nullptr) {
    if (!(V.features & F_XSTAGE) && A.top_prim_begin == 0) baked = generate_baked_scene(A, shapes, packed_prims);
    // one mesh under a small top level, traversal stages (C5's class): the top level as code, the mesh left to the flat traversal
    else if ((V.features & F_XSTAGE) && A.bvhs.size() == 2 && getenv("PINE_GPU_SPECIALIZE_NO_TOP") == nullptr)
      baked = generate_baked_scene(A, shapes, packed_prims, true);
  }
  if (baked.empty() && exact == V.features && getenv("PINE_GPU_SPECIALIZE_FORCE") == nullptr) return 0;  // (FORCE: experiments through $PINE_GPU_SPECIALIZE_EXTRA)
  if (!baked.empty()) exact |= F_BAKED;
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, prm->device));
  std::string arch = prop.gcnArchName;  // "gfx950:sramecc+:xnack-" -> "gfx950"
  if (arch.find(':') != std::string::npos) arch = arch.substr(0, arch.find(':'));
  KernelRequest& R = p->spec_request;
  R.baked = baked, R.features = exact, R.ctx = V.ctx, R.arch = arch;
  p->spec_t0 = t0;
  p->spec_explicit = explicit_want;
  std::string err;
  const int found = kernel_cache_lookup(R, err);
  if (found < 0) {
    if (!explicit_want) {  // (no headers / no cache directory: the precompiled kernel it is)
      R = KernelRequest();
      return 0;
    }
    set_error(err);
    return -1;
  }
  if (found == 1) {
    // in the cache already: load it now (a file the runtime refuses is recompiled -- here when the caller waits for the
    // scene's kernel anyway, in the background otherwise)
    p->spec_source = kSpecSourceCache;
    if (plan_adopt_kernel(p, explicit_want && !(prm->flags & PINE_GPU_FLAG_SPECIALIZE_ASYNC)) == 0) return 0;
    if (explicit_want && !(prm->flags & PINE_GPU_FLAG_SPECIALIZE_ASYNC)) return -1;
    // a code object the runtime refuses (cut short by a full disk, another ROCm's output): out of the way, and built afresh in
    // the background; a packaged one that cannot be removed is bypassed for the user's cache
    const bool packaged = R.packaged;
    (void)unlink(R.path.c_str());
    const int again = kernel_cache_lookup(R, err, packaged);
    if (again < 0 || (again == 1 && plan_adopt_kernel(p, false) != 0)) {
      p->spec_state.store(kSpecFailed);
      p->spec_async_error = again < 0 ? err : std::string(pine_gpu_last_error());
      return 0;
    }
    if (again == 1) return 0;
  } else if (explicit_want && !(prm->flags & PINE_GPU_FLAG_SPECIALIZE_ASYNC)) {
    p->spec_source = kSpecSourceCompiledHere;
    return plan_adopt_kernel(p, true);
  }
  // the compiler runs beside the first renders (host work only: files and a child process); a launch adopts the kernel once
  // it is there.  Until then -- and for good if the build fails -- the precompiled kernel renders the same film.
  p->spec_source = kSpecSourceBackground;
  p->spec_job = SpecQueue::get().submit(R, explicit_want);
  if (!p->spec_job) {
    p->spec_state.store(explicit_want ? kSpecFailed : kSpecNone);
    p->spec_async_error = "the background compile queue is full";
    return 0;
  }
  p->spec_job->waiters.fetch_add(1);
  p->spec_state.store(kSpecBuilding);
  return 0;
}

// Load the plan's scene-specialised kernel (compiling it first when `compile_here`) and make it the one that launches.
// (Two attempts when compiling here: a cached code object the runtime refuses -- a file cut short by a full disk, another
// ROCm's output -- is removed and compiled afresh, once; a packaged one that cannot be removed is bypassed.)
static int plan_adopt_kernel(pine_gpu_plan* p, bool compile_here) {
  KernelRequest& R = p->spec_request;
  bool skip_packaged = false;
  for (int attempt = 0;; attempt++) {
    std::string err;
    const int found = kernel_cache_lookup(R, err, skip_packaged);
    if (found < 0 || (found == 0 && (!compile_here || !kernel_compile(R, err)))) {
      set_error(found == 0 && !compile_here ? "scene specialisation: the code object is not in the cache" : err);
      return -1;
    }
    const bool from_cache = found == 1;
    if (auto loaded = LoadedKernels::get().find(p->device, R.key)) {  // this geometry's kernel is in this device's context already
      p->spec_loaded = loaded;
      p->spec_module = loaded->module, p->spec_fn = loaded->fn;
      break;
    }
    hipError_t e = hipErrorInvalidImage;
    auto k = std::make_shared<LoadedKernel>();
    k->device = p->device, k->key = R.key;
    if (read_file(R.path, k->image) && code_object_is_whole(k->image)) {
      e = hipModuleLoadData(&k->module, k->image.data());
      if (e == hipSuccess) e = hipModuleGetFunction(&k->fn, k->module, kernel_symbol(R.features, R.ctx).c_str());
    }
    if (e == hipSuccess) {
      LoadedKernels::get().remember(k);
      p->spec_loaded = k;
      p->spec_module = k->module, p->spec_fn = k->fn;
      break;
    }
    (void)hipGetLastError();
    k.reset();  // (unloads what was loaded)
    p->spec_module = nullptr, p->spec_fn = nullptr;
    if (attempt == 0 && compile_here && from_cache) {
      if (unlink(R.path.c_str()) != 0) skip_packaged = true;  // (a read-only install: compile into the user's cache instead)
      continue;
    }
    set_error("scene specialisation: the runtime does not load " + R.path + " (" + hipGetErrorString(e) + ")");
    return -1;
  }
  p->spec_features = R.features;
  p->spec_baked = !R.baked.empty();
  p->specialize_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - p->spec_t0).count();
  return 0;
}

// PINE_GPU_FLAG_SPECIALIZE_ASYNC / automatic mode: has the background build finished?  Called by every launch and by stats_get.
static void plan_poll_background(pine_gpu_plan* p) {
  if (p->spec_state.load() != kSpecBuilding || !p->spec_job) return;
  const int js = p->spec_job->state.load();
  if (js == kSpecBuilding) return;
  if (js == kSpecBuilt && plan_adopt_kernel(p, false) == 0) {
    p->spec_state.store(kSpecAdopted);
  } else {
    p->spec_async_error = js == kSpecFailed ? p->spec_job->error : std::string(pine_gpu_last_error());
    p->spec_state.store(kSpecFailed);
  }
  p->spec_job->waiters.fetch_sub(1);
  p->spec_job.reset();
}

static int plan_check_counters(const Counters& c) {
  if (c.bail_count == 0) return 0;
  static const char* const kWhat[] = {"?", "idle budget exhausted with work outstanding", "ring slot never filled", "item-pool lock never released",
                                      "work-item hand-out did not converge", "?", "scene-specialised kernel: a closest-hit ray with an unexpected tmax", "forced by PINE_GPU_FLAG_DEBUG_FORCE_BAIL"};
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
  if (p->spec_job) {  // (a background build goes on for the next plan of this geometry; a job nobody waits for is dropped from the queue)
    p->spec_job->waiters.fetch_sub(1);
    p->spec_job.reset();
  }
  (void)hipSetDevice(p->device);
  // the plan's buffers go back to the pool, which hands them to the next plan without the device-wide synchronisation hipFree
  // implies: the plan's last launch must have finished first
  if (p->launched) (void)hipStreamSynchronize(p->last_stream);
  DevicePool::get().free(p->d_blob);
  DevicePool::get().free(p->d_tri);
  DevicePool::get().free(p->d_tri_leaf);
  DevicePool::get().free(p->d_tri_packets);
  DevicePool::get().free(p->d_halton);
  DevicePool::get().free(p->d_tri_attrs);
  DevicePool::get().free(p->d_tables);
  DevicePool::get().free(p->d_ctxg);
  DevicePool::get().free(p->d_ckpt);
  if (p->ckpt_done) (void)hipEventDestroy(p->ckpt_done);
  DevicePool::get().free(p->d_tile_order);
  DevicePool::get().free(p->d_vertex_log);
  DevicePool::get().free(p->d_samples);
  DevicePool::get().free(p->d_fold);
  DevicePool::get().free(p->d_counters);
  p->spec_loaded.reset();  // (the module stays loaded while the process-wide table or another plan holds it)
  if (p->h_progress) (void)hipHostFree(p->h_progress);
  for (auto& slot : p->ev)
    for (auto& e : slot)
      if (e) (void)hipEventDestroy(e);
  delete p;
}

// HaltonSampler's tables for the device (sampler.cpp:39-62, lowdiscrepancy.cpp:5-17, primes.cpp): the first kHaltonDims
// primes, their prefix sums (PrimeSums) and, per prime p, a permutation of 0 .. p-1 -- the identity shuffled with ONE
// default-seeded RNG running through all the primes in order (shuffle(): element i swaps with i + next32u(p - i)), so the
// permutations of the first kHaltonDims primes are a prefix of the reference's table of 1000.  Derived, not stored.
struct HaltonHostTables {
  std::vector<int> primes_and_sums;  // [kHaltonDims] primes, [kHaltonDims] prefix sums
  std::vector<uint16_t> perms;
};
static const HaltonHostTables& halton_host_tables() {
  static const HaltonHostTables tables = [] {
    HaltonHostTables t;
    t.primes_and_sums.assign(size_t(2 * kHaltonDims), 0);
    int count = 0, sum = 0;
    for (int n = 2; count < kHaltonDims; n++) {
      bool prime = true;
      for (int d = 2; d * d <= n && prime; d++) prime = n % d != 0;
      if (!prime) continue;
      t.primes_and_sums[size_t(count)] = n;
      t.primes_and_sums[size_t(kHaltonDims + count)] = sum;
      sum += n;
      count++;
    }
    t.perms.resize(size_t(sum));
    DRng rng = rng_seed(0);  // `RNG rng;`
    for (int k = 0; k < kHaltonDims; k++) {
      const int p = t.primes_and_sums[size_t(k)];
      uint16_t* perm = &t.perms[size_t(t.primes_and_sums[size_t(kHaltonDims + k)])];
      for (int j = 0; j < p; j++) perm[j] = uint16_t(j);
      for (int j = 0; j < p; j++) {
        const uint64_t u = rng_next64(rng);  // RNG::next32u(n) = uint32(u ^ (u >> 32)) % n  (rng.h:107-114)
        const uint32_t other = uint32_t(j) + uint32_t(u ^ (u >> 32)) % uint32_t(p - j);
        std::swap(perm[j], perm[other]);
      }
    }
    return t;
  }();
  return tables;
}

// The mesh triangles as LDS-sized packets (DeviceScene::tri_packets): per leaf-ordered triangle an 8-byte entry -- three
// 16-bit numbers into the table of the scene's DISTINCT vertices (by bit pattern) and the 16-bit triangle index -- then the
// vertices as float4.  Returns false when a number does not fit 16 bits (the traversal then reads tri_leaf from memory).
static bool build_tri_packets(const FlatAccel& A, std::vector<uint32_t>& out, int& entries, int& verts) {
  entries = A.top_prim_begin;
  verts = 0;
  out.clear();
  if (entries <= 0) return false;
  if (size_t(entries) * 8 > 96 * 1024) return false;  // (the entries alone would not fit the LDS the variants leave: no point in building the table)
  struct Key {
    uint32_t a, b, c;
    bool operator<(const Key& o) const { return a != o.a ? a < o.a : b != o.b ? b < o.b : c < o.c; }
  };
  std::map<Key, uint32_t> index;
  std::vector<Key> table;
  std::vector<uint32_t> ent(size_t((entries + 1) & ~1) * 2, 0u);
  for (int i = 0; i < entries; i++) {
    uint32_t w[10];
    memcpy(w, &A.tri_leaf[size_t(i) * 12], sizeof w);
    uint32_t id[3];
    for (int k = 0; k < 3; k++) {
      const Key key{w[3 * k], w[3 * k + 1], w[3 * k + 2]};
      auto it = index.find(key);
      if (it == index.end()) {
        if (table.size() >= 65536) return false;
        it = index.emplace(key, uint32_t(table.size())).first;
        table.push_back(key);
      }
      id[k] = it->second;
    }
    if (w[9] >= 65536u) return false;  // (the triangle index, as tri_leaf stores it)
    ent[size_t(i) * 2] = id[0] | (id[1] << 16);
    ent[size_t(i) * 2 + 1] = id[2] | (w[9] << 16);
  }
  verts = int(table.size());
  out = ent;
  for (const Key& k : table) {
    out.push_back(k.a);
    out.push_back(k.b);
    out.push_back(k.c);
    out.push_back(0u);
  }
  return true;
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
  // HaltonSampler(n): n as given too (sampler.h:44-46).  Both run in the F_SOBOL kernel variants.
  const bool halton = prm->sampler == PINE_GPU_SAMPLER_HALTON;
  const bool sobol = prm->sampler == PINE_GPU_SAMPLER_SOBOL || halton;  // (the sampler is not BlueSampler)
  if (prm->sampler != PINE_GPU_SAMPLER_BLUE && !sobol) {
    set_error("unknown sampler kind");
    return -1;
  }
  const int spp = sobol ? prm->spp : effective_spp(prm->spp);
  if (spp <= 0) {
    set_error(halton ? "`HaltonSampler` should have positive samples per pixel"
                     : sobol ? "`SobolSampler` should have positive samples per pixel" : "samples per pixel must be positive");
    return -1;
  }
  if (sobol && spp > kMaxDeviceSpp) {  // (the packed path state holds 12 bits of sample index)
    set_error(halton ? "HaltonSampler on the device: at most 4096 samples per pixel" : "SobolSampler on the device: at most 4096 samples per pixel");
    return -1;
  }
  const bool spp_pow2 = (spp & (spp - 1)) == 0;  // (BlueSampler's effective spp always is; SobolSampler / HaltonSampler take any count)
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
  if (!H.accel.built) {
    const char* e = getenv("PINE_GPU_DEVICE_BVH");
    H.build_on_device = ((prm->flags & PINE_GPU_FLAG_DEVICE_BVH) || (e && atoi(e) != 0)) ? prm->device : -1;
    H.build_accel();
  }
  const auto t_build1 = std::chrono::steady_clock::now();
  p->accel_build_ms = std::chrono::duration<float, std::milli>(t_build1 - t_build0).count();
  p->accel_on_device = H.built_on_device;
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
  // PINE_GPU_FLAG_ORDER_EMBREE: the hierarchy EmbreeAccel walks over the non-mesh shapes (pine_embree_order.h), the meshes' places
  // in `leaf` (tested first), and -- behind the part of the blob that scene-in-LDS variants copy -- the RCPPS estimates
  const bool order_embree = (prm->flags & PINE_GPU_FLAG_ORDER_EMBREE) != 0;
  S.off_etree = S.off_emesh = S.off_rcpps = 0;
  S.etree_root = kEmbreeNoChild;
  S.num_emesh = 0;
  if (order_embree) {
    if (prm->flags & (PINE_GPU_FLAG_FAST | PINE_GPU_FLAG_VERTEX_LOG)) {
      set_error("PINE_GPU_FLAG_ORDER_EMBREE cannot be combined with PINE_GPU_FLAG_FAST / _VERTEX_LOG");
      return -1;
    }
    const int ntop = int(A.top_boxes.size() / 8);
    std::vector<float> boxes;
    std::vector<int> places, mesh_places;
    for (int t = 0; t < ntop; t++) {  // (FlatAccel::top_boxes: the meshes first, then the other shapes, each in geometry order)
      const float* r = &A.top_boxes[size_t(t) * 8];
      int geom, place = -1;
      memcpy(&geom, &r[3], 4);
      for (size_t i = size_t(A.top_prim_begin); i < A.prims.size(); i++)
        if (A.prims[i] == geom) place = int(i);
      if (place < 0) {
        set_error("internal: a top-level primitive without a leaf entry");
        return -1;
      }
      if (shapes[size_t(geom)].kind == SHAPE_MESH) {
        mesh_places.push_back(place);
      } else {
        boxes.insert(boxes.end(), {r[0], r[1], r[2], r[4], r[5], r[6]});
        places.push_back(place);
      }
    }
    EmbreeOrderTree tree;
    std::string why;
    if (!tree.build(boxes, places, why)) {
      set_error("PINE_GPU_FLAG_ORDER_EMBREE: " + why);
      return -1;
    }
    S.etree_root = tree.root;
    S.off_etree = put(tree.nodes.data(), tree.nodes.size() * sizeof(EmbreeNode));
    S.num_emesh = int(mesh_places.size());
    S.off_emesh = put(mesh_places.data(), mesh_places.size() * sizeof(int));
  }
  // (the RCPPS estimates ride in the part of the blob that scene-in-LDS variants stage when the scene is small enough to stay one
  //  of theirs with them -- three lookups per ray from LDS instead of L2 -- and behind it, in global memory only, otherwise)
  const bool table_in_lds = order_embree && blob.size() + sizeof(kRcppsTable) + 64 <= 32 * 1024;
  if (table_in_lds) S.off_rcpps = put(kRcppsTable, sizeof(kRcppsTable));
  blob.resize((blob.size() + 15) & ~size_t(15));
  S.blob_bytes = int(blob.size());
  if (order_embree && !table_in_lds) S.off_rcpps = put(kRcppsTable, sizeof(kRcppsTable));
  HIP_OK(POOL_ALLOC(p->d_blob, blob.size()));
  HIP_OK(hipMemcpy(p->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
  if (upload(p->d_tri, A.tri_verts)) return -1;
  if (upload(p->d_tri_leaf, A.tri_leaf)) return -1;
  std::vector<uint32_t> tri_packets;
  int tri_packet_entries = 0, tri_packet_verts = 0;
  const bool have_tri_packets = build_tri_packets(A, tri_packets, tri_packet_entries, tri_packet_verts);
  if (have_tri_packets) {
    HIP_OK(POOL_ALLOC(p->d_tri_packets, tri_packets.size() * 4));
    HIP_OK(hipMemcpy(p->d_tri_packets, tri_packets.data(), tri_packets.size() * 4, hipMemcpyHostToDevice));
  }
  const size_t tri_packet_bytes = tri_packets.size() * 4;
  if (upload(p->d_tri_attrs, A.tri_attrs)) return -1;
  // tables: sobol + the selected spp variant
  int k = 0;
  while ((1 << k) < spp) k++;
  if (sobol) k = 0;  // (SobolSampler reads no table; any variant keeps the BlueSampler window loads in bounds)
  // device layout: sobolT 64 KiB | scramble 128 KiB | rank 128 KiB | 64 bytes = rank[0..63] again, so
  // a pixel's 40 consecutive ranking bytes never need the reference's modulo wrap
  HIP_OK(POOL_ALLOC(p->d_tables, 65536 + 262144 + 64));
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
  S.tri_attrs = p->d_tri_attrs;
  S.lds_nodes = 0;
  S.tri_packets = p->d_tri_packets;
  S.tri_packet_entries = tri_packet_entries;
  S.tri_packet_verts = tri_packet_verts;
  S.lds_tris = 0;
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
  S.tables.kind = halton ? 2 : sobol ? 1 : 0;
  S.tables.halton_primes = nullptr;
  S.tables.halton_perms = nullptr;
  if (halton) {
    const HaltonHostTables& ht = halton_host_tables();
    const size_t head = size_t(2 * kHaltonDims) * sizeof(int), bytes = head + ht.perms.size() * sizeof(uint16_t);
    HIP_OK(POOL_ALLOC(p->d_halton, bytes));
    HIP_OK(hipMemcpy(p->d_halton, ht.primes_and_sums.data(), head, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(p->d_halton + head, ht.perms.data(), ht.perms.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    S.tables.halton_primes = reinterpret_cast<const int*>(p->d_halton);
    S.tables.halton_perms = reinterpret_cast<const uint16_t*>(p->d_halton + head);
  }
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
  if (order_embree) need |= F_EMBREE;
  // (SobolSampler / HaltonSampler with Subsurface: a BSSRDF walk draws from the sampler at every step and the sampler's dimension
  //  counter outgrows the packed path state -- the F_SSS | F_SOBOL variants keep it in a word of its own: kBigDim)
  for (auto& L : light_list)
    if (L.kind != LIGHT_AREA) need |= F_LIGHTS;
  const bool lds_ok = size_t(S.blob_bytes) <= 32 * 1024 && getenv("PINE_GPU_NO_LDS_SCENE") == nullptr;
  p->variant = -1;
  for (int v = 0; v < kNumVariants; v++) {
    const unsigned F = kVariants[v].features;
    if ((F & need) != need) continue;
    if (((F & F_LDS_SCENE) != 0) != lds_ok) continue;
    if (((F & F_EMBREE) != 0) != order_embree) continue;  // (the order mode's twin variants: never without the flag)
    if (getenv("PINE_GPU_WPS") && atoi(getenv("PINE_GPU_WPS")) != kVariants[v].waves_per_simd) continue;
    p->variant = v;
    break;
  }
  if (p->variant >= 0 && (kVariants[p->variant].features & F_LDS_SCENE)) p->lds_bytes += size_t(S.blob_bytes);
  // The stage-queued kernel is the default whenever a variant covers the scene and its LDS fits;
  // PINE_GPU_KERNEL=mega forces the lane-owns-a-path kernel, which covers every scene.
  // LDS left after a variant's fixed parts goes to the BVH node cache first, then -- when ALL nodes are in and there is
  // still room -- to the triangle packets.  Measured on the icosphere scene (profiles/HISTORY.md 6.3): packets in place of the 288
  // deepest nodes change nothing (191.7 vs 190.1 ms), so nodes are never evicted for them.
  // PINE_GPU_LDS_TRIS=0 / 1: never / whenever the packets fit, before the nodes (measurement aid).
  auto lds_tris_fit = [&](unsigned F, size_t lds) -> bool {
    if (!(F & F_LDS_TOP) || !(F & F_MESH) || !have_tri_packets || lds >= 160 * 1024) return false;
    const size_t room = 160 * 1024 - lds;
    const char* e = getenv("PINE_GPU_LDS_TRIS");
    if (e && atoi(e) == 0) return false;
    if (e && atoi(e) == 1) return tri_packet_bytes <= room;
    return tri_packet_bytes + A.nodes.size() * sizeof(DNode) <= room;
  };
  p->queue_variant = -1;
  {
    const char* ksel = getenv("PINE_GPU_KERNEL");
    const bool want_queue = !(ksel && std::string(ksel) == "mega");
    if (want_queue) {
      const char* no_top = getenv("PINE_GPU_NO_LDS_TOP");  // (measurement aid: keep every node in global memory)
      for (int v = 0; v < kNumQueueVariants; v++) {
        const unsigned F = kQueueVariants[v].features;
        if ((F & need) != need) continue;
        if (((F & F_VLOG) != 0) != ((prm->flags & PINE_GPU_FLAG_VERTEX_LOG) != 0)) continue;  // (the test hook's twin variants)
        if (((F & F_EMBREE) != 0) != order_embree) continue;                                   // (the order mode's twin variants)
        if ((F & F_LDS_SCENE) && !lds_ok) continue;  // (a scene-in-global variant later in the table is the fallback when LDS is short)
        if ((F & F_LDS_TOP) && A.nodes.size() > 65535) continue;  // 16-bit stack entries
        const size_t rest_bytes = size_t(S.blob_bytes - S.off_shapes);
        if ((F & F_LDS_REST) && rest_bytes > 12 * 1024) continue;
        const size_t stack_bytes = std::max(kQueueVariants[v].min_stack,
                                            size_t(S.stack_total) * kQBlock * ((F & F_LDS_TOP) ? sizeof(unsigned short) : sizeof(int)));
        const size_t lds = kQueueVariants[v].fixed_lds + stack_bytes + ((F & F_LDS_SCENE) ? size_t(S.blob_bytes) : 0) +
                           ((F & F_LDS_REST) ? rest_bytes : 0);
        if (lds > 160 * 1024) continue;
        // (measurement aid: PINE_GPU_XSTAGE=1 also passes over the stage-less F_LDS_TOP variants, so that the scene lands on a
        //  traversal-stage variant whose layout its own kernel -- exact feature set, pine_specialize.h -- then inherits)
        if ((F & F_LDS_TOP) && !(F & F_XSTAGE) && getenv("PINE_GPU_XSTAGE") && atoi(getenv("PINE_GPU_XSTAGE")) == 1) continue;
        if (F & F_XSTAGE) {
          // traversal stages only where refilling pays: a scene with meshes, (nearly) all nodes in this variant's LDS.
          // PINE_GPU_XSTAGE=0 / 1: never / always (measurement aid, tools/xstage_ab.py).
          const size_t cached = no_top ? 0 : std::min<size_t>(A.nodes.size(), (160 * 1024 - lds) / sizeof(DNode));
          bool want = (need & F_MESH) != 0 && cached * 10 >= A.nodes.size() * 9;
          if (const char* e = getenv("PINE_GPU_XSTAGE")) want = atoi(e) != 0;
          if (!want) continue;  // (the same feature set without F_XSTAGE follows in the table)
        }
        p->queue_variant = v;
        p->lds_bytes = lds;
        S.lds_nodes = 0;
        S.lds_tris = lds_tris_fit(F, lds) ? 1 : 0;
        if (S.lds_tris) p->lds_bytes += tri_packet_bytes;
        if (F & F_LDS_TOP) {
          S.lds_nodes = int(std::min<size_t>(A.nodes.size(), (160 * 1024 - p->lds_bytes) / sizeof(DNode)));
          if (no_top) S.lds_nodes = 0;
          p->lds_bytes += size_t(S.lds_nodes) * sizeof(DNode);
        }
        break;
      }
    }
  }

  if (plan_specialize(p, A, shapes, packed_prims, prm, need)) return -1;

  if (p->variant < 0 && p->queue_variant < 0) {  // (only experiment builds lack the all-features megakernel)
    set_error("no kernel variant covers this scene");
    return -1;
  }
  if (prm->flags & PINE_GPU_FLAG_FAST) {
    // declared-tolerance arithmetic: one of the few variants pine_kernels_fast.hip compiles, chosen by the same rules
    int nf = 0;
    const PineFastVariant* fv = pine_gpu_fast_variants(&nf);
    p->fast = nullptr;
    for (int v = 0; v < nf && !p->fast; v++) {
      const unsigned F = fv[v].features;
      if ((F & need) != need) continue;
      if ((F & F_LDS_SCENE) && !lds_ok) continue;
      if ((F & F_LDS_TOP) && A.nodes.size() > 65535) continue;
      const size_t rest_bytes = size_t(S.blob_bytes - S.off_shapes);
      if ((F & F_LDS_REST) && rest_bytes > 12 * 1024) continue;
      const size_t stack_bytes = std::max(fv[v].min_stack, size_t(S.stack_total) * kQBlock * ((F & F_LDS_TOP) ? sizeof(unsigned short) : sizeof(int)));
      const size_t lds = fv[v].fixed_lds + stack_bytes + ((F & F_LDS_SCENE) ? size_t(S.blob_bytes) : 0) + ((F & F_LDS_REST) ? rest_bytes : 0);
      if (lds > 160 * 1024) continue;
      p->fast = &fv[v];
      p->lds_bytes = lds;
      S.lds_nodes = 0;
      S.lds_tris = lds_tris_fit(F, lds) ? 1 : 0;
      if (S.lds_tris) p->lds_bytes += tri_packet_bytes;
      if (F & F_LDS_TOP) {
        S.lds_nodes = int(std::min<size_t>(A.nodes.size(), (160 * 1024 - p->lds_bytes) / sizeof(DNode)));
        p->lds_bytes += size_t(S.lds_nodes) * sizeof(DNode);
      }
    }
    if (!p->fast) {
      set_error("PINE_GPU_FLAG_FAST: no declared-tolerance kernel variant covers this scene (exact mode renders it)");
      return -1;
    }
    p->queue_variant = -1;
  }

  // scenes whose materials draw from the per-pixel RNG inside radiance() (Uber with fractional
  // metallic/transmission: sampler.h:317-324; BSSRDF channel pick: bxdf.cpp:335) make a pixel's
  // samples sequentially dependent: one item = the whole pixel.
  bool in_path_rng = false;
  bool uber_rng = false;  // ... at any Uber vertex of a path, whatever came before it
  for (auto& m : dev_materials) {
    if (m.kind == MAT_SUBSURFACE) in_path_rng = true;
    if (m.kind == MAT_UBER && (m.prog[2] >= 0 || m.prog[3] >= 0)) uber_rng = true;  // value known only at the surface
    if (m.kind == MAT_UBER) {
      if (m.metallic != 0 && m.metallic != 1) uber_rng = true;
      if (m.metallic != 1 && m.transmission != 0 && m.transmission != 1) uber_rng = true;
    }
  }
  in_path_rng |= uber_rng;
  p->serial_rng = in_path_rng;
  int kspi = prm->samples_per_item;
  if (in_path_rng) kspi = spp;
  else if (kspi <= 0) {
    // stage-queued kernel: two samples per item for the scene-in-LDS variants (cbox-class scenes, all pixels alike: half the
    // checkpoint prepass and hand-outs, C2 13.93 -> 13.75 ms per step, C3 69.9 -> 68.4), one where pixels differ a lot
    // (10 000 cones: 8.06 ms at one, 8.52 at two); the megakernel four
    const unsigned qf = p->fast ? p->fast->features : p->queue_variant >= 0 ? kQueueVariants[p->queue_variant].features : 0u;
    kspi = (p->queue_variant >= 0 || p->fast) ? ((qf & F_LDS_SCENE) ? std::min(spp, 2) : 1) : std::min(spp, 4);
  }
  if (kspi > spp) kspi = spp;
  if (!spp_pow2) {
    // SobolSampler(12): the work decomposition splits a pixel's samples by shifts and masks, so a count that is not a power
    // of two is ONE item per pixel -- all its samples in sequence, no checkpoints (sampler.cpp:81-113 takes any count)
    kspi = spp;
  } else {  // k must be a power of two dividing spp
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
  W.tile_order = nullptr;
  W.serial_tiles = 0;
  {
    // Tile classes (WorkParams::serial_tiles): in a scene whose only in-path RNG consumer is the BSSRDF channel pick and
    // whose other materials are Diffuse / Emissive, a path draws from the pixel's RNG only while it has met nothing but
    // Subsurface surfaces -- so a pixel none of whose camera rays can reach a Subsurface shape makes NO in-path draw, its
    // samples are independent (RNG state of sample s = the seed advanced 4 s steps, as in a scene without in-path draws)
    // and need not form a chain.  Conservative test per 8x8 tile: the world boxes of the Subsurface shapes projected
    // through the pinhole camera, two pixels of margin.  Everything else (thin lens, a box behind the camera, other
    // materials, the megakernel) keeps one whole-pixel item per pixel.  PINE_GPU_NO_TILE_CLASSES: off (measurement aid).
    const unsigned qf = p->queue_variant >= 0 ? kQueueVariants[p->queue_variant].features : p->fast ? p->fast->features : 0u;
    bool ok = in_path_rng && !uber_rng && (qf & F_SSS) != 0 && H.camera.len_radius == 0.0f && getenv("PINE_GPU_NO_TILE_CLASSES") == nullptr &&
              getenv("PINE_GPU_NO_FORK") == nullptr && spp > 1 && spp_pow2;  // (the independent class splits a pixel's samples by shifts and masks)
    for (auto& m : dev_materials)
      if (m.kind != MAT_EMISSIVE && m.kind != MAT_DIFFUSE && m.kind != MAT_SUBSURFACE) ok = false;
    if (ok) {
      // inverse of the camera's linear part (columns x, y, z of c2w), in double
      const float* c = H.camera.c2w;
      const double a[3][3] = {{c[0], c[3], c[6]}, {c[1], c[4], c[7]}, {c[2], c[5], c[8]}};  // a[row][col]
      const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                         a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
      if (!(std::fabs(det) > 1e-12) || !(H.camera.fov2d[0] > 0) || !(H.camera.fov2d[1] > 0)) ok = false;
      double inv[3][3];
      if (ok) {
        inv[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det, inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det, inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
        inv[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det, inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det, inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
        inv[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det, inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det, inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
      }
      // pixel rectangles [x0, x1] x [y0, y1] that the Subsurface shapes can project into
      struct Rect2 { double x0, y0, x1, y1; };
      std::vector<Rect2> rects;
      for (size_t g = 0; ok && g < shapes.size(); g++) {
        if (dev_materials[size_t(shapes[g].material)].kind != MAT_SUBSURFACE) continue;
        const HostAABB b = H.geometry_aabb(int(g));
        Rect2 r{1e300, 1e300, -1e300, -1e300};
        for (int corner = 0; corner < 8 && ok; corner++) {
          const double P[3] = {double((corner & 1) ? b.upper.x : b.lower.x) - H.camera.position[0], double((corner & 2) ? b.upper.y : b.lower.y) - H.camera.position[1],
                               double((corner & 4) ? b.upper.z : b.lower.z) - H.camera.position[2]};
          if (!(std::isfinite(P[0]) && std::isfinite(P[1]) && std::isfinite(P[2]))) { ok = false; break; }
          const double qx = inv[0][0] * P[0] + inv[0][1] * P[1] + inv[0][2] * P[2], qy = inv[1][0] * P[0] + inv[1][1] * P[1] + inv[1][2] * P[2],
                       qz = inv[2][0] * P[0] + inv[2][1] * P[1] + inv[2][2] * P[2];
          if (!(qz > 1e-4)) { ok = false; break; }  // a corner at or behind the camera plane: no bounded projection
          const double fx = ((qx / qz) / H.camera.fov2d[0] * 0.5 + 0.5) * H.camera.W, fy = ((qy / qz) / H.camera.fov2d[1] * 0.5 + 0.5) * H.camera.H;
          r.x0 = std::min(r.x0, fx), r.y0 = std::min(r.y0, fy), r.x1 = std::max(r.x1, fx), r.y1 = std::max(r.y1, fy);
        }
        // margin: the pixel's own extent (jitter in [0, 1)) + two pixels for every rounding on the way
        r.x0 -= 3.0, r.y0 -= 3.0, r.x1 += 2.0, r.y1 += 2.0;
        rects.push_back(r);
      }
      if (ok) {
        std::vector<int> serial, free_tiles;
        for (int lt = 0; lt < W.num_local_tiles; lt++) {
          const int tile = lt * W.shard_world + W.shard_rank;
          const int tx = tile % W.tiles_x, ty = tile / W.tiles_x;
          const double x0 = tx * kTile, y0 = ty * kTile, x1 = x0 + kTile, y1 = y0 + kTile;
          bool touched = false;
          for (const Rect2& r : rects)
            if (x0 <= r.x1 && x1 >= r.x0 && y0 <= r.y1 && y1 >= r.y0) touched = true;
          (touched ? serial : free_tiles).push_back(tile);
        }
        if (!free_tiles.empty() && !serial.empty()) {
          p->tile_order = serial;
          p->tile_order.insert(p->tile_order.end(), free_tiles.begin(), free_tiles.end());
          W.serial_tiles = int(serial.size());
          // the independent class: one sample per item, RNG checkpoints (the Subsurface variants are F_LDS_TOP ones)
          W.samples_per_item = 1;
          W.items_per_pixel = spp;
          W.log2_items_per_pixel = 0;
          while ((1 << W.log2_items_per_pixel) < W.items_per_pixel) W.log2_items_per_pixel++;
          W.total_items = (unsigned long long)W.serial_tiles * 64ull + (unsigned long long)(W.num_local_tiles - W.serial_tiles) * W.items_per_pixel * 64ull;
        } else if (serial.empty()) {
          // no camera ray can reach a Subsurface shape from this shard's tiles: nothing to do here, the scene stays
          // "one item per pixel" (rare, and a launch without chains would need the checkpoint prepass for every tile)
        }
      }
    }
  }
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
  // Traversal stages of the X variants (pine_queue_kernel.h): a wave goes to retire / refill its lanes when fewer than
  // trav_min_lanes of them are still travelling, at the earliest trav_min_trips trips after the last time.  Measured
  // (profiles/HISTORY.md 6.3): refilling pays when (nearly) the whole BVH sits in LDS (icosphere scene: 212 -> 185 ms); when most node
  // fetches go to L2 it costs -- the rays a wave picks up later are not the neighbours of the ones it has, and the
  // traversal waits on memory (10 000 cones: 9.8 ms without, 10.9 ms with) -- so there a wave runs its rays to the end.
  W.trav_min_lanes = (S.lds_nodes > 0 && size_t(S.lds_nodes) * 10 >= A.nodes.size() * 9) ? 48 : 0;  // (nearly) all nodes in LDS
  W.trav_min_trips = 8;
  // (a kernel with the top level baked in, pine_specialize.h: most rays end in the code part and free their lanes at once;
  //  refilling after three trips instead of eight: C5 107.3 -> 100.2 ms)
  if (!p->spec_request.baked.empty() && (p->spec_request.features & F_XSTAGE)) W.trav_min_trips = 3;
  if (const char* e = getenv("PINE_GPU_TRAV_MIN_LANES")) W.trav_min_lanes = atoi(e);
  if (const char* e = getenv("PINE_GPU_TRAV_MIN_TRIPS")) W.trav_min_trips = atoi(e) > 0 ? atoi(e) : 1;
  // Subsurface is the only in-path user of the RNG and only before a path's first non-delta bounce: such a path hands its
  // pixel's next sample on when it has made that bounce (pine_queue_kernel.h, "sample tokens")
  W.fork_sealed = (in_path_rng && !uber_rng && getenv("PINE_GPU_NO_FORK") == nullptr) ? 1 : 0;
  // ... and then a workgroup keeps at most 320 pixels in flight (each with one unsealed path; the other contexts trace the
  // sealed rest of earlier samples): when the work-item pool runs dry little is left half-done, so the workgroups end closer
  // together (C5: 178 -> 171 ms; 192 ... 384 within 1 %, 512 and more as without a limit)
  W.max_pixels = W.fork_sealed ? 320 : (1 << 20);
  if (const char* e = getenv("PINE_GPU_MAX_PIXELS")) W.max_pixels = atoi(e) > 0 ? atoi(e) : W.max_pixels;
  // A workgroup claims 512 items at a time; when an item is a pixel's whole sample sequence (serial-RNG scenes) that is
  // tens of milliseconds of its time, and the last claims decide when the launch ends: one 8x8 tile at a time there.
  W.pick_spins = 8;
  W.fair_period = 8;
  if (const char* e = getenv("PINE_GPU_FAIR_PERIOD")) W.fair_period = atoi(e);
  if (const char* e = getenv("PINE_GPU_PICK_SPINS")) W.pick_spins = atoi(e) > 0 ? atoi(e) : 1;
  W.pool_items = in_path_rng ? 64 : 512;
  if (const char* e = getenv("PINE_GPU_POOL_ITEMS")) W.pool_items = atoi(e) > 0 ? atoi(e) : W.pool_items;
  if (W.serial_tiles > 0) W.pool_items = 64;  // tile classes: a claim never straddles the boundary between the classes (a multiple of 64)
  W.progress = nullptr;
  W.vertex_log = nullptr;
  if (prm->flags & PINE_GPU_FLAG_PROGRESS) {
    HIP_OK(hipHostMalloc((void**)&p->h_progress, sizeof(unsigned long long), hipHostMallocMapped));
    *p->h_progress = 0;
    HIP_OK(hipHostGetDevicePointer((void**)&W.progress, p->h_progress, 0));
  }

  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, prm->device));
  int blocks_per_cu = 0;
  const bool queued = p->queue_variant >= 0 || p->fast;  // the stage-queued kernel (exact or declared-tolerance build)
  const void* const queue_fn = p->fast ? p->fast->fn : p->queue_variant >= 0 ? (const void*)kQueueVariants[p->queue_variant].fn : nullptr;
  const unsigned queue_features = p->fast ? p->fast->features : p->queue_variant >= 0 ? kQueueVariants[p->queue_variant].features : 0u;
  if (queued) {
    blocks_per_cu = 1;  // one 1024-thread workgroup per CU owns the CU's LDS
    HIP_OK(hipFuncSetAttribute(queue_fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(p->lds_bytes)));
  } else {
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, (const void*)kVariants[p->variant].fn, kBlock, p->lds_bytes));
  }
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  if (blocks_per_cu > 8) blocks_per_cu = 8;
  const char* env_bpc = getenv("PINE_GPU_BLOCKS_PER_CU");
  if (env_bpc && atoi(env_bpc) > 0) blocks_per_cu = atoi(env_bpc);
  unsigned long long want = (W.total_items + kBlock - 1) / kBlock;
  const int qctx = p->fast ? p->fast->ctx : p->queue_variant >= 0 ? kQueueVariants[p->queue_variant].ctx : 0;
  if (queued) want = (W.total_items + qctx - 1) / qctx;
  p->grid = int(std::min<unsigned long long>(want, (unsigned long long)prop.multiProcessorCount * blocks_per_cu));
  if (p->grid < 1) p->grid = 1;
  if (!in_path_rng && getenv("PINE_GPU_POOL_ITEMS") == nullptr) {
    // work-item claims of the stage-queued kernel: 1/32 of a workgroup's share, between 512 and 2048 (a claim is a run of
    // neighbouring tiles: larger ones keep a workgroup's camera rays together and are fewer -- 10 000 cones 8.00 -> 7.88 ms
    // at 2048, 7.81 at 4096, 8.4 at 8192 where the last claims unbalance the end; cbox indifferent up to 2048)
    unsigned long long share = W.total_items / ((unsigned long long)p->grid * 32ull);
    int claim = 512;
    while (claim < 2048 && (unsigned long long)claim * 2ull <= share) claim *= 2;
    W.pool_items = claim;
  }

  if (W.items_per_pixel > 1)
    HIP_OK(POOL_ALLOC(p->d_ckpt, (size_t)(W.num_local_tiles - W.serial_tiles) * W.items_per_pixel * 64 * sizeof(ulonglong2)));
    p->ckpt_every_launch = getenv("PINE_GPU_CKPT_EVERY_LAUNCH") && atoi(getenv("PINE_GPU_CKPT_EVERY_LAUNCH")) != 0;
  if (!p->tile_order.empty()) {
    HIP_OK(POOL_ALLOC(p->d_tile_order, p->tile_order.size() * sizeof(int)));
    HIP_OK(hipMemcpy(p->d_tile_order, p->tile_order.data(), p->tile_order.size() * sizeof(int), hipMemcpyHostToDevice));
    W.tile_order = p->d_tile_order;
  }
  HIP_OK(POOL_ALLOC(p->d_samples, (size_t)W.num_local_tiles * spp * 64 * sizeof(float4)));
  const size_t fold_slots = queued ? size_t(p->grid) * qctx : size_t(p->grid) * kBlock;
  HIP_OK(POOL_ALLOC(p->d_fold, size_t(prm->max_path_length) * 8 * fold_slots * sizeof(float)));
  if (queued) {
    // per-context records, then (Subsurface variants) every workgroup's ring of sample-token slots
    const size_t token_dwords = (queue_features & F_SSS) ? size_t(p->grid) * (qctx <= 1024 ? 1024 : 2048) * kQTokenDwords : 0;
    HIP_OK(POOL_ALLOC(p->d_ctxg, (size_t(p->grid) * qctx * q_ctx_global_dwords(queue_features) + token_dwords) * sizeof(uint32_t)));
  }
  HIP_OK(POOL_ALLOC(p->d_counters, sizeof(Counters)));
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
  // a background build that has finished: this launch and every later one run the scene's own kernel.  (A code object the
  // runtime refuses leaves the precompiled kernel in place -- same film; plan stats say which one runs.)
  plan_poll_background(p);
  g_progress.store(0.0f);
  const size_t film_bytes = size_t(p->film_w) * p->film_h * sizeof(float4);
  if (p->W.shard_world > 1 && !packed) HIP_OK(hipMemsetAsync(film_dev, 0, film_bytes, stream));
  HIP_OK(hipMemsetAsync(p->d_counters, 0, sizeof(Counters), stream));
  hipEvent_t* ev = p->ev[p->launch_count % pine_gpu_plan::kEvRing];
  if (p->timed) HIP_OK(hipEventRecord(ev[0], stream));
  // (a shard can own no tile at all -- more ranks than 8x8 tiles: nothing to launch, the film / slab stays zero)
  const bool has_work = p->W.num_local_tiles > 0;
  if (has_work && p->W.items_per_pixel > 1) {
    if (!p->ckpt_valid || p->ckpt_every_launch) {
      const unsigned long long n = (unsigned long long)(p->W.num_local_tiles - p->W.serial_tiles) * 64ull;
      hipLaunchKernelGGL(rng_checkpoint_kernel, dim3(unsigned((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                         p->W, p->film_w, p->film_h, p->S.spp, p->d_ckpt);
      if (!p->ckpt_done) HIP_OK(hipEventCreateWithFlags(&p->ckpt_done, hipEventDisableTiming));
      HIP_OK(hipEventRecord(p->ckpt_done, stream));
      p->ckpt_stream = stream;
      p->ckpt_valid = true;
    } else if (stream != p->ckpt_stream) {
      HIP_OK(hipStreamWaitEvent(stream, p->ckpt_done, 0));
    }
  }
  if (p->timed) HIP_OK(hipEventRecord(ev[1], stream));
  if (!has_work) {
  } else if (p->fast) {
    // (a kernel of the other translation unit: same argument layout, launched untyped)
    const ulonglong2* ckpt = p->d_ckpt;
    void* args[] = {&p->S, &p->W, &ckpt, &p->d_samples, &p->d_fold, &p->d_ctxg, &p->d_counters};
    HIP_OK(hipLaunchKernel(p->fast->fn, dim3(p->grid), dim3(kQBlock), args, p->lds_bytes, stream));
  } else if (p->spec_fn) {
    const ulonglong2* ckpt = p->d_ckpt;
    void* args[] = {&p->S, &p->W, &ckpt, &p->d_samples, &p->d_fold, &p->d_ctxg, &p->d_counters};
    HIP_OK(hipModuleLaunchKernel(p->spec_fn, unsigned(p->grid), 1, 1, kQBlock, 1, 1, unsigned(p->lds_bytes), stream, args, nullptr));
  } else if (p->queue_variant >= 0) {
    const ulonglong2* ckpt = p->d_ckpt;
    void* args[] = {&p->S, &p->W, &ckpt, &p->d_samples, &p->d_fold, &p->d_ctxg, &p->d_counters};
    HIP_OK(hipLaunchKernel(kQueueVariants[p->queue_variant].fn, dim3(p->grid), dim3(kQBlock), args, p->lds_bytes, stream));
  } else {
    const ulonglong2* ckpt = p->d_ckpt;
    void* args[] = {&p->S, &p->W, &ckpt, &p->d_samples, &p->d_fold, &p->d_counters};
    HIP_OK(hipLaunchKernel(kVariants[p->variant].fn, dim3(p->grid), dim3(kBlock), args, p->lds_bytes, stream));
  }
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

void pine_gpu_release_cached_memory(void) {
  DevicePool::get().release_all();
  std::lock_guard<std::mutex> lock(LoadedKernels::get().mu);
  LoadedKernels::get().recent.clear();  // (a module is unloaded when the last plan that launches it is gone)
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
      int tile = p->tile_order.empty() ? lt * p->W.shard_world + p->W.shard_rank : p->tile_order[size_t(lt)];
      int tx = tile % p->W.tiles_x, ty = tile / p->W.tiles_x;
      int w = std::min(kTile, p->film_w - tx * kTile), h = std::min(kTile, p->film_h - ty * kTile);
      px += (unsigned long long)w * h;
    }
    out->camera_samples = px * p->S.spp;
  }
  out->spp_effective = p->S.spp;
  out->samples_per_item = p->W.samples_per_item;
  out->serial_tiles = p->W.serial_tiles;
  plan_poll_background(p);
  out->specialized = p->spec_fn ? (p->spec_baked ? 2 : 1) : p->spec_state.load() == kSpecFailed ? -1 : 0;
  out->specialize_source = p->spec_fn || p->spec_state.load() == kSpecBuilding ? p->spec_source : 0;
  out->specialize_pending = p->spec_state.load() == kSpecBuilding ? 1 : 0;
  out->kernel_features = p->spec_fn ? p->spec_features : p->fast ? p->fast->features : p->queue_variant >= 0 ? kQueueVariants[p->queue_variant].features
                                                                                         : p->variant >= 0 ? kVariants[p->variant].features : 0u;
  out->specialize_ms = p->specialize_ms;
  out->grid_blocks = p->grid;
  out->block_threads = (p->queue_variant >= 0 || p->fast) ? kQBlock : kBlock;
  out->lds_bytes = int(p->lds_bytes);
  out->accel_build_ms = p->accel_build_ms;
  out->accel_built_on_device = p->accel_on_device ? 1 : 0;
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
  if (c.t_end > c.t_start && c.t_start)
    fprintf(stderr, "timeline: kernel %.2f ms, work-item pool dry after %.2f ms\n", double(c.t_end - c.t_start) * 1e-5,
            c.t_pool_dry ? double(c.t_pool_dry - c.t_start) * 1e-5 : -1.0);
  if (getenv("PINE_GPU_WG_TIMELINE") && c.t_start) {
    // per workgroup, ms after the launch's start: last whole-pixel item sealed | pool found dry | out | whole-pixel items it claimed
    for (int b = 0; b < p->grid && b < 1024; b++)
      fprintf(stderr, "wg %3d: whole-pixel items done %7.2f  pool dry %7.2f  out %7.2f  pixels %llu\n", b,
              c.wg_t[b][0] ? double(c.wg_t[b][0] - c.t_start) * 1e-5 : -1.0, c.wg_t[b][1] ? double(c.wg_t[b][1] - c.t_start) * 1e-5 : -1.0,
              c.wg_t[b][2] ? double(c.wg_t[b][2] - c.t_start) * 1e-5 : -1.0, c.wg_t[b][3]);
  }
  unsigned long long rl[16] = {0}, rh[16] = {0};
  {
    using RegFn = int (*)(unsigned long long*, unsigned long long*);
    static const RegFn regs[kPineKernelParts] = {pine_gpu_kernel_part_regions_0, pine_gpu_kernel_part_regions_1, pine_gpu_kernel_part_regions_2,
                                                 pine_gpu_kernel_part_regions_3, pine_gpu_kernel_part_regions_4, pine_gpu_kernel_part_regions_5,
                                                 pine_gpu_kernel_part_regions_6, pine_gpu_kernel_part_regions_7};
    for (RegFn f : regs)
      if (f(rl, rh)) return -1;
  }
  if (p->spec_module) {
    // the scene's own kernel (a module of its own) keeps its own copies of the REGION counters
    hipDeviceptr_t dl = nullptr, dh = nullptr;
    size_t bl = 0, bh = 0;
    if (hipModuleGetGlobal(&dl, &bl, p->spec_module, "_ZN8pine_gpuL14g_region_lanesE") == hipSuccess &&
        hipModuleGetGlobal(&dh, &bh, p->spec_module, "_ZN8pine_gpuL13g_region_hitsE") == hipSuccess && bl == sizeof rl && bh == sizeof rh) {
      unsigned long long ml[16], mh[16];
      if (hipMemcpy(ml, dl, sizeof ml, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(mh, dh, sizeof mh, hipMemcpyDeviceToHost) == hipSuccess)
        for (int i = 0; i < 16; i++) rl[i] += ml[i], rh[i] += mh[i];
    } else {
      (void)hipGetLastError();
    }
  }
  for (int i = 0; i < 16; i++)
    if (rh[i]) fprintf(stderr, "region %2d: entries %llu avg active lanes %.2f\n", i, rh[i], double(rl[i]) / double(rh[i]));
#endif
  return 0;
}

int pine_gpu_plan_test_traverse_baked(pine_gpu_plan* p, const float* rays, int64_t nrays, uint32_t* out) {
  if (!p || !rays || !out || nrays <= 0) {
    set_error("null argument");
    return -1;
  }
  if (!p->spec_module || !p->spec_baked) {
    set_error("the plan has no baked scene (PINE_GPU_FLAG_SPECIALIZE, a scene that qualifies)");
    return -1;
  }
  HIP_OK(hipSetDevice(p->device));
  hipFunction_t fn;
  HIP_OK(hipModuleGetFunction(&fn, p->spec_module, "pine_baked_traverse_test"));
  float* dr = nullptr;
  unsigned* dout = nullptr;
  int rc = -1;
  do {
    if (hipMalloc((void**)&dr, size_t(nrays) * 32) != hipSuccess || hipMalloc((void**)&dout, size_t(nrays) * 16) != hipSuccess) break;
    if (hipMemcpy(dr, rays, size_t(nrays) * 32, hipMemcpyHostToDevice) != hipSuccess) break;
    long long n = nrays;
    void* args[] = {&dr, &n, &dout};
    if (hipModuleLaunchKernel(fn, unsigned((nrays + 63) / 64), 1, 1, 64, 1, 1, 0, nullptr, args, nullptr) != hipSuccess) break;
    if (hipMemcpy(out, dout, size_t(nrays) * 16, hipMemcpyDeviceToHost) != hipSuccess) break;
    rc = 0;
  } while (0);
  if (rc) set_error(std::string("pine_gpu_plan_test_traverse_baked: ") + hipGetErrorString(hipGetLastError()));
  (void)hipFree(dr);
  (void)hipFree(dout);
  return rc;
}

int64_t pine_gpu_plan_vertex_log(pine_gpu_plan* p, float* out, int64_t capacity) {
  if (!p) {
    set_error("null argument");
    return -1;
  }
  if (p->queue_variant < 0 || !(kQueueVariants[p->queue_variant].features & F_VLOG)) {
    set_error("the per-vertex log needs a plan created with PINE_GPU_FLAG_VERTEX_LOG (stage-queued kernel, the two variants compiled with the hook)");
    return -1;
  }
  const int64_t n = int64_t(p->film_w) * p->film_h * p->S.spp * p->S.max_path_length * kVertexLogFloats;
  if (n > (int64_t(1) << 27)) {
    set_error("the per-vertex log is meant for small films (at most 2^27 floats)");
    return -1;
  }
  HIP_OK(hipSetDevice(p->device));
  if (!out) {  // switch the log on: the NEXT launches fill it
    if (!p->d_vertex_log) HIP_OK(POOL_ALLOC(p->d_vertex_log, size_t(n) * 4));
    HIP_OK(hipMemset(p->d_vertex_log, 0, size_t(n) * 4));
    p->W.vertex_log = p->d_vertex_log;
    return n;
  }
  if (!p->d_vertex_log || capacity < n) {
    set_error("vertex log not enabled, or capacity too small");
    return -1;
  }
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(out, p->d_vertex_log, size_t(n) * 4, hipMemcpyDeviceToHost));
  return n;
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
    int tile = p->tile_order.empty() ? lt * p->W.shard_world + p->W.shard_rank : p->tile_order[size_t(lt)];
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
    if (DevicePool::get().alloc(&d_film, bytes) != hipSuccess) {
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
  if (rc) (void)hipDeviceSynchronize();  // (a failed launch may still be running: nothing of it may touch a block the pool hands out again)
  DevicePool::get().free(d_film);
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

/* Test hook (host only): the reference's Lomuto partition as a sequential swap loop (perm_seq) and as the prefix-sum +
 * pointer-jumping formulation the device build uses (perm_par); returns the number of trues, < 0 if the two disagree. */
int pine_gpu_test_lomuto(const unsigned char* pred, int n, int* perm_seq, int* perm_par) {
  if (!pred || !perm_seq || !perm_par || n < 0) {
    set_error("bad argument");
    return -1;
  }
  for (int i = 0; i < n; i++) perm_seq[i] = i;
  const int a = build_lomuto(pred, perm_seq, n);
  const int b = build_lomuto_by_chains(pred, perm_par, n);
  if (a != b || memcmp(perm_seq, perm_par, size_t(n) * sizeof(int)) != 0) {
    set_error("the two partition formulations disagree");
    return -2;
  }
  return a;
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
int pine_gpu_test_atan(int device, const float* y, const float* x, int64_t n, float* at2, float* ac) {
  if (need_device(device)) return -1;
  float *dy, *dx, *da, *dc;
  HIP_OK(hipMalloc((void**)&dy, n * 4));
  HIP_OK(hipMalloc((void**)&dx, n * 4));
  HIP_OK(hipMalloc((void**)&da, n * 4));
  HIP_OK(hipMalloc((void**)&dc, n * 4));
  HIP_OK(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(test_atan_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, dy, dx, (long long)n, da, dc);
  HIP_OK(hipMemcpy(at2, da, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(ac, dc, n * 4, hipMemcpyDeviceToHost));
  hipFree(dy);
  hipFree(dx);
  hipFree(da);
  hipFree(dc);
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
int pine_gpu_test_embree_tree(const float* boxes, int n, int* words, int cap) {
  if (!boxes || !words || n < 0) {
    set_error("bad argument");
    return -1;
  }
  std::vector<float> bx(boxes, boxes + 6 * size_t(n));
  std::vector<int> places(size_t(n), 0);
  for (int i = 0; i < n; i++) places[size_t(i)] = i;
  EmbreeOrderTree tree;
  std::string why;
  if (!tree.build(bx, places, why)) {
    set_error(why);
    return -1;
  }
  if (1 + 8 * int(tree.nodes.size()) > cap) {
    set_error("capacity");
    return -1;
  }
  int k = 0;
  words[k++] = tree.root;
  for (const EmbreeNode& nd : tree.nodes)
    for (int i = 0; i < 8; i++) words[k++] = nd.child[i];
  return k;
}
int pine_gpu_test_traverse(pine_gpu_scene* scene, int device, const float* rays, int64_t nrays, int flat, int cap, uint32_t* out) {
  if (!scene || !rays || !out || cap < 2 || nrays < 0) {
    set_error("bad argument");
    return -1;
  }
  if (need_device(device)) return -1;
  // the scene as the kernels see it: a plan's device records (nothing is rendered)
  pine_gpu_render_params prm{};
  prm.spp = 1, prm.max_path_length = 2, prm.device = device, prm.shard_rank = 0, prm.shard_world = 1;
  prm.flags = PINE_GPU_FLAG_NO_SPECIALIZE | (flat == 2 ? PINE_GPU_FLAG_ORDER_EMBREE : 0);
  pine_gpu_plan* p = pine_gpu_plan_create(scene, &prm);
  if (!p) return -1;
  int rc = -1;
  float* dr = nullptr;
  unsigned* dout = nullptr;
  const size_t words = size_t(nrays) * (2 * size_t(cap) + 5);
  do {
    if (flat == 1 && p->S.stack_total > 0 && scene_host(scene).accel.nodes.size() > 65535) {
      set_error("the flat traversal keeps 16-bit node ids");
      break;
    }
    if (hipMalloc((void**)&dr, std::max<int64_t>(nrays, 1) * 32) != hipSuccess || hipMalloc((void**)&dout, std::max<size_t>(words, 1) * 4) != hipSuccess) break;
    if (hipMemcpy(dr, rays, nrays * 32, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dout, 0, std::max<size_t>(words, 1) * 4) != hipSuccess) break;
    const size_t lds = size_t(std::max(1, p->S.stack_total)) * 64 * (flat == 1 ? sizeof(unsigned short) : sizeof(int));
    if (nrays > 0) {
      if (flat == 1) hipLaunchKernelGGL(test_traverse_kernel<1>, dim3(unsigned((nrays + 63) / 64)), dim3(64), lds, 0, p->S, dr, (long long)nrays, cap, dout);
      else if (flat == 2) hipLaunchKernelGGL(test_traverse_kernel<2>, dim3(unsigned((nrays + 63) / 64)), dim3(64), lds, 0, p->S, dr, (long long)nrays, cap, dout);
      else hipLaunchKernelGGL(test_traverse_kernel<0>, dim3(unsigned((nrays + 63) / 64)), dim3(64), lds, 0, p->S, dr, (long long)nrays, cap, dout);
    }
    if (hipMemcpy(out, dout, words * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
    rc = 0;
  } while (0);
  if (rc) set_error(std::string("pine_gpu_test_traverse: ") + hipGetErrorString(hipGetLastError()));
  (void)hipFree(dr);
  (void)hipFree(dout);
  pine_gpu_plan_destroy(p);
  return rc;
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
