// pine_amd/csrc/pine_variants.h -- the compiled specialisations of the two path kernels, and the plain-typed table
// through which the translation units that instantiate them hand them to the host code in pine_kernels.hip.
//
// One translation unit per PART (pine_kernels_part.hip, compiled once per part with -DPINE_PART=k) instantiates a few
// variants; the parts build in parallel (one variant of the stage-queued kernel is 10 - 25 s of hipcc, and there are
// more than twenty).  Kernels travel as untyped pointers and are launched with hipLaunchKernel: every unit sees the
// same device headers, so the argument layouts are identical by construction.  pine_kernels_fast.hip (declared
// tolerance) exports its variants the same way.
#pragma once
#include <cstddef>

struct PineKernelVariant {
  unsigned features;   // F_* feature set the variant covers
  int ctx;             // stage-queued kernel: path contexts per workgroup; megakernel: 0
  int order;           // position in the host's first-fit search (most specific first)
  int waves_per_simd;  // megakernel: occupancy the variant is compiled for; stage-queued kernel: kQBlock / 256
  size_t fixed_lds;    // stage-queued kernel: LDS bytes before the traversal stack
  size_t min_stack;    // ... and the least size of the stack region
  const void* fn;      // the __global__ function
  const char* name;
};
using PineFastVariant = PineKernelVariant;

constexpr int kPineKernelParts = 8;
// part k's table (k < kPineKernelParts); defined by pine_kernels_part.hip compiled with -DPINE_PART=k
#define PINE_PART_FN_(k) pine_gpu_kernel_part_##k
#define PINE_PART_FN(k) PINE_PART_FN_(k)
extern "C" {
const PineKernelVariant* pine_gpu_kernel_part_0(int* queue_count, int* mega_count);
const PineKernelVariant* pine_gpu_kernel_part_1(int* queue_count, int* mega_count);
const PineKernelVariant* pine_gpu_kernel_part_2(int* queue_count, int* mega_count);
const PineKernelVariant* pine_gpu_kernel_part_3(int* queue_count, int* mega_count);
const PineKernelVariant* pine_gpu_kernel_part_4(int* queue_count, int* mega_count);
const PineKernelVariant* pine_gpu_kernel_part_5(int* queue_count, int* mega_count);
const PineKernelVariant* pine_gpu_kernel_part_6(int* queue_count, int* mega_count);
const PineKernelVariant* pine_gpu_kernel_part_7(int* queue_count, int* mega_count);
// diagnostic builds (-DPINE_PROFILE_SECTIONS): each part adds its REGION counters to the caller's totals
int pine_gpu_kernel_part_regions_0(unsigned long long*, unsigned long long*);
int pine_gpu_kernel_part_regions_1(unsigned long long*, unsigned long long*);
int pine_gpu_kernel_part_regions_2(unsigned long long*, unsigned long long*);
int pine_gpu_kernel_part_regions_3(unsigned long long*, unsigned long long*);
int pine_gpu_kernel_part_regions_4(unsigned long long*, unsigned long long*);
int pine_gpu_kernel_part_regions_5(unsigned long long*, unsigned long long*);
int pine_gpu_kernel_part_regions_6(unsigned long long*, unsigned long long*);
int pine_gpu_kernel_part_regions_7(unsigned long long*, unsigned long long*);
}
const PineKernelVariant* pine_gpu_fast_variants(int* count);

// ---- the list: PINE_Q(part, order, features, contexts, name) stage-queued kernel; PINE_M(part, order, features,
// waves per SIMD, name) lane-owns-a-path megakernel.  The host takes the FIRST variant in `order` that covers a scene.
// (kFBoxes / kFAnalytic: pine_kernels_device.h)
//
// Stage-queued kernel.  Scenes that do not fit LDS whole (F_LDS_TOP): 1024 contexts; the top of the BVH (breadth-first
// numbering) is cached in whatever LDS the contexts and the 16-bit traversal stack leave.  The mesh-capable feature
// sets twice: with traversal stages (F_XSTAGE: XS / XC queues, lanes refilled -- taken for scenes WITH meshes whose BVH
// (nearly) fits the LDS node cache: the rays of a two-level BVH need very different numbers of trips and refilling
// pays, profiles/HISTORY.md 6.3), and with the flat traversal inside stages S / T (two queue hops per vertex fewer -- taken
// otherwise).  plan_build decides.  F_LDS_REST: few geometries (big meshes or not): their shape / leaf / material /
// light records are staged in LDS too.  Subsurface: the BSSRDF random walk is a third stage (W) with its own queue.
// (experiment builds, -DPINE_EXPERIMENT_C4X: the cone scene's kinds with traversal stages and 1536 / 1280 contexts)
#ifdef PINE_EXPERIMENT_C4X
#define PINE_EXPERIMENT_VARIANTS(PINE_Q)                                                                                       \
  PINE_Q(1, 20, F_SPHERE | F_DISK | F_CONE | F_UBER | F_LDS_TOP | F_XSTAGE, PINE_EXPERIMENT_C4X, "experiment: classic.pine's kinds, traversal stages, more contexts")
#else
#define PINE_EXPERIMENT_VARIANTS(PINE_Q)
#endif
#define PINE_VARIANT_LIST(PINE_Q, PINE_M)                                                                                                   \
  PINE_Q(0, 0, F_OBB | F_LDS_SCENE, PINE_QCTX, "queue: rect+transformed box/diffuse, scene in LDS") /* cbox exactly */                      \
  PINE_Q(0, 1, kFBoxes | F_LDS_SCENE, PINE_QCTX, "queue: rect+box/diffuse, scene in LDS")                                                   \
  PINE_Q(0, 2, kFAnalytic | F_LDS_SCENE, PINE_QCTX, "queue: analytic shapes/uber, scene in LDS")                                            \
  PINE_Q(0, 3, F_SPHERE | F_DISK | F_CONE | F_UBER | F_LDS_TOP, 1024,                                                                       \
         "queue: rect+sphere+disk+cone/uber, 1024 contexts, BVH top in LDS (classic.pine's kinds exactly)")                                 \
  PINE_Q(1, 4, kFAnalytic | F_LDS_TOP, 1024, "queue: analytic shapes/uber, 1024 contexts, BVH top in LDS")                                  \
  PINE_Q(2, 5, (F_ALL & ~F_SSS) | F_LDS_REST | F_LDS_TOP | F_XSTAGE, 1024,                                                                  \
         "queue: all but SSS, 1024 contexts, BVH top + scene records in LDS, traversal stages")                                             \
  PINE_Q(2, 6, (F_ALL & ~F_SSS) | F_LDS_REST | F_LDS_TOP, 1024, "queue: all but SSS, 1024 contexts, BVH top + scene records in LDS")       \
  PINE_Q(3, 7, (F_ALL & ~F_SSS) | F_LDS_TOP | F_XSTAGE, 1024, "queue: all but SSS, 1024 contexts, BVH top in LDS, traversal stages")        \
  PINE_Q(3, 8, (F_ALL & ~F_SSS) | F_LDS_TOP, 1024, "queue: all but SSS, 1024 contexts, BVH top in LDS")                                     \
  PINE_Q(1, 9, F_MESH | F_SSS | F_LDS_REST | F_LDS_TOP | F_XSTAGE, 1024,                                                                    \
         "queue: rect+mesh/diffuse+subsurface, walk stage, 1024 contexts, BVH top + scene records in LDS, traversal stages")                \
  PINE_Q(1, 10, F_MESH | F_SSS | F_LDS_REST | F_LDS_TOP, 1024,                                                                              \
         "queue: rect+mesh/diffuse+subsurface, walk stage, 1024 contexts, BVH top + scene records in LDS")                                  \
  PINE_Q(4, 11, F_ALL | F_LDS_REST | F_LDS_TOP | F_XSTAGE, 1024,                                                                            \
         "queue: all features, walk stage, 1024 contexts, BVH top + scene records in LDS, traversal stages")                                \
  PINE_Q(4, 12, F_ALL | F_LDS_REST | F_LDS_TOP, 1024, "queue: all features, walk stage, 1024 contexts, BVH top + scene records in LDS")    \
  PINE_Q(5, 13, F_ALL | F_LDS_TOP | F_XSTAGE, 1024, "queue: all features, walk stage, 1024 contexts, BVH top in LDS, traversal stages")     \
  PINE_Q(5, 14, F_ALL | F_LDS_TOP, 1024, "queue: all features, walk stage, 1024 contexts, BVH top in LDS")                                  \
  /* BVHs of 65 536 nodes and more: 32-bit traversal stack, no node cache */                                                                \
  PINE_Q(6, 15, F_ALL, 1024, "queue: all features, walk stage, 1024 contexts")                                                              \
  /* test hook: twins of variants 0 and 6 with the per-vertex log compiled in (PINE_GPU_FLAG_VERTEX_LOG; tests/test_bvh_fixtures.py) */     \
  PINE_Q(7, 16, F_OBB | F_LDS_SCENE | F_VLOG, PINE_QCTX, "queue: rect+transformed box/diffuse, scene in LDS, per-vertex log")                \
  PINE_Q(6, 17, (F_ALL & ~F_SSS) | F_LDS_REST | F_LDS_TOP | F_VLOG, 1024, "queue: all but SSS, 1024 contexts, BVH top + scene records in LDS, per-vertex log") \
  PINE_EXPERIMENT_VARIANTS(PINE_Q)                                                                                                           \
  /* PINE_GPU_FLAG_ORDER_EMBREE: closest hits in the order of the reference's EmbreeAccel; cbox-class scenes */ \
  PINE_Q(7, 18, kFAnalytic | F_LDS_SCENE | F_EMBREE, PINE_QCTX, "queue: analytic shapes/uber, scene in LDS, EmbreeAccel's order")         \
  PINE_Q(2, 19, F_ALL | F_EMBREE, 1024, "queue: all features, walk stage, 1024 contexts, EmbreeAccel's order")                              \
  PINE_M(0, 0, kFBoxes | F_LDS_SCENE, 4, "rect+box/diffuse, scene in LDS")                                                                  \
  PINE_M(6, 1, kFAnalytic | F_LDS_SCENE, 2, "analytic shapes/uber, scene in LDS")                                                           \
  PINE_M(6, 2, kFAnalytic, 2, "analytic shapes/uber")                                                                                       \
  PINE_M(7, 3, F_ALL | F_LDS_SCENE, 2, "all features, scene in LDS")                                                                        \
  PINE_M(7, 4, F_ALL, 2, "all features")                                                                                                    \
  PINE_M(5, 5, F_ALL | F_LDS_SCENE | F_EMBREE, 2, "all features, scene in LDS, EmbreeAccel's order")                                \
  PINE_M(3, 6, F_ALL | F_EMBREE, 2, "all features, EmbreeAccel's order")
