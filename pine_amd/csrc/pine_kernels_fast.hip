// pine_amd/csrc/pine_kernels_fast.hip -- PINE_GPU_FLAG_FAST: the stage-queued path kernel with declared-tolerance arithmetic.
//
// The parity build (pine_kernels.hip) reproduces the reference bit for bit, which costs: no multiply-add contraction,
// IEEE-correct division and square root everywhere, glibc-exact sin / cos / pow / log through binary64.  north_star asks
// only for a stated per-pixel L2 tolerance on floats, so this file compiles the SAME device sources a second time -- under
// another namespace, with PINE_FAST_MATH (pine_math.h) and -ffp-contract=fast -freciprocal-math -fapprox-func
// -fno-signed-zeros (Makefile) -- for the feature sets of the BASELINE scenes.  Selected per plan by PINE_GPU_FLAG_FAST;
// never the default, never the parity gate.  Tolerance and measured gain: DESIGN.md 7, tests/test_gpu_parity.py
// (test_fast_mode_within_declared_tolerance).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

#define PINE_FAST_MATH 1
#define pine_gpu pine_gpu_fast
#define pine_libm pine_libm_fast
#include "pine_kernels_device.h"
#undef pine_gpu
#undef pine_libm
#include "pine_variants.h"

namespace pine_gpu_fast {
#define PINE_FV(F, CTX, NAME)                                                                                        \
  {F, CTX, 0, kQBlock / 256, QLayout<CTX, q_num_queues(F), ((F) & F_LDS_TOP) != 0>::fixed_bytes, QLayout<CTX, q_num_queues(F), ((F) & F_LDS_TOP) != 0>::min_stack_bytes, \
   (const void*)path_queue_kernel<F, CTX>, NAME}
static const PineFastVariant kFast[] = {
    PINE_FV(F_OBB | F_LDS_SCENE, PINE_QCTX, "fast queue: rect+transformed box/diffuse, scene in LDS"),
    PINE_FV(kFAnalytic | F_LDS_SCENE, PINE_QCTX, "fast queue: analytic shapes/uber, scene in LDS"),
    PINE_FV(F_SPHERE | F_DISK | F_CONE | F_UBER | F_LDS_TOP, 1024, "fast queue: rect+sphere+disk+cone/uber, 1024 contexts, BVH top in LDS"),
    PINE_FV(F_MESH | F_SSS | F_LDS_TOP | F_XSTAGE | F_LDS_REST, 1024, "fast queue: rect+mesh/diffuse+subsurface, walk stage, 1024 contexts, BVH top + scene records in LDS"),
};
}  // namespace pine_gpu_fast

const PineFastVariant* pine_gpu_fast_variants(int* count) {
  *count = int(sizeof(pine_gpu_fast::kFast) / sizeof(pine_gpu_fast::kFast[0]));
  return pine_gpu_fast::kFast;
}
