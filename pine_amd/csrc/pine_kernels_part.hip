// pine_amd/csrc/pine_kernels_part.hip -- one PART of the path kernels' specialisations (pine_variants.h): compiled
// kPineKernelParts times, with -DPINE_PART=0 ... kPineKernelParts-1, so that the twenty-odd instantiations of
// path_queue_kernel / path_trace_kernel build in parallel.  Exports one plain-typed table; pine_kernels.hip merges them.
//
// Experiment builds: -DPINE_ONLY_Q=<bit mask of queue-variant orders> / -DPINE_ONLY_M=<mask of megakernel orders>
// compile only those variants (the others stay in the table without a kernel and the host skips them).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

#include "pine_kernels_device.h"
#include "pine_variants.h"

#ifndef PINE_PART
#error "compile with -DPINE_PART=k"
#endif
#ifndef PINE_ONLY_Q
#define PINE_ONLY_Q 0xffffffffu
#endif
#ifndef PINE_ONLY_M
#define PINE_ONLY_M 0xffffffffu
#endif

namespace pine_gpu {
namespace {

template <bool ENABLED, unsigned F, int CTX>
const void* queue_fn() {
  if constexpr (ENABLED) return (const void*)path_queue_kernel<F, CTX>;
  else return nullptr;
}
template <bool ENABLED, unsigned F, int WPS>
const void* mega_fn() {
  if constexpr (ENABLED) return (const void*)path_trace_kernel<F, WPS>;
  else return nullptr;
}

// queue variants first, then megakernel variants (the two counts are returned separately)
#define PINE_Q_ENTRY(PART, ORDER, F, CTX, NAME)                                                                      \
  PineKernelVariant{F, CTX, ORDER, kQBlock / 256, QLayout<CTX, q_num_queues(F), ((F) & F_LDS_TOP) != 0>::fixed_bytes, \
                    QLayout<CTX, q_num_queues(F), ((F) & F_LDS_TOP) != 0>::min_stack_bytes,                          \
                    queue_fn<(PART) == PINE_PART && ((PINE_ONLY_Q >> (ORDER)) & 1u) != 0, F, CTX>(), NAME},
#define PINE_M_ENTRY(PART, ORDER, F, WPS, NAME) \
  PineKernelVariant{F, 0, ORDER, WPS, 0, 0, mega_fn<(PART) == PINE_PART && ((PINE_ONLY_M >> (ORDER)) & 1u) != 0, F, WPS>(), NAME},
#define PINE_SKIP(PART, ORDER, F, X, NAME)

const PineKernelVariant kAllQueue[] = {PINE_VARIANT_LIST(PINE_Q_ENTRY, PINE_SKIP)};
const PineKernelVariant kAllMega[] = {PINE_VARIANT_LIST(PINE_SKIP, PINE_M_ENTRY)};
constexpr int kNumQ = int(sizeof(kAllQueue) / sizeof(kAllQueue[0])), kNumM = int(sizeof(kAllMega) / sizeof(kAllMega[0]));
PineKernelVariant g_table[kNumQ + kNumM];

}  // namespace
}  // namespace pine_gpu

// this part's entries: the queue variants it compiled (fn != null), then the megakernel variants
extern "C" const PineKernelVariant* PINE_PART_FN(PINE_PART)(int* queue_count, int* mega_count) {
  using namespace pine_gpu;
  int n = 0, nq = 0;
  for (int i = 0; i < kNumQ; i++)
    if (kAllQueue[i].fn) g_table[n++] = kAllQueue[i];
  nq = n;
  for (int i = 0; i < kNumM; i++)
    if (kAllMega[i].fn) g_table[n++] = kAllMega[i];
  *queue_count = nq;
  *mega_count = n - nq;
  return g_table;
}

#ifdef PINE_PROFILE_SECTIONS
// diagnostic builds: add this part's REGION counters (per-translation-unit device globals) to the caller's totals
#define PINE_PART_REGIONS_FN_(k) pine_gpu_kernel_part_regions_##k
#define PINE_PART_REGIONS_FN(k) PINE_PART_REGIONS_FN_(k)
extern "C" int PINE_PART_REGIONS_FN(PINE_PART)(unsigned long long* lanes, unsigned long long* hits) {
  unsigned long long rl[16], rh[16];
  if (hipMemcpyFromSymbol(rl, HIP_SYMBOL(pine_gpu::g_region_lanes), sizeof rl) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(rh, HIP_SYMBOL(pine_gpu::g_region_hits), sizeof rh) != hipSuccess) return -1;
  for (int i = 0; i < 16; i++) {
    lanes[i] += rl[i];
    hits[i] += rh[i];
  }
  return 0;
}
#endif
