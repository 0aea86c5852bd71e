// pine_amd/csrc/pine_fast.h -- what pine_kernels_fast.hip (the declared-tolerance kernel variants) exports to the host side in
// pine_kernels.hip.  Plain types only: the two translation units instantiate the same device headers under different
// namespaces (and different floating-point compile flags), so kernels travel as untyped pointers and are launched with
// hipLaunchKernel; their argument structs have identical layouts by construction (same headers).
#pragma once
#include <cstddef>

struct PineFastVariant {
  unsigned features;   // F_* feature set the variant covers
  int ctx;             // path contexts per workgroup
  size_t fixed_lds;    // LDS bytes before the traversal stack
  size_t min_stack;    // least size of the stack region
  const void* fn;      // __global__ path_queue_kernel<features, ctx> of the fast namespace
  const char* name;
};
const PineFastVariant* pine_gpu_fast_variants(int* count);
