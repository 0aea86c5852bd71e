// tools/sanitize/nogpu_entry_points.cpp -- SANITIZER BUILD ONLY (tools/sanitize/Makefile), never part of the product.
//
// The host side of libpine_gpu.so (pine_amd/csrc/pine_host.cpp: scene building, shape constructors, BVH build,
// node folding, .pscene dump, film finalize) is plain C++ and is what parses caller-supplied data; the CPU leg of
// the test-suite exercises exactly that code.  GPU AddressSanitizer is not available on this pool, so the
// sanitizer build compiles pine_host.cpp with g++ -fsanitize=address,undefined and takes the device-side entry
// points (pine_amd/csrc/pine_kernels.hip) from here: every one of them fails the way the real library fails on
// a host without a HIP device.  Nothing is rendered by this build.
#include <cstdint>
#include <string>

#include "../../include/pine_gpu.h"
#include "../../pine_amd/csrc/pine_host.h"

#include "../../pine_amd/csrc/pine_specialize.h"
#include "../../pine_amd/csrc/pine_embree_order.h"

// (the structure sizes a run-time compiled kernel is checked against come from the device half of the library: none here --
//  a kernel compiled through this build's pine_gpu_test_specialize_compile fails its static_assert, as it should)
pine_gpu::AbiFingerprint pine_gpu::abi_fingerprint() { return pine_gpu::AbiFingerprint{}; }

using pine_gpu::set_error;
static const char* const kNoDevice =
    "no HIP device available: the PathIntegrator hot path requires an AMD GPU (no CPU fallback) [sanitizer build: host code only]";
static int fail() {
  set_error(kNoDevice);
  return -1;
}

extern "C" {
float pine_gpu_progress(void) { return 0.0f; }
void pine_gpu_release_cached_memory(void) {}
int pine_gpu_set_table_path(const char* path) { return path ? 0 : fail(); }
int pine_gpu_path_render(pine_gpu_scene*, const pine_gpu_render_params*, float*) { return fail(); }
int pine_gpu_path_render_multi(pine_gpu_scene*, const pine_gpu_render_params*, uint64_t, float*) { return fail(); }
int pine_gpu_path_render_devices(pine_gpu_scene*, const pine_gpu_render_params*, const int*, int, float*) { return fail(); }
pine_gpu_plan* pine_gpu_plan_create(pine_gpu_scene*, const pine_gpu_render_params*) {
  fail();
  return nullptr;
}
int pine_gpu_plan_launch(pine_gpu_plan*, void*, void*) { return fail(); }
int pine_gpu_plan_launch_packed(pine_gpu_plan*, void*, void*) { return fail(); }
void pine_gpu_plan_destroy(pine_gpu_plan*) {}
int pine_gpu_plan_stats_get(pine_gpu_plan*, pine_gpu_plan_stats*) { return fail(); }
int pine_gpu_plan_check(pine_gpu_plan*) { return fail(); }
int pine_gpu_plan_debug_sections(pine_gpu_plan*, uint64_t*) { return fail(); }
int pine_gpu_plan_read_samples(pine_gpu_plan*, float*, int64_t) { return fail(); }
int64_t pine_gpu_plan_vertex_log(pine_gpu_plan*, float*, int64_t) { return fail(); }
int64_t pine_gpu_packed_slab_floats(int film_w, int film_h, int world) {
  if (film_w <= 0 || film_h <= 0 || world < 1) return -1;
  const int64_t tiles = int64_t((film_w + 7) / 8) * ((film_h + 7) / 8);
  return (tiles + world - 1) / world * 64 * 4;
}
int pine_gpu_packed_offset(int film_w, int film_h, int world, int x, int y, int* rank_out, int64_t* float4_index_out) {
  if (film_w <= 0 || film_h <= 0 || world < 1 || x < 0 || y < 0 || x >= film_w || y >= film_h) {
    set_error("bad argument");
    return -1;
  }
  const int tile = (y / 8) * ((film_w + 7) / 8) + x / 8;
  if (rank_out) *rank_out = tile % world;
  if (float4_index_out) *float4_index_out = int64_t(tile / world) * 64 + (y % 8) * 8 + x % 8;
  return 0;
}
int pine_gpu_film_unpack(int, int, int, int, const void*, void*, void*) { return fail(); }
int pine_gpu_test_lomuto(const unsigned char*, int, int*, int*) { return fail(); }
int pine_gpu_test_sampler(int, int, float*, int64_t) { return fail(); }
int pine_gpu_test_rng(int, uint64_t*, int64_t) { return fail(); }
int pine_gpu_test_sincos(int, const float*, int64_t, float*, float*) { return fail(); }
int pine_gpu_test_powlog(int, const float*, const float*, int64_t, float*, float*) { return fail(); }
int pine_gpu_test_atan(int, const float*, const float*, int64_t, float*, float*) { return fail(); }
int pine_gpu_test_traverse(pine_gpu_scene*, int, const float*, int64_t, int, int, uint32_t*) { return fail(); }
// (host code: the real thing, so that the sanitizers see the hierarchy builder -- same body as in pine_kernels.hip)
int pine_gpu_test_embree_tree(const float* boxes, int n, int* words, int cap) {
  if (!boxes || !words || n < 0) return fail();
  std::vector<float> bx(boxes, boxes + 6 * size_t(n));
  std::vector<int> places(size_t(n), 0);
  for (int i = 0; i < n; i++) places[size_t(i)] = i;
  pine_gpu::EmbreeOrderTree tree;
  std::string why;
  if (!tree.build(bx, places, why)) return fail();
  if (1 + 8 * int(tree.nodes.size()) > cap) return fail();
  int k = 0;
  words[k++] = tree.root;
  for (const pine_gpu::EmbreeNode& nd : tree.nodes)
    for (int i = 0; i < 8; i++) words[k++] = nd.child[i];
  return k;
}
int pine_gpu_test_shapes(pine_gpu_scene*, int, const float*, int64_t, float*, int64_t) { return fail(); }
int pine_gpu_plan_test_traverse_baked(pine_gpu_plan*, const float*, int64_t, uint32_t*) { return fail(); }
}
