/* oracle/pine_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C API of this repo's CPU restatement of wicstas/pine's PathIntegrator hot path (the "oracle").
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the
 * product (pine_amd/, include/pine_gpu.h) never links or calls it.
 */
#ifndef PINE_ORACLE_H
#define PINE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Stats filled by oracle_render (all counts are totals over the image). */
typedef struct {
  double seconds;          /* wall time of the render loop (includes BVH build, as the reference) */
  uint64_t camera_samples; /* W*H*spp_effective */
  uint64_t vertices;       /* radiance() invocations (unit of B_vertex) */
  uint64_t shadow_rays;    /* any-hit queries */
  uint64_t bsdf_samples;
  int threads;
  int spp_effective;
} oracle_stats;

/* Render `pscene` (text, see pine_amd/scene_io.py) with PathIntegrator(BVH, BlueSobolSampler(spp),
 * UniformLightSampler, depth).  `tables` = the packed bluesobol_u8.bin blob (2424832 bytes).
 * film_out: W*H*4 floats, row 0 first (path.cpp:38 layout).  threads<=0 => hardware_concurrency.
 * Restricts the render to pixels [y0,y1) rows when y1>y0 (others left untouched).
 * Returns 0 on success, nonzero + message in oracle_last_error() otherwise. */
int oracle_render(const char* pscene, const uint8_t* tables, int spp, int depth, int threads,
                  int y0, int y1, float* film_out, oracle_stats* stats);

/* Which sampler the render entry points construct: 0 = BlueSobolSampler(spp) (default), 1 = SobolSampler(spp)
 * (sampler.h:83-164), 2 = HaltonSampler(spp) (sampler.h:40-81); for the latter two spp is used as given.  Process-wide, not thread-safe: test use only. */
void oracle_set_sampler(int kind);
/* closest-hit primitive order: 0 = pine's BVH order (the parity oracle), 1 = nearest bounds first (SURVEY.md Appendix A3's second
 * order), 2 = the order of the reference's default accel, EmbreeAccel (restated from the vendored Embree's BVH8 builder and
 * single-ray traverser) */
void oracle_set_order(int mode);
/* test hook: the order in which mode 2 calls the user callback for one ray over n boxes (see pine_oracle.cpp) */
int oracle_embree_order(const float* boxes, int n, const float* ray8, const float* hit_t, int* ids, int cap, int* hit_id, float* tfar);
/* ... that hierarchy itself (root word, then 8 child words per node), and the closest-hit query of mode 2 on a scene's rays */
int oracle_embree_tree(const float* boxes, int n, int* words, int cap);
/* ... and Embree's triangle test as restated over a triangle list (closest hit): per ray prim (-1: miss), t, u, v, Ng */
int oracle_embree_triangles(const float* verts, const uint32_t* idx, int nt, const float* rays, int64_t nrays, float* out);
int oracle_embree_traverse(const char* pscene, const float* rays, int64_t nrays, int cap, uint32_t* out);

/* Render only the pixels of this shard (8x8 tiles dealt round-robin, the product's multi-GPU
 * partition); film_out must be zero-initialised by the caller, other pixels are left untouched. */
int oracle_render_shard(const char* pscene, const uint8_t* tables, int spp, int depth, int threads,
                        int shard_rank, int shard_world, float* film_out);

/* Per-sample radiance (before the per-pixel sum): out[(y*W+x)*spp + s] = (r,g,b,vertices). */
int oracle_render_samples(const char* pscene, const uint8_t* tables, int spp, int depth,
                          int threads, float* samples_out);

/* BlueSobolSampler stream in the exact layout of `pine_ref sampler` (oracle/ref_driver.cpp). */
int oracle_sampler_stream(const uint8_t* tables, int spp, float* out, int64_t capacity);
/* hash/RNG known answers in the layout of `pine_ref rng`: 6 pixels x 19 u64 words. */
int oracle_rng_stream(uint64_t* out, int64_t capacity);
/* Host math known answers in the layout of `pine_ref host`. */
int oracle_host_math(float* out, int64_t capacity);
/* Per-shape records in the layout of `pine_ref shapes` (11 floats per (geometry, ray)). */
int oracle_shapes(const char* pscene, const float* rays, int64_t nrays, float* out,
                  int64_t capacity);
/* libm-compatible sinf/cosf restatement check helpers (see pine_amd/csrc/pine_libm.h). */
const char* oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
