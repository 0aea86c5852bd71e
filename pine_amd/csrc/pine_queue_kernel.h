// pine_amd/csrc/pine_queue_kernel.h -- stage-queued persistent path kernel (included by
// pine_kernels.hip; uses its DeviceScene / SceneView / traversal templates).
//
// Why: in the lane-owns-a-path kernel (path_trace_kernel) a wave's 64 lanes are in different phases
// of radiance(): measured VALU lane utilisation is 53 %, and a lockstep workload costs 45 ps per
// path vertex against 78 ps for the real one.  Here a path is a *context* (12 dwords of state in
// LDS, 1.5 contexts per thread); two LDS ring queues hold the ids of contexts whose last ray has
// been traced and classified:
//     S  shade    : surface, light sampling, shadow ray, BSDF sampling, fold push, next ray
//     T  terminal : emission / miss, backward fold, sample store, next camera sample or work item
// Both stages end by tracing the context's NEW ray (closest hit) and classifying the hit, then push
// the context to S or T.  Every wave repeatedly pops up to 64 ids from the fuller queue and runs
// that one stage for them -- wave-uniform control flow, all lanes doing the same thing: the ballot
// compaction / regrouping "between bounces" happens every time a stage pushes its survivors.
//
// One workgroup of 1024 threads per CU (all 160 KB of LDS, 16 waves sharing the queues).
// Per-context cold data (sampler tile slice, RNG, fold stack) lives in L2-resident global memory.
// Every result is bit-identical to path_trace_kernel and to the oracle: the stages run the same
// device functions in the same order per path; only the scheduling differs.
#pragma once


namespace pine_gpu {

#ifndef PINE_QBLOCK
#define PINE_QBLOCK 1024
#endif
constexpr int kQBlock = PINE_QBLOCK;  // threads per workgroup (one workgroup per CU: kQBlock / 256 waves per SIMD)
#ifndef PINE_QCTX
#define PINE_QCTX 1536
#endif
constexpr int kQFields = 12;   // dwords of context state
// queues: S shade, T terminal; W walk step (Subsurface variants); XS / XC shadow / closest-hit rays waiting for the
// traversal stages (F_XSTAGE variants: pine_trav.h)
enum : int { QS_S = 0, QS_T = 1, QS_W = 2, QS_XS = 3, QS_XC = 4 };
constexpr int q_num_stage_queues(unsigned F) { return ((F & F_LDS_TOP) && (F & F_XSTAGE)) ? 5 : (F & F_SSS) ? 3 : 2; }
// Subsurface variants have two more rings that are not stages (see "sample tokens" below): K, the flags of the
// workgroup's sample-token slots, and F, the ids of contexts with nothing to do
constexpr int q_num_queues(unsigned F) { return q_num_stage_queues(F) + ((F & F_SSS) ? 2 : 0); }
// context state fields (SoA in LDS: cst[field * kQCtx + id])
enum : int { CF_OX, CF_OY, CF_OZ, CF_DX, CF_DY, CF_DZ, CF_TMAX, CF_ST, CF_PXY, CF_SBASE, CF_GEOM, CF_PRIM };
constexpr unsigned kStFresh = 0xffffffffu;  // context has no path yet
// control words: heads at 0..6; tails at 8..14 -- the S and T tails are ONE u64 (8..9) so that a stage reserves
// slots in both with one atomic
enum : int { QC_HEAD = 0, QC_TAIL = 8, QC_BUSY = 16, QC_LOCK = 17, QC_EXHAUSTED = 18, QC_PNEXT = 20, QC_PEND = 22, QC_ABORT = 24, QC_PIXELS = 25, QC_WORDS = 32 };
constexpr int kQTokenDwords = 8;  // a sample token (Subsurface variants): RNG state (4) | pixel, sample-buffer base, sample index, -
constexpr unsigned kQSpinLimit = 1u << 22;  // every spin loop is bounded: a protocol bug must end the kernel, not hang the GPU
constexpr int kQWinDwords = 5;  // per-thread sampler window: 3 dwords of ranking bytes (12 dimensions) + 2 of scrambling bytes

// LDS layout (dword offsets) for CTX path contexts per workgroup: the default is 1.5 contexts per thread;
// scenes whose BVH needs a deep traversal stack (4 KB of LDS per stack slot) use the 1024-context layout.
// ALIAS: the per-thread sampler window shares its LDS with the traversal stack (variants whose stage S makes all its
// draws before it traces: the F_LDS_TOP ones)
template <int CTX, int NQ = 2, bool ALIAS = false>
struct QLayout {
  static constexpr int ctx = CTX;
  static constexpr int ring = CTX <= 1024 ? 1024 : 2048;  // ring capacity (power of two >= CTX)
  static constexpr int off_ctl = 0;
  static constexpr int off_ring = off_ctl + QC_WORDS;  // NQ rings of `ring` dwords (id + 1, 0 = empty)
  static constexpr int off_state = off_ring + NQ * ring;
  static constexpr int off_sobol = off_state + kQFields * CTX;
  static constexpr int off_win = off_sobol + kLdsSamplerDims * 256 / 4;
  static constexpr int off_stack = ALIAS ? off_win : off_win + kQWinDwords * kQBlock;
  static constexpr size_t fixed_bytes = size_t(off_stack) * 4;
  static constexpr size_t min_stack_bytes = ALIAS ? size_t(kQWinDwords) * kQBlock * 4 : 0;  // the stack region also holds the window
};

// per-context global record (L2-resident): the pixel's RNG state (2 x u64); in the Subsurface variants also
// the BSSRDF beta channel of every pending level (2 bits each, bxdf.cpp:335 / path.cpp:121) and the state
// of a random walk in flight.  The sampler's per-pixel ranking and scrambling bytes are read straight from
// the 256 KB tables (L2-resident, shared by every context).
//   float4 0: RNG            1: beta flags lo, hi, -, -
//   float4 2: walk ray origin (at exit: the exit point) | channel    3: walk direction | tmax    4: exit normal
//   F_LDS_TOP variants, after those: the vertex's shadow ray for stage XS (origin, direction, tmax, flags)
constexpr int kQCtxGlobalDwordsPlain = 4, kQCtxGlobalDwordsSss = 20;
constexpr int q_ctx_trav_offset(unsigned F) { return (F & F_SSS) ? kQCtxGlobalDwordsSss : kQCtxGlobalDwordsPlain; }
constexpr int q_ctx_global_dwords(unsigned F) { return q_ctx_trav_offset(F) + ((F & F_LDS_TOP) ? kTravRecordDwords : 0); }

// Subsurface (F_SSS): the BSSRDF random walk (bxdf.cpp:329-353) is a stage of its own, W -- ONE free-flight
// step per pass (closest hit inside the shape, exponential free flight, exit or uniform-sphere scatter), the
// context re-queued between steps, so that a wave's 64 lanes are 64 walks in the same phase whatever their
// lengths.  Stage S starts a walk (refraction into the shape, channel pick) and is entered a second time when
// the walk has ended, with the exit point in the context's global record.
// The kernel's body is a device function so that two kinds of __global__ entry can share it: the precompiled
// path_queue_kernel<F, CTX> below, and the extern "C" entry a scene-specialised translation unit declares under a plain name
// (pine_specialize.h).  It reads cold kernel arguments straight from the kernel-argument segment (kS / kW below): every entry
// point must take exactly these arguments in this order, and the body must be inlined into it.
template <unsigned F, int CTX = PINE_QCTX>
__device__ __forceinline__ void path_queue_body(const DeviceScene& S, const WorkParams& W, const ulonglong2* __restrict__ ckpt, float4* __restrict__ samples,
                                                float* __restrict__ fold, uint32_t* __restrict__ ctxg, Counters* __restrict__ counters) {
  constexpr int kNQ = q_num_queues(F);            // rings in LDS
  constexpr int kNStage = q_num_stage_queues(F);  // ... of which the first kNStage are stage queues
  constexpr bool kFork = (F & F_SSS) != 0;        // sample tokens (below)
  constexpr bool kVlog = (F & F_VLOG) != 0;       // test hook: per-vertex log (WorkParams::vertex_log)
  constexpr int QS_K = kNStage, QS_F = kNStage + 1;
  constexpr bool kTop = (F & F_LDS_TOP) != 0;         // flat traversal (pine_trav.h), node cache, 16-bit per-wave stacks
  constexpr bool kX = kTop && (F & F_XSTAGE) != 0;    // ... as stages of its own: the XS / XC queues
#if defined(PINE_BAKED_SCENE) && !defined(PINE_BAKED_TOP)  /* scene-specialised build (pine_specialize.h): the scene's BVH and leaf records are immediates */
  constexpr bool kBaked = true;
  static_assert(!kX, "scenes with meshes are not baked");
  static_assert((F & F_BAKED) != 0, "a baked build instantiates the F_BAKED name");
#else
  constexpr bool kBaked = false;
#endif
  constexpr bool kFlat = kTop && !kBaked;  // rays are traced by the flat traversal (a specialised build: by the scene's own code)
  static_assert(!(F & F_LDS_REST) || kTop, "F_LDS_REST is an option of the F_LDS_TOP variants");
  static_assert(!(F & F_EMBREE) || (!kTop && !kBaked), "EmbreeAccel's order lives in scene_traverse: not in the flat traversal, not in baked scenes");
  static_assert(!(F & F_XSTAGE) || kTop, "F_XSTAGE is an option of the F_LDS_TOP variants");
  constexpr int kQCtxGlobalDwords = q_ctx_global_dwords(F);
  constexpr int kQCtxTravOffset = q_ctx_trav_offset(F);
  using L = QLayout<CTX, kNQ, kTop>;
  constexpr int kQCtx = L::ctx, kQRing = L::ring, kQOffCtl = L::off_ctl, kQOffRing = L::off_ring, kQOffState = L::off_state,
                kQOffSobol = L::off_sobol, kQOffWin = L::off_win, kQOffStack = L::off_stack;
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  // The kernel arguments are ~130 dwords, and the compiler keeps every one it meets inside the persistent loop in an SGPR for
  // the whole kernel: 80 - 300 of them spilled to VGPR lanes (v_writelane / v_readlane), a few VGPRs gone at 128 per lane.
  // The cold ones -- the camera (used once per camera sample), the sampler-table pointers, the work decomposition -- are
  // therefore read where they are used, from the kernel-argument segment itself (scalar loads, K$ hits), through a pointer
  // the compiler cannot see through: Sk / Wk are S / W, just not hoisted.
  typedef const __attribute__((address_space(4))) uint32_t* KargPtr;  // (constant address space: the loads are s_load)
  KargPtr kargs = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kargs));
  constexpr size_t kWOffset = ((sizeof(DeviceScene) + alignof(WorkParams) - 1) / alignof(WorkParams)) * alignof(WorkParams);
  // kS(field) / kW(field): that field of S / W, loaded here and now
  auto kload = [&](size_t byte_offset, auto* out) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
#pragma unroll
    for (size_t i = 0; i < sizeof(*out) / 4; i++) o[i] = kargs[byte_offset / 4 + i];
  };
#define kS(FIELD) ([&] { std::remove_cv_t<decltype(S.FIELD)> v_; kload(offsetof(DeviceScene, FIELD), &v_); return v_; }())
#define kW(FIELD) ([&] { std::remove_cv_t<decltype(W.FIELD)> v_; kload(kWOffset + offsetof(WorkParams, FIELD), &v_); return v_; }())
  constexpr int kSM = kSmLds | ((F & F_SOBOL) ? kSmSobol : 0);  // sampler front mode (pine_device.h)
  // SobolSampler / HaltonSampler in a scene with Subsurface: the sampler's dimension counter outgrows the nine bits of the packed
  // state (a BSSRDF walk draws three dimensions per step, SobolSampler's counter never wraps, HaltonSampler's wraps at 1000);
  // those variants keep it in a free word of the context's global record (float4 1, third word) while the sampler is not BlueSampler
  constexpr bool kBigDim = (F & F_SSS) != 0 && (F & F_SOBOL) != 0;
  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & 63;
  unsigned* const qctl = reinterpret_cast<unsigned*>(lds_raw + kQOffCtl);
  unsigned* const ring = reinterpret_cast<unsigned*>(lds_raw + kQOffRing);
  float* const cstf = reinterpret_cast<float*>(lds_raw + kQOffState);
  unsigned* const cstu = reinterpret_cast<unsigned*>(lds_raw + kQOffState);
  // Traversal stack: [slot][thread].  F_LDS_TOP variants: 16-bit node ids (the host guarantees < 65536 nodes), and both the
  // stack and the sampler window are laid out PER WAVE ([wave][slot or row][lane]) in one region that the two share: a
  // wave is in one stage at a time and stage S makes all its draws before it traces, so within a wave the window is dead
  // when the stack is live.  (Sharing [row][thread] arrays of the whole workgroup is NOT safe: another wave's stack
  // slots cover this wave's window rows while it still draws -- that version corrupted node ids and faulted.)
  using StackT = typename std::conditional<(F & F_LDS_TOP) != 0, unsigned short, int>::type;
  constexpr int kStride = (F & F_LDS_TOP) ? 64 : kQBlock;  // element stride between stack slots / window rows of one thread
  size_t wave_region_bytes = 0;
  if constexpr (F & F_LDS_TOP) {
    wave_region_bytes = size_t(S.stack_total) * 64 * sizeof(StackT);
    if (wave_region_bytes < size_t(kQWinDwords) * 64 * 4) wave_region_bytes = size_t(kQWinDwords) * 64 * 4;
  }
  char* const wave_region = reinterpret_cast<char*>(lds_raw + kQOffStack) + size_t(tid >> 6) * wave_region_bytes;
  StackT* const stack = (F & F_LDS_TOP) ? reinterpret_cast<StackT*>(wave_region) + lane : reinterpret_cast<StackT*>(lds_raw + kQOffStack) + tid;

  // ---- one-time staging ----
  {
    const uint4* src = reinterpret_cast<const uint4*>(S.tables.sobol);
    uint4* dst = reinterpret_cast<uint4*>(lds_raw + kQOffSobol);
    for (int i = tid; i < kLdsSamplerDims * 256 / 16; i += kQBlock) dst[i] = src[i];
  }
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
  // (F_EMBREE variants read these; the hierarchy and the mesh list move with the blob when it is staged in LDS)
  V.etree_root = S.etree_root;
  V.num_emesh = S.num_emesh;
  V.rcpps = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(S.blob) + S.off_rcpps);
  V.etree = reinterpret_cast<const EmbreeNode*>(reinterpret_cast<const char*>(S.blob) + S.off_etree);
  V.emesh = reinterpret_cast<const int*>(reinterpret_cast<const char*>(S.blob) + S.off_emesh);
  char* lds_after_stack = nullptr;
  if constexpr (F & F_LDS_TOP) {
    size_t stack_bytes = size_t(S.stack_total) * kQBlock * sizeof(StackT);
    if (stack_bytes < L::min_stack_bytes) stack_bytes = L::min_stack_bytes;
    lds_after_stack = reinterpret_cast<char*>(lds_raw + kQOffStack) + stack_bytes;
    if constexpr (F & F_LDS_REST) lds_after_stack += S.blob_bytes - S.off_shapes;  // (the records staged below)
    // the first S.lds_nodes nodes (breadth-first numbering: the top levels of the top-level BVH and of the mesh
    // BVHs) live in LDS behind the traversal stack
    if constexpr (F & F_MESH) {
      // the mesh triangles as packets (entries, then distinct vertices), when plan_build found room for them
      if (S.lds_tris) {
        const int entry_quads = (S.tri_packet_entries + 1) >> 1, n16 = entry_quads + S.tri_packet_verts;
        uint4* tdst = reinterpret_cast<uint4*>(lds_after_stack);
        for (int i = tid; i < n16; i += kQBlock) tdst[i] = S.tri_packets[i];
        V.lds_tri_entries = reinterpret_cast<const uint2*>(tdst);
        V.lds_tri_verts = reinterpret_cast<const float4*>(tdst + entry_quads);
        lds_after_stack += size_t(n16) * 16;
      }
    }
    uint4* dst = reinterpret_cast<uint4*>(lds_after_stack);
    const uint4* src = reinterpret_cast<const uint4*>(S.nodes);
    for (int i = tid; i < S.lds_nodes * 4; i += kQBlock) dst[i] = src[i];
    V.lds_nodes = reinterpret_cast<const DNode*>(dst);
    V.lds_node_count = S.lds_nodes;
  }
  if constexpr (F & F_LDS_SCENE) {
    static_assert(!(F & F_LDS_TOP), "F_LDS_SCENE already has every node in LDS");
    uint4* dst = reinterpret_cast<uint4*>(lds_raw + kQOffStack + S.stack_total * kQBlock);
    const int n16 = S.blob_bytes >> 4;
    for (int i = tid; i < n16; i += kQBlock) dst[i] = S.blob[i];
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
  } else if constexpr (F & F_LDS_REST) {
    // blob = nodes | shapes | materials | node programs | bvhs | leaf records | lights: everything after the nodes
    size_t stack_bytes = size_t(S.stack_total) * kQBlock * sizeof(StackT);
    if (stack_bytes < L::min_stack_bytes) stack_bytes = L::min_stack_bytes;
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds_raw + kQOffStack) + stack_bytes);
    const uint4* src = S.blob + (S.off_shapes >> 4);
    const int n16 = (S.blob_bytes - S.off_shapes) >> 4;
    for (int i = tid; i < n16; i += kQBlock) dst[i] = src[i];
    const char* base = reinterpret_cast<const char*>(dst) - S.off_shapes;
    V.nodes = S.nodes;
    V.shapes = reinterpret_cast<const DShape*>(base + S.off_shapes);
    V.materials = reinterpret_cast<const DMaterial*>(base + S.off_materials);
    V.bvhs = reinterpret_cast<const DBvh*>(base + S.off_bvhs);
    V.prims = nullptr;
    V.lights = reinterpret_cast<const DLight*>(base + S.off_lights);
    V.node_ops = reinterpret_cast<const DNodeOp*>(base + S.off_node_ops);
    V.leaf = reinterpret_cast<const DShape*>(base + S.off_leaf) - S.top_prim_begin;
    V.etree = reinterpret_cast<const EmbreeNode*>(base + S.off_etree);
    V.emesh = reinterpret_cast<const int*>(base + S.off_emesh);
  } else {
    V.leaf = S.leaf;
    V.nodes = S.nodes;
    V.shapes = S.shapes;
    V.materials = S.materials;
    V.bvhs = S.bvhs;
    V.prims = nullptr;
    V.lights = S.lights;
    V.node_ops = S.node_ops;
  }
  // queues: every context starts "fresh" in the terminal queue (stage T hands out work items)
  if (tid < QC_WORDS) qctl[tid] = 0;
  for (int i = tid; i < kNQ * kQRing; i += kQBlock) ring[i] = 0;
  __syncthreads();
  for (int i = tid; i < kQCtx; i += kQBlock) {
    ring[QS_T * kQRing + i] = unsigned(i + 1);
    cstu[CF_ST * kQCtx + i] = kStFresh;
  }
  if (tid == 0) qctl[QC_TAIL + QS_T] = kQCtx;
  __syncthreads();

  const size_t ctx_base = size_t(blockIdx.x) * kQCtx;
  auto fold_entry = [&](int id, int level) -> float4* {
    return reinterpret_cast<float4*>(fold) + ((ctx_base + size_t(id)) * size_t(S.max_path_length) + size_t(level)) * 2;
  };
  auto ctx_global = [&](int id) -> uint32_t* { return ctxg + (ctx_base + size_t(id)) * kQCtxGlobalDwords; };
  // sample tokens (kFork): the workgroup's kQRing slots sit behind ALL context records
  auto token_slot = [&](unsigned pos) -> uint32_t* {
    return ctxg + size_t(gridDim.x) * kQCtx * kQCtxGlobalDwords + (size_t(blockIdx.x) * kQRing + pos) * kQTokenDwords;
  };
  auto lds_load = [](const unsigned* p) -> unsigned { return __atomic_load_n(p, __ATOMIC_RELAXED); };
  // test hook (WorkParams::vertex_log): the record of the radiance() invocation at `level` of sample `s` of pixel pxy
  auto vlog = [&](unsigned pxy, int s, int level) -> float* {
    return W.vertex_log + ((size_t(pxy >> 16) * size_t(S.cam.W) + size_t(pxy & 0xffffu)) * size_t(S.spp) + size_t(s)) * size_t(S.max_path_length) * kVertexLogFloats +
           size_t(level) * kVertexLogFloats;
  };

  // push the contexts of lanes with to_s / to_t to the shade / terminal queue: ballot compaction,
  // and ONE 64-bit LDS atomic per wave reserves the slots in both rings (the two tails are one u64)
  auto push2 = [&](bool to_s, bool to_t, int id) {
    const unsigned long long ms = __ballot(to_s), mt = __ballot(to_t);
    if ((ms | mt) == 0) return;
    unsigned long long base = 0;
    if (lane == 0)
      base = atomicAdd(reinterpret_cast<unsigned long long*>(&qctl[QC_TAIL]),
                       (unsigned long long)__popcll(ms) | ((unsigned long long)__popcll(mt) << 32));
    const unsigned base_s = __builtin_amdgcn_readfirstlane(unsigned(base));
    const unsigned base_t = __builtin_amdgcn_readfirstlane(unsigned(base >> 32));
    const unsigned long long below = (1ull << lane) - 1ull;
    if (to_s | to_t) {
      const unsigned slot = to_s ? unsigned(QS_S * kQRing) + ((base_s + unsigned(__popcll(ms & below))) & (kQRing - 1))
                                 : unsigned(QS_T * kQRing) + ((base_t + unsigned(__popcll(mt & below))) & (kQRing - 1));
      __atomic_store_n(&ring[slot], unsigned(id + 1), __ATOMIC_RELAXED);
    }
  };
  // the other queues have one tail word each
  auto push_q = [&](int q, bool flag, int id) {
    const unsigned long long m = __ballot(flag);
    if (m == 0) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&qctl[QC_TAIL + q], unsigned(__popcll(m)));
    base = __builtin_amdgcn_readfirstlane(base);
    if (flag) {
      const unsigned slot = unsigned(q * kQRing) + ((base + unsigned(__popcll(m & ((1ull << lane) - 1ull)))) & (kQRing - 1));
      __atomic_store_n(&ring[slot], unsigned(id + 1), __ATOMIC_RELAXED);
    }
  };
  auto push_w = [&](bool to_w, int id) { push_q(QS_W, to_w, id); };

  // samples of a pixel that form one sequential chain (kFork): the whole pixel -- the item itself without tile classes
  const int chain_spi = W.serial_tiles > 0 ? S.spp : W.samples_per_item;
  // "sample index s closes its chain / item": a chain is a power of two that divides spp, or -- SobolSampler / HaltonSampler with
  // another count -- the whole pixel (then chain_spi == spp and the chain starts at sample 0)
  const bool chain_pow2 = (chain_spi & (chain_spi - 1)) == 0;
  auto chain_ends_at = [&](int s_next) -> bool { return chain_pow2 ? (s_next & (chain_spi - 1)) == 0 : s_next == chain_spi; };
  SEC_DECL;
  // record a finished closest-hit query in the context and classify the vertex it reaches:
  // emissive / miss / path-length limit -> terminal queue, otherwise -> shade queue
  auto finish_hit = [&](int id, bool hit, float tmax, int geom, int prim, PackedState st, bool& to_shade, bool& to_term) {
    cstf[CF_TMAX * kQCtx + id] = tmax;
    cstu[CF_GEOM * kQCtx + id] = unsigned(hit ? geom : -1);
    cstu[CF_PRIM * kQCtx + id] = unsigned(prim);
    // (geom is the packed primitive word: the emissive flag rides along, no record fetch needed)
    const bool terminal = !hit || (geom & kPrimEmissiveBit) != 0 || st.length() + 1 >= S.max_path_length;
    to_shade = !terminal;
    to_term = terminal;
  };
#ifdef PINE_TRIP_STATS
  // diagnostic: per stage pass, the wave's longest shadow traversal + its longest closest-hit traversal (what two loops
  // cost) against its longest shadow + closest sequence of ONE lane (what a merged loop would cost); needs -DPINE_PROFILE_REGIONS
  int stat_shadow_trips = 0;
  auto trip_stats = [&](int& shadow_trips, int closest_trips) {
    int ms = shadow_trips, mc = closest_trips, mb = shadow_trips + closest_trips;
    for (int off = 32; off > 0; off >>= 1) {
      ms = max(ms, __shfl_xor(ms, off));
      mc = max(mc, __shfl_xor(mc, off));
      mb = max(mb, __shfl_xor(mb, off));
    }
    if (lane == 0) {
      atomicAdd(&g_region_lanes[10], (unsigned long long)(ms + mc));
      atomicAdd(&g_region_hits[10], 1ull);
      atomicAdd(&g_region_lanes[11], (unsigned long long)mb);
      atomicAdd(&g_region_hits[11], 1ull);
    }
    shadow_trips = 0;
  };
#endif
  auto trav_record = [&](int id) -> uint32_t* { return ctxg + (ctx_base + size_t(id)) * kQCtxGlobalDwords + kQCtxTravOffset; };
#ifdef PINE_BAKED_TOP
  // Top level baked (pine_specialize.h), closest-hit rays: the first pass over the top-level primitives runs where the ray is
  // CREATED (stages S / T: every lane of the wave has one).  A ray that does not reach the mesh is resolved here -- true:
  // hit word and tmax are in the context as finish_hit leaves them -- and never sees the traversal stages.  Otherwise the
  // context carries what stage XC needs: CF_TMAX = tmax at the mesh's place, CF_PRIM = tmax after ALL top-level primitives,
  // CF_GEOM = ids of the hit at the mesh's place and of the final hit (4 bits each, 0 = none) | bit 8: a camera ray (its
  // original tmax is the float maximum; a spawned ray's is spawn_ray's constant -- the replay starts from it).
  auto top_first_pass = [&](int id, f3 o, f3 d, float tmax) -> bool {
    DRay r{o, d, 0.0f, tmax};
    BakedMesh bm{false, 0.0f, -1};
    int g = -1;
    const bool hit = scene_traverse_baked_top<false, F, 0>(r, g, bm);
    if (!bm.reached) {
      cstf[CF_TMAX * kQCtx + id] = r.tmax;
      cstu[CF_GEOM * kQCtx + id] = unsigned(hit ? baked_top_word(g) : -1);
      cstu[CF_PRIM * kQCtx + id] = 0u;
      return true;
    }
    const bool camera = tmax == kFloatMax;
    if (!camera && tmax != kFloatMax * (1.0f - 1e-3f)) {  // (no other closest-hit ray exists: bail-out code 6, reported by the host)
      __atomic_store_n(&qctl[QC_ABORT], 1u, __ATOMIC_RELAXED);
      atomicAdd(&counters->bail_count, 1ull);
      counters->bail_code = 6;
      counters->bail_a = __float_as_uint(tmax);
      counters->bail_b = 0;
    }
    cstf[CF_TMAX * kQCtx + id] = bm.t;
    cstf[CF_PRIM * kQCtx + id] = r.tmax;
    cstu[CF_GEOM * kQCtx + id] = unsigned(bm.geom + 1) | (unsigned((hit ? g : -1) + 1) << 4) | (camera ? 256u : 0u);
    return false;
  };
#endif
  // trace the context's new ray (closest hit), record the hit, classify (finish_hit).  X variants: the ray waits in
  // the context and the context goes to XC; the traversal stage traces it.
  auto extend = [&](int id, f3 o, f3 d, float tmax, PackedState st, bool& to_shade, bool& to_term, bool& to_xc) {
    DRay ray{o, d, 0.0f, tmax};
    SEC_MARK(5);  // S: BSDF sample + fold store + spawn  /  T: camera ray
    cstf[CF_OX * kQCtx + id] = o.x;
    cstf[CF_OY * kQCtx + id] = o.y;
    cstf[CF_OZ * kQCtx + id] = o.z;
    cstf[CF_DX * kQCtx + id] = d.x;
    cstf[CF_DY * kQCtx + id] = d.y;
    cstf[CF_DZ * kQCtx + id] = d.z;
    cstu[CF_ST * kQCtx + id] = st.v;
    if constexpr (kX) {
#if defined(PINE_BAKED_TOP) && !defined(PINE_TOP_CLOSEST_IN_XC)
      if (top_first_pass(id, o, d, tmax)) {
        const int geom = int(cstu[CF_GEOM * kQCtx + id]);
        finish_hit(id, geom >= 0, cstf[CF_TMAX * kQCtx + id], geom, 0, st, to_shade, to_term);
      } else {
        to_xc = true;  // the mesh's BVH: stage XC
      }
#else
      cstf[CF_TMAX * kQCtx + id] = tmax;
      to_xc = true;  // traced by stage XC
#endif
    } else if constexpr (kFlat) {
      // the flat traversal, here and now: every lane of the wave to its end
      TravState ts;
      trav_begin(V, ts);
      const DRayOct oct = make_oct(ray);
#ifdef PINE_TRIP_STATS
      int ctrips = 0;
      trav_trips<false, F, kStride>(V, ray, oct, ts, stack, 0, 1 << 30, &ctrips);
      trip_stats(stat_shadow_trips, ctrips);
#else
      trav_trips<false, F, kStride>(V, ray, oct, ts, stack, 0, 1 << 30);
#endif
      SEC_MARK(6);  // closest-hit traversal
      finish_hit(id, ts.hit_geom >= 0, ray.tmax, ts.hit_geom, ts.hit_prim, st, to_shade, to_term);
    } else {
      int geom = -1, prim = 0;
#if defined(PINE_BAKED_SCENE) && !defined(PINE_BAKED_TOP)
      const bool hit = scene_traverse_baked<false, F>(ray, geom);
#else
      const bool hit = scene_traverse<false, F, kStride>(V, ray, stack, geom, prim);
#endif
      SEC_MARK(6);  // closest-hit traversal
      finish_hit(id, hit, ray.tmax, geom, prim, st, to_shade, to_term);
    }
  };

  unsigned shadow_count = 0, walk_count = 0;
  unsigned spins = 0;
  unsigned pick_no = 0;  // picks since this wave's last starvation-guard pick
  unsigned idle_polls = 0;
  // bounded-spin bail-out: record where, raise the workgroup's abort flag, leave
  auto bail = [&](unsigned code, unsigned a, unsigned b) {
    __atomic_store_n(&qctl[QC_ABORT], 1u, __ATOMIC_RELAXED);
    atomicAdd(&counters->bail_count, 1ull);
    counters->bail_code = code;
    counters->bail_a = a;
    counters->bail_b = b;
  };

  // (kFork) contexts with nothing to do wait in the ring F; take up to `count` of them back into stage T (whole wave)
  auto wake_free = [&](unsigned count) {
    if constexpr (kFork) {
      const unsigned fh = __builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_HEAD + QS_F]));
      const unsigned favail = __builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_TAIL + QS_F])) - fh;
      if (favail != 0u && count != 0u) {
        const unsigned want = favail < count ? favail : count;
        unsigned got = 0;
        if (lane == 0) got = atomicCAS(&qctl[QC_HEAD + QS_F], fh, fh + want) == fh ? 1u : 0u;  // (lost the race: somebody else woke them)
        if (__builtin_amdgcn_readfirstlane(got)) {
          int fid = -1;
          if (lane < want) {
            unsigned* slot = &ring[QS_F * kQRing + ((fh + lane) & (kQRing - 1))];
            unsigned v, tries = 0;
            while ((v = atomicExch(slot, 0u)) == 0u) {
              if (++tries > kQSpinLimit) {
                bail(6, unsigned(QS_F), fh + lane);
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            fid = int(v) - 1;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          push2(false, fid >= 0, fid);
        }
      }
    }
  };

  // ---- traversal stage with refill (X variants) ----
  // A wave that picked XS / XC keeps its lanes' traversals in registers and loops: advance every travelling lane a few
  // trips; retire the lanes that finished (XS: clear the direct term of an occluded vertex, context -> XC or T; XC:
  // record the hit, context -> S or T); hand the free lanes NEW contexts from the same queue.  So the rays of a wave
  // are in different phases of their traversals, but nearly all 64 lanes are travelling as long as the queue has
  // entries -- where one traversal per stage pass runs each wave until its longest ray is done (measured: 19.5 of 64
  // lanes travelling in an average trip of the icosphere scene, 37 on the 10 000-cone scene).
  const int trav_keep_lanes = W.trav_min_lanes, trav_min_trips = W.trav_min_trips;
  auto trav_stage = [&](auto any_tag, int id) {
    constexpr bool ANY = decltype(any_tag)::value;
    constexpr int Q = ANY ? QS_XS : QS_XC;
    bool fresh = id >= 0;
    TravState ts;
    trav_begin(V, ts);
    ts.done = 1;
    DRay ray{mk3(0.0f), mk3(1.0f), 0.0f, 0.0f};
    DRayOct oct = make_oct(ray);
    unsigned flags = 0;
    while (true) {
      // ---- the rays of the lanes that just got a context ----
      if (fresh) {
        REGION(ANY ? 9 : 10);  // (lanes that start a ray in this round of a traversal stage)
        if (ANY) {
          const float4* r4 = reinterpret_cast<const float4*>(trav_record(id));
          const float4 a = r4[0], b = r4[1];
          ray = DRay{f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, 0.0f, b.z};
          flags = __float_as_uint(b.w);
        } else {
          ray.o = f3{cstf[CF_OX * kQCtx + id], cstf[CF_OY * kQCtx + id], cstf[CF_OZ * kQCtx + id]};
          ray.d = f3{cstf[CF_DX * kQCtx + id], cstf[CF_DY * kQCtx + id], cstf[CF_DZ * kQCtx + id]};
          ray.tmin = 0.0f;
          ray.tmax = cstf[CF_TMAX * kQCtx + id];
        }
        oct = make_oct(ray);
        trav_begin(V, ts);
#ifdef PINE_BAKED_TOP
        {
          // The top level as code, once per ray (pine_specialize.h): its primitives are tested here, in pine's order, and the
          // flat machine gets the mesh's BVH alone.  TravState's return-to-the-top-level fields are free in this mode and
          // carry the first pass's results: r_pa / r_pan = tmax and hit after ALL top-level primitives, r_pb = "a primitive
          // stored after the mesh was accepted", r_next = 1 "the mesh's place was reached".
          static_assert(kX, "the top level is baked for the traversal-stage variants");
          BakedMesh bm{false, 0.0f, -1};
          int g = -1;
          ts.done = 1;
          ts.r_next = 0;
          if (ANY) {
#ifdef PINE_TOP_SHADOW_IN_XS  /* (measurement: the shadow rays' top-level pass here instead of in stage S) */
            const bool occluded = scene_traverse_baked_top<true, F, 0>(ray, g, bm);
            ts.hit_geom = occluded ? 0 : -1;
            if (occluded) bm.reached = false;
#else
            ts.hit_geom = -1;  // (stage S ran the top level for this ray and found it clear up to the mesh)
            bm.reached = true;
#endif
          } else {
#ifdef PINE_TOP_CLOSEST_IN_XC  /* (measurement: the closest-hit rays' first pass here instead of where the ray is created) */
            const bool ray_is_camera = ray.tmax == kFloatMax;
            const bool hit = scene_traverse_baked_top<false, F, 0>(ray, g, bm);
            ts.hit_geom = hit ? baked_top_word(g) : -1;
            if (bm.reached) {
              ts.r_pa = __float_as_int(ray.tmax), ts.r_pan = ts.hit_geom;
              ts.r_pb = (hit ? g : -1) != bm.geom ? 1 : 0;  // (an accepted primitive replaces the hit id: ids are unique)
              ts.r_pbn = ray_is_camera ? 1 : 0;
              ray.tmax = bm.t;
              ts.hit_geom = bm.geom >= 0 ? baked_top_word(bm.geom) : -1;
            }
#else
            // the first pass ran where the ray was created (top_first_pass): ray.tmax = tmax at the mesh's place already
            const unsigned packed = cstu[CF_GEOM * kQCtx + id];
            const int id_a = int(packed & 15u) - 1, id_r = int((packed >> 4) & 15u) - 1;
            ts.hit_geom = id_a >= 0 ? baked_top_word(id_a) : -1;
            ts.r_pa = __float_as_int(cstf[CF_PRIM * kQCtx + id]);
            ts.r_pan = id_r >= 0 ? baked_top_word(id_r) : -1;
            ts.r_pb = id_r != id_a ? 1 : 0;
            ts.r_pbn = int((packed >> 8) & 1u);
            bm.reached = true;
#endif
          }
          if (bm.reached) {
            const DBvh mb = V.bvhs[1];
            ts.r_next = 1;
            ts.done = 0;
            ts.mesh_base = 0;
            ts.mesh_word = kBakedMeshWord;
            ts.pb = ts.pbn = 0;
            if (mb.root_count > 0) ts.next = -1, ts.pa = mb.root_start, ts.pan = mb.root_count;
            else ts.next = mb.root, ts.pa = ts.pan = 0;
          }
        }
#endif
        fresh = false;
      }
#ifdef PINE_BAKED_TOP
      // Most rays end in the top-level code (they never reach the mesh): while fewer than the refill threshold travel and the
      // queue has rays, the finished lanes are retired and refilled at once -- the trips start on a full wave.
      bool run_trips = true;
      if (__popcll(__ballot(ts.done == 0)) < trav_keep_lanes) {
        const unsigned h0 = __builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_HEAD + Q]));
        if (__builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_TAIL + Q])) != h0) run_trips = false;
      }
      if (run_trips)
#endif
      trav_trips<ANY, F, kStride>(V, ray, oct, ts, stack, trav_keep_lanes, trav_min_trips);
      SEC_MARK(ANY ? 3 : 6);
      // ---- retire ----
      bool to_shade = false, to_term = false, to_xc = false;
      const bool retire = id >= 0 && ts.done != 0;
      if (retire) {
        const PackedState st{cstu[CF_ST * kQCtx + id]};
        if (ANY) {
          const bool occluded = ts.hit_geom >= 0;
          if (flags & kTravTerminalAfterShadow) {
            if (occluded) {  // Lo = min(beta * 0, 8) = 0
              cstf[CF_DX * kQCtx + id] = 0.0f;
              cstf[CF_DY * kQCtx + id] = 0.0f;
              cstf[CF_DZ * kQCtx + id] = 0.0f;
              if (kVlog && W.vertex_log) {
                float* r = vlog(cstu[CF_PXY * kQCtx + id], st.s_cur(), st.length());
                r[2] = r[3] = r[4] = 0.0f;
              }
            }
            to_term = true;
          } else {
            if (occluded) {  // the vertex that cast this ray is one level up: its direct term is zero
              float4* q = fold_entry(id, st.length() - 1);
              const float4 e = q[0];
              q[0] = make_float4(0.0f, 0.0f, 0.0f, e.w);
              if (kVlog && W.vertex_log) {
                float* r = vlog(cstu[CF_PXY * kQCtx + id], st.s_cur(), st.length() - 1);
                r[2] = r[3] = r[4] = 0.0f;
              }
            }
            if (flags & kTravClosestResolved) {  // (top level baked: its new ray never reached the mesh -- hit word and tmax are in the context)
              const int geom = int(cstu[CF_GEOM * kQCtx + id]);
              finish_hit(id, geom >= 0, cstf[CF_TMAX * kQCtx + id], geom, 0, st, to_shade, to_term);
            } else {
              to_xc = true;  // its new ray waits in the context
            }
          }
        } else {
#ifdef PINE_BAKED_TOP
          if (ts.r_next == 1) {  // the ray went through the mesh's BVH
            if (ts.hit_geom != kBakedMeshWord) {
              // no triangle accepted: every bound of the first pass was the reference's -- its result stands
              ray.tmax = __int_as_float(ts.r_pa), ts.hit_geom = ts.r_pan;
            } else if (ts.r_pb && __int_as_float(ts.r_pa) < ray.tmax) {
              // a triangle at ray.tmax, and a primitive stored AFTER the mesh had been accepted against the larger bound with
              // a SMALLER t (were the first pass's final tmax not below the triangle's, no such primitive could pass the
              // reference's test `t < tmax` either: whatever the reference accepts, the first pass accepted or bettered):
              // replay the top level with the mesh's result in its place (the reference's own sequence)
              DRay rr{ray.o, ray.d, 0.0f, ts.r_pbn ? kFloatMax : kFloatMax * (1.0f - 1e-3f)};  // (camera_gen_ray's / spawn_ray's tmax)
              BakedMesh bm{true, ray.tmax, -1};
              int g = -1;
              const bool hit = scene_traverse_baked_top<false, F, 1>(rr, g, bm);
              ray.tmax = rr.tmax;
              ts.hit_geom = hit ? baked_top_word(g) : -1;
            }
          }
#endif
          finish_hit(id, ts.hit_geom >= 0, ray.tmax, ts.hit_geom, ts.hit_prim, st, to_shade, to_term);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      push2(to_shade, to_term, id);
      if (ANY) push_q(QS_XC, to_xc, id);
      if (retire) id = -1;
      // ---- refill the free lanes from this stage's queue ----
      const unsigned long long free_lanes = __ballot(id < 0);
      if (free_lanes != 0) {
        const unsigned h0 = __builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_HEAD + Q]));  // (head before tail; the CAS below confirms the head)
        const unsigned avail = __builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_TAIL + Q])) - h0;
        if (avail != 0u) {
          const unsigned nfree = unsigned(__popcll(free_lanes));
          const unsigned want = avail < nfree ? avail : nfree;
          unsigned got = 0;
          if (lane == 0) got = atomicCAS(&qctl[QC_HEAD + Q], h0, h0 + want) == h0 ? 1u : 0u;  // (lost the race: try again next round)
          if (__builtin_amdgcn_readfirstlane(got)) {
            const unsigned rank = unsigned(__popcll(free_lanes & ((1ull << lane) - 1ull)));
            if (id < 0 && rank < want) {
              unsigned* slot = &ring[Q * kQRing + ((h0 + rank) & (kQRing - 1))];
              unsigned v, tries = 0;
              while ((v = atomicExch(slot, 0u)) == 0u) {
                if (++tries > kQSpinLimit) {
                  bail(2, unsigned(Q), h0 + rank);
                  break;
                }
                __builtin_amdgcn_s_sleep(1);
              }
              id = int(v) - 1;
              fresh = id >= 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          }
        }
        if (__ballot(id >= 0) == 0) break;
      }
      if (__builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_ABORT])) != 0u) break;
    }
  };

  // (No cap on the number of trips: a trip that finds work retires at least one stage of one path, and the
  // work of a launch is finite; the trips that find none are bounded in wall-clock time below.)
  unsigned long long idle_since = 0;
#ifdef PINE_PROFILE_SECTIONS
  if (tid == 0) atomicCAS(&counters->t_start, 0ull, wall_clock64());
#endif
  if (kW(debug_force_bail) && blockIdx.x == 0 && tid == 0) bail(7, 0, 0);  // test hook: the host must report this launch as failed
  __syncthreads();
  while (true) {
    SEC_MARK(10);  // push + busy release
    // ---------------- pick a stage: the fullest queue (lane 0 decides, result broadcast) ----------------
    // All lanes read the same control words (LDS broadcast) and the values are made wave-uniform
    // (readfirstlane), so the decision itself is scalar code; only the atomics run on lane 0.
    int stage = -1;
    unsigned n = 0, h = 0;
    bool finished = false;
    {
      auto uload = [&](int w) -> unsigned { return __builtin_amdgcn_readfirstlane(lds_load(&qctl[w])); };
      // heads first, then tails: heads and tails only grow, so a count can only be over-estimated
      unsigned hq[kNStage], cq[kNStage];
#pragma unroll
      for (int q = 0; q < kNStage; q++) hq[q] = uload(QC_HEAD + q);
#pragma unroll
      for (int q = 0; q < kNStage; q++) cq[q] = uload(QC_TAIL + q) - hq[q];
      const unsigned busy = uload(QC_BUSY);
      int best = cq[QS_T] > cq[QS_S] ? QS_T : QS_S;
      unsigned cnt_best = cq[best], h_best = hq[best], cnt_all = cq[QS_S] + cq[QS_T];
#pragma unroll
      for (int q = 2; q < kNStage; q++) {
        cnt_all += cq[q];
        if (cq[q] > cnt_best) {
          best = q;
          cnt_best = cq[q];
          h_best = hq[q];
        }
      }
      // Starvation guard: "the fullest queue" alone never serves a queue that stays short while the others are long -- the
      // walk queue W when only a few pixels see the Subsurface shape: their sample chains then crawl (measured on C5 with
      // tile classes: chains that need 20 ms took the whole 155 ms launch).  Every W.fair_period-th pick of a wave takes
      // the walk queue, however few entries it has.
      bool fair_pick = false;
      if constexpr ((F & F_SSS) != 0) {
        const unsigned period = unsigned(W.fair_period < 0 ? -W.fair_period : W.fair_period);
        if (period != 0u && ++pick_no >= period) {
          pick_no = 0;
          if (W.fair_period > 0) {  // the walk queue
            if (cq[QS_W] != 0u) {
              best = QS_W;
              cnt_best = cq[QS_W];
              h_best = hq[QS_W];
              fair_pick = true;
            }
          } else {  // (measurement aid) the shortest non-empty queue, whichever it is
            unsigned cnt_min = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < kNStage; q++) {
              if (cq[q] != 0u && cq[q] < cnt_min) {
                cnt_min = cq[q];
                best = q;
                h_best = hq[q];
              }
            }
            if (cnt_min != 0xffffffffu) {
              cnt_best = cnt_min;
              fair_pick = true;
            }
          }
        }
      }
      if (cnt_best >= 64u || fair_pick || (cnt_best > 0u && (busy == 0u || spins >= unsigned(W.pick_spins)))) {
        // count ourselves busy BEFORE taking items out of the queue, so that "all queues empty and
        // nobody busy" really means no work can appear any more (idle pollers never touch the
        // counter: two of them must not keep each other alive)
        const unsigned h0 = h_best;
        const unsigned want = cnt_best < 64u ? cnt_best : 64u;
        unsigned got = 0;
        if (lane == 0) {
          atomicAdd(&qctl[QC_BUSY], 1u);
          if (atomicCAS(&qctl[QC_HEAD + best], h0, h0 + want) == h0) got = 1;  // head unchanged => [h0, h0+want) is ours (tail >= the value read)
          else atomicSub(&qctl[QC_BUSY], 1u);
        }
        if (__builtin_amdgcn_readfirstlane(got)) {
          stage = best;
          n = want;
          h = h0;
        }
      } else {
        finished = cnt_all == 0u && busy == 0u;
      }
    }
    if (__builtin_amdgcn_readfirstlane(lds_load(&qctl[QC_ABORT])) != 0u) break;
    if (stage < 0) {
      SEC_MARK(11);  // idle poll
      if (finished) break;
      spins++;
      // idle bound in wall-clock time (100 MHz s_memrealtime), not in polls: other waves of the workgroup
      // may legitimately keep the last long paths of a launch for seconds
      const unsigned long long now = wall_clock64();
      if (idle_polls++ == 0) idle_since = now;
      if (now - idle_since > kW(idle_budget_ticks)) {
        if (lane == 0)
          bail(1, lds_load(&qctl[QC_BUSY]),
               (lds_load(&qctl[QC_TAIL]) - lds_load(&qctl[QC_HEAD])) | ((lds_load(&qctl[QC_TAIL + 1]) - lds_load(&qctl[QC_HEAD + 1])) << 16));
        break;
      }
      __builtin_amdgcn_s_sleep(4);
      continue;
    }
    spins = 0;
    idle_polls = 0;
    bool valid = lane < n;
    int id = -1;
    if (valid) {
      // take the slot with an atomic exchange: reading and clearing are ONE operation.  (A separate
      // plain "slot = 0" store may be sunk by the compiler to the end of the stage; by then the ring
      // can have wrapped and the late store wipes a NEW entry -- its consumer then spins forever.
      // That was an intermittent hang of the first version.)
      unsigned* slot = &ring[stage * kQRing + ((h + lane) & (kQRing - 1))];
      unsigned v, tries = 0;
      while ((v = atomicExch(slot, 0u)) == 0u) {
        if (++tries > kQSpinLimit) {
          bail(2, unsigned(stage), h + lane);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      id = int(v) - 1;
    }
    if (id < 0) valid = false;  // (only after a bounded-spin bail-out)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    SEC_MARK(0);  // pick + pop

    if (stage == QS_S) {
      // ================= shade a non-terminal vertex (path.cpp:91-120) =================
      bool to_shade = false, to_term = false, to_walk = false, to_xs = false, to_xc = false;
      bool release = false;  // (kFork) this lane's path seals here and its pixel has samples left: a token goes out
      if (valid) {
        REGION(12);  // (lanes of a stage-S pass that hold a context)
        const f3 ray_o{cstf[CF_OX * kQCtx + id], cstf[CF_OY * kQCtx + id], cstf[CF_OZ * kQCtx + id]};
        const f3 ray_d{cstf[CF_DX * kQCtx + id], cstf[CF_DY * kQCtx + id], cstf[CF_DZ * kQCtx + id]};
        const float ray_tmax = cstf[CF_TMAX * kQCtx + id];
        PackedState st{cstu[CF_ST * kQCtx + id]};
        const unsigned pxy = cstu[CF_PXY * kQCtx + id];
        const int geom = int(cstu[CF_GEOM * kQCtx + id]) & kPrimIndexMask;
        const int prim = int(cstu[CF_PRIM * kQCtx + id]);
        uint32_t* const cg = ctx_global(id);
        // Sampler window: this vertex draws dimensions dim .. dim+4 (light 2+1, BSDF 2).  The pixel's
        // ranking bytes for dimension d sit at rank[pix*8 + d] (bluenoise_*spp.cpp:14-34), so the
        // dwords covering [4*(dim/4), +12) and the 8 scrambling bytes are fetched NOW, all at once
        // (one L2 round trip instead of one per draw), and parked in this thread's LDS slots; draws
        // outside the window fall back to the byte-wise table path.
        const int wbase = (st.dim() >> 2) < 9 ? (st.dim() >> 2) : 9;  // (the device table has 64 bytes of wrap-around padding)
        unsigned* const win = (F & F_LDS_TOP) ? reinterpret_cast<unsigned*>(wave_region) + lane : reinterpret_cast<unsigned*>(lds_raw + kQOffWin) + tid;
        {
          const int pix = int(pxy & 127u) + int((pxy >> 16) & 127u) * 128;
          const uint32_t* rk = reinterpret_cast<const uint32_t*>(S.tables.rank + size_t(pix) * 8) + wbase;
          const uint32_t w0 = rk[0], w1 = rk[1], w2 = rk[2];
          const uint2 sc = *reinterpret_cast<const uint2*>(S.tables.scramble + size_t(pix) * 8);
          win[0] = w0;
          win[kStride] = w1;
          win[2 * kStride] = w2;
          win[3 * kStride] = sc.x;
          win[4 * kStride] = sc.y;
        }
        DTables T = S.tables;
        T.lds_sobol = reinterpret_cast<const uint8_t*>(lds_raw + kQOffSobol);
        T.lds_tile = win - wbase * kStride;
        T.lds_scr = win + 3 * kStride;
        T.tile_stride = kStride;
        T.win_lo = wbase * 4;
        T.win_len = (wbase * 4 + 12 <= kLdsSamplerDims ? 12 : kLdsSamplerDims - wbase * 4);
        DSampler sampler;
        sampler.px = int(pxy & 0xffffu);
        sampler.py = int(pxy >> 16);
        sampler.index = st.s_cur();
        sampler.dimension = st.dim();
        if constexpr (kBigDim)
          if (S.tables.kind != 0) sampler.dimension = int(cg[6]);
        const int pv_length = st.length();

        const DShape* shape = &V.shapes[geom];
        const DMaterial* mat = &V.materials[shape->material];
        DSurface it;
        it.p = it.n = mk3(0.0f);
        it.uv = f2{0, 0};
        {
          const f3 ph = ray_o + ray_tmax * ray_d;
          bool on_mesh = false;
          if constexpr (F & F_MESH) on_mesh = shape->kind == SHAPE_MESH;
          if (on_mesh) {
            if constexpr (F & F_EMBREE) mesh_surface_info_embree(V.rcpps, V.tri_verts, V.tri_attrs, as_int(shape->f[4]), prim, ray_o, ray_d, it);
            else mesh_surface_info(V.tri_verts, V.tri_attrs, as_int(shape->f[4]), prim, ph, it);
          } else shape_surface_info<F>(shape, ph, it);
        }
        const f3 wi = -ray_d;
        m3 l2w = coordinate_system(it.n);
        m3 w2l = transpose(l2w);
        const bool diffused = st.diffuse_length() > 0;
        const float min_roughness = diffused ? 0.6f : 0.0f;
        DBxdf bx;
        bx.kind = BX_DIFFUSE;
        bx.roughness = 0.0f;
        bx.ior = 1.0f;
        bool is_uber = false, is_sss = false, is_lobe = false;
        if constexpr (F & F_UBER) is_uber = mat->kind == MAT_UBER;
        if constexpr (F & F_UBER) is_lobe = mat->kind >= MAT_METAL;  // Metal / Glossy / Glass: one fixed lobe
        if constexpr (F & F_SSS) is_sss = mat->kind == MAT_SUBSURFACE;
        const MatParams mp = material_params<F>(mat, V.node_ops, it.p, it.n, it.uv);
        auto rng_load = [&]() -> DRng {
          return DRng{uint64_t(cg[0]) | (uint64_t(cg[1]) << 32), uint64_t(cg[2]) | (uint64_t(cg[3]) << 32)};
        };
        auto rng_store = [&](const DRng& g) {
          cg[0] = uint32_t(g.s0);
          cg[1] = uint32_t(g.s0 >> 32);
          cg[2] = uint32_t(g.s1);
          cg[3] = uint32_t(g.s1 >> 32);
        };
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
          if (st.walk() != kWalkNone) {  // second entry, after the walk: the lobe was chosen (and its draw made) the first time
            bx.kind = BX_BSSRDF;
            bx.ior = mat->ior;
          } else {
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
        }
        bx.wi = mul(w2l, wi);
        // ---- BSSRDF random walk (bxdf.cpp:329-353, :375-382): started here, stepped by stage W ----
        int beta_channel = 0;
        bool start_walk = false;
        if constexpr (F & F_SSS) {
          if (bx.kind == BX_BSSRDF) {
            float4* const cg4 = reinterpret_cast<float4*>(cg);
            if (st.walk() == kWalkNone) {
              f3 w = -wi;
              if (Refract(wi, it.n, bx.ior, w, nullptr)) {
                DRng g = rng_load();
                const int channel = int(rng_nextf(g) * 3);
                rng_store(g);
                const DRay wr = spawn_ray_raw(it.p, it.n, w);  // (later steps start AT the scattering point, with tmax = float max)
                cg4[2] = make_float4(wr.o.x, wr.o.y, wr.o.z, __int_as_float(channel));
                cg4[3] = make_float4(wr.d.x, wr.d.y, wr.d.z, wr.tmax);
                st.set_dim(sampler.dimension & 0x1ff);
                if constexpr (kBigDim) cg[6] = uint32_t(sampler.dimension);
                st.set_walk(kWalkRunning);
                if constexpr (kFork) {
                  // The channel pick was this sample's LAST draw from the pixel's RNG: whatever the walk does (it draws
                  // from the sampler), the vertex is then shaded with the BSSRDF lobe, a non-delta bounce, after which the
                  // path would seal anyway.  Seal NOW and hand the pixel's next sample on: the walk's steps and the second
                  // pass through this stage leave the pixel's chain (a chain step is T -> XC -> S instead of T -> XC -> S ->
                  // W ... -> S).
                  if (kW(fork_sealed) && !st.sealed()) {
                    st.set_sealed();
                    release = !chain_ends_at(st.s_cur() + 1);
                    if (!release) {
                      const unsigned before = atomicSub(&qctl[QC_PIXELS], 1u);
                      (void)before;
#ifdef PINE_PROFILE_SECTIONS
                      if (before == 1u && blockIdx.x < 1024) counters->wg_t[blockIdx.x][0] = wall_clock64();
#endif
                    }
                  }
                }
                cstu[CF_ST * kQCtx + id] = st.v;
                start_walk = true;
              }  // else sample_p returns nullopt: nothing changes, the vertex is shaded where it was hit
            } else if (st.walk() == kWalkExited) {
              // the walk left the shape: bc.it.p / n move to the exit point, wi becomes the reversed last walk direction
              const float4 a = cg4[2], b = cg4[3], c = cg4[4];
              beta_channel = __float_as_int(a.w) + 1;
              it.p = f3{a.x, a.y, a.z};
              it.n = f3{c.x, c.y, c.z};
              l2w = coordinate_system(it.n);
              w2l = transpose(l2w);
              bx.wi = mul(w2l, -f3{b.x, b.y, b.z});
            }  // kWalkFailed: a walk ray found no surface (sample_p returns nullopt): nothing changes
          }
        }
        SEC_MARK(1);  // S: state load, surface, frame, material
        // ---- next-event estimation (path.cpp:98-113) ----
        // The visibility test: traced here and now, or (X variants) deferred until the BSDF has been sampled -- then
        // nothing but a handful of values is live across the traversal, and a parked shadow ray has a trivial
        // continuation (clear the direct term it gated).  No draw depends on the result, so the sample streams are
        // the same either way.
        bool have_shadow = false;
        DRay shadow_ray{};
        auto shadow_test = [&](const DRay& sr) -> bool {
          if constexpr (kFlat) {
            have_shadow = true;
            shadow_ray = sr;
            return false;  // evaluated as if visible; cleared below when the ray turns out occluded
          } else {
            DRay r = sr;
            int g2, p2;
#if defined(PINE_BAKED_SCENE) && !defined(PINE_BAKED_TOP)
            return scene_traverse_baked<true, F>(r, g2);
#else
            return scene_traverse<true, F, kStride>(V, r, stack, g2, p2);
#endif
          }
        };
        f3 nee = mk3(0.0f);
        if (!start_walk && !bxdf_is_delta<F>(bx)) {
          const f2 u2 = sampler_get2d<kSM>(T, sampler);  // g++ order: get2d first (lightsampler.h:27)
          float u1 = sampler_get1d<kSM>(T, sampler);
          if constexpr (F & F_LIGHTS) {
            // general light list (light.cpp:11-84): area lights, delta lights (no MIS, path.cpp:104-106), Sky
            if (S.num_lights > 0) {
              if (S.num_lights != 1) u1 *= float(S.num_lights);
              const int index = int(u1);
              const DLight* L = &V.lights[index];
              const int lkind = L->kind;
              bool lvalid = false;
              f3 lw = mk3(0.0f), lle = mk3(0.0f);
              float ldist = 0.0f, lpdf = 0.0f;
              if (lkind == LIGHT_AREA) {
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
                lvalid = light_sample_other(L, it.p, u2, lw, ldist, lpdf, lle);
              }
              const bool ldelta = lkind == LIGHT_POINT || lkind == LIGHT_SPOT || lkind == LIGHT_DIRECTIONAL;
              if (lvalid) {
                const float ls_pdf = S.num_lights != 1 ? lpdf / float(S.num_lights) : lpdf;
                shadow_count++;
                const DRay sr = spawn_ray(it.p, it.n, lw, ldist);
                if (!shadow_test(sr)) {
                  bx.albedo = mp.albedo;
                  bx.albedo_over_pi = mp.albedo_over_pi;
                  const float cosine = absdot(lw, it.n);
                  const f3 wo = mul(w2l, lw);
                  const f3 f = bxdf_f<F>(bx, wo);
                  if (ldelta) {
                    nee = mk3(0.0f) + lle * mk3(1.0f) * cosine * f / ls_pdf;
                  } else {
                    const float mis = balance_heuristic(ls_pdf, bxdf_pdf<F>(bx, wo));
                    nee = mk3(0.0f) + lle * mk3(1.0f) * cosine * f / ls_pdf * mis;
                  }
                }
              }
            }
          } else if (S.num_lights > 0) {
            if (S.num_lights != 1) u1 *= float(S.num_lights);
            const int index = int(u1);
            const DShape* lshape = &V.shapes[V.lights[index].geom];  // (no other light kinds in this variant)
            DShapeSample gs;
            if (shape_sample<F>(lshape, V.tri_verts, it.p, u2, u1 - float(index), gs)) {
              const DMaterial* lmat = &V.materials[lshape->material];
              if (!is_zero(material_le(lmat, gs.n, -gs.w))) {
                const float ls_pdf = S.num_lights != 1 ? gs.pdf / float(S.num_lights) : gs.pdf;
                shadow_count++;
                const DRay sr = spawn_ray(it.p, it.n, gs.w, gs.distance);
                SEC_MARK(2);  // S: sampler draws + light sampling
                const bool occluded = shadow_test(sr);
                SEC_MARK(3);  // S: shadow traversal (the not-X variants)
                if (!occluded) {
                  const f3 le = ld3(lmat->color);
                  bx.albedo = mp.albedo;
                  bx.albedo_over_pi = mp.albedo_over_pi;
                  const float cosine = absdot(gs.w, it.n);
                  const f3 wo = mul(w2l, gs.w);
                  const f3 f = bxdf_f<F>(bx, wo);
                  const float mis = balance_heuristic(ls_pdf, bxdf_pdf<F>(bx, wo));
                  nee = mk3(0.0f) + le * mk3(1.0f) * cosine * f / ls_pdf * mis;
                }
              }
            }
          }
        }
        SEC_MARK(4);  // S: NEE evaluation
        // ---- BSDF sampling + continuation (path.cpp:114-120) ----
        bx.albedo = mp.albedo;
        bx.albedo_over_pi = mp.albedo_over_pi;
        DBsdfSample bs;
        bs.f = mk3(0.0f);
        bs.wo = mk3(0.0f);
        bs.pdf = 1.0f;
        bs.is_delta = false;
        const bool continues = !start_walk && bxdf_sample<F, kSM>(bx, T, sampler, bs);
        const f3 wo_world = mul(l2w, bs.wo);
        const float cosine = absdot(wo_world, it.n);
        // ---- X variants: the shadow ray is traced by stage XS; the fold entry / result below hold the direct term as if
        // visible, and XS clears it when the ray turns out occluded ----
#if defined(PINE_BAKED_TOP) && !defined(PINE_TOP_SHADOW_IN_XS)
        // Top level baked (pine_specialize.h): the shadow ray meets the top-level primitives HERE, where the wave's lanes all have
        // one -- an occluded or a clear ray never sees the traversal stages, and stage XS gets the others at the mesh's root.
        if (have_shadow) {
          DRay sr = shadow_ray;
          BakedMesh bm{false, 0.0f, -1};
          int g = -1;
          if (scene_traverse_baked_top<true, F, 0>(sr, g, bm)) {
            nee = mk3(0.0f);  // occluded
            have_shadow = false;
          } else if (!bm.reached) {
            have_shadow = false;  // visible
          }
        }
#endif
        const bool parked_shadow = kX && have_shadow;
        bool closest_resolved = false;  // (top level baked: the new ray ended in the top-level code)
        (void)closest_resolved;
        if constexpr (kFlat && !kX) {
          // ... or (F_LDS_TOP variants without traversal stages) here, now that the BSDF has been sampled and little is live
          if (__ballot(have_shadow) != 0) {
            TravState sts;
            trav_begin(V, sts);
            if (!have_shadow) sts.done = 1;
            const DRayOct soct = make_oct(shadow_ray);
#ifdef PINE_TRIP_STATS
            trav_trips<true, F, kStride>(V, shadow_ray, soct, sts, stack, 0, 1 << 30, &stat_shadow_trips);
#else
            trav_trips<true, F, kStride>(V, shadow_ray, soct, sts, stack, 0, 1 << 30);
#endif
            if (have_shadow && sts.hit_geom >= 0) nee = mk3(0.0f);  // occluded
          }
          SEC_MARK(3);  // S: shadow traversal
        }
        if (start_walk) {
          to_walk = true;  // (state stored above; the context continues in stage W)
        } else if (continues) {
          float4* q = fold_entry(id, pv_length);
          q[0] = make_float4(nee.x, nee.y, nee.z, bs.f.x);
          q[1] = make_float4(bs.f.y, bs.f.z, cosine / bs.pdf, bs.pdf);
          if (kVlog && W.vertex_log) {
            float4* r = reinterpret_cast<float4*>(vlog(pxy, st.s_cur(), pv_length));
            r[0] = make_float4(3.0f, float(pv_length), nee.x, nee.y);
            r[1] = make_float4(nee.z, bs.f.x, bs.f.y, bs.f.z);
            r[2] = make_float4(cosine, bs.pdf, bs.is_delta ? 1.0f : 0.0f, 0.0f);
          }
          if constexpr (F & F_SSS) {  // this level's BSSRDF beta channel (0 = none): 2 bits in the context's global record
            const int word = pv_length >> 4, sh = 2 * (pv_length & 15);
            cg[4 + word] = (cg[4 + word] & ~(3u << sh)) | (unsigned(beta_channel) << sh);
          }
          const DRay nr = spawn_ray(it.p, it.n, wo_world, kFloatMax);
          st.set_dim(sampler.dimension & 0x1ff);
          if constexpr (kBigDim) cg[6] = uint32_t(sampler.dimension);
          st.next_vertex(bs.is_delta);
          if constexpr (kFork) {
            // sealed: after its first non-delta bounce a path makes no RNG draw any more (a Subsurface vertex is then
            // plain diffuse, above), so the pixel's RNG state is final for this sample and the NEXT sample can start
            if (kW(fork_sealed) && !st.sealed() && st.diffuse_length() > 0) {
              st.set_sealed();
              release = !chain_ends_at(st.s_cur() + 1);
              if (!release) {  // the pixel's last sample is under way: no longer "in flight" for the intake limit
                const unsigned before = atomicSub(&qctl[QC_PIXELS], 1u);
                (void)before;
#ifdef PINE_PROFILE_SECTIONS
                if (before == 1u && blockIdx.x < 1024) counters->wg_t[blockIdx.x][0] = wall_clock64();  // (the workgroup's last whole-pixel item: its last sample sealed)
#endif
              }
            }
          }
          if (kX && parked_shadow) {
            // the fold entry above holds the direct term as if visible; stage XS clears it if the ray is occluded, then
            // traces the new ray, which waits in the context
            cstf[CF_OX * kQCtx + id] = nr.o.x;
            cstf[CF_OY * kQCtx + id] = nr.o.y;
            cstf[CF_OZ * kQCtx + id] = nr.o.z;
            cstf[CF_DX * kQCtx + id] = nr.d.x;
            cstf[CF_DY * kQCtx + id] = nr.d.y;
            cstf[CF_DZ * kQCtx + id] = nr.d.z;
            cstf[CF_TMAX * kQCtx + id] = nr.tmax;
            cstu[CF_ST * kQCtx + id] = st.v;
#if defined(PINE_BAKED_TOP) && !defined(PINE_TOP_CLOSEST_IN_XC)
            closest_resolved = top_first_pass(id, nr.o, nr.d, nr.tmax);  // (stage XS classifies the context itself then)
#endif
          } else {
            extend(id, nr.o, nr.d, nr.tmax, st, to_shade, to_term, to_xc);
          }
        } else {
          // no continuation: the vertex resolves with lo = nee (path.cpp:121); stage T folds it
          f3 beta = mk3(1.0f);
          if (beta_channel) {
            beta = mk3(0.0f);
            set(beta, beta_channel - 1, 3.0f);
          }
          const f3 Lo = mk3(0.0f) + vmin(mk3(1.0f) * beta * nee, mk3(8.0f));
          if (kVlog && W.vertex_log) {
            float4* r = reinterpret_cast<float4*>(vlog(pxy, st.s_cur(), pv_length));
            r[0] = make_float4(3.0f, float(pv_length), nee.x, nee.y);
            r[1] = make_float4(nee.z, 0.0f, 0.0f, 0.0f);
            r[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          }
          cstf[CF_DX * kQCtx + id] = Lo.x;
          cstf[CF_DY * kQCtx + id] = Lo.y;
          cstf[CF_DZ * kQCtx + id] = Lo.z;
          cstu[CF_GEOM * kQCtx + id] = unsigned(-2);
          to_term = !(kX && parked_shadow);
        }
        if constexpr (kX) {
          if (parked_shadow) {
            uint32_t* rec = trav_record(id);
            float4* r4 = reinterpret_cast<float4*>(rec);
            r4[0] = make_float4(shadow_ray.o.x, shadow_ray.o.y, shadow_ray.o.z, shadow_ray.d.x);
            r4[1] = make_float4(shadow_ray.d.y, shadow_ray.d.z, shadow_ray.tmax,
                                __uint_as_float((continues ? 0u : kTravTerminalAfterShadow) | (closest_resolved ? kTravClosestResolved : 0u)));
            to_xs = true;
          }
        }
      }
      SEC_MARK(9);
      unsigned long long rel_mask = 0;
      unsigned rel_pos = 0;
      if constexpr (kFork) {
        rel_mask = __ballot(release);
        if (rel_mask != 0) {
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(&qctl[QC_TAIL + QS_K], unsigned(__popcll(rel_mask)));
          base = __builtin_amdgcn_readfirstlane(base);
          if (release) {
            rel_pos = (base + unsigned(__popcll(rel_mask & ((1ull << lane) - 1ull)))) & (kQRing - 1);
            const uint4 g = *reinterpret_cast<const uint4*>(ctx_global(id));  // the pixel's RNG state: final for this sample
            const PackedState st{cstu[CF_ST * kQCtx + id]};
            uint4* tok = reinterpret_cast<uint4*>(token_slot(rel_pos));
            tok[0] = g;
            tok[1] = make_uint4(cstu[CF_PXY * kQCtx + id], cstu[CF_SBASE * kQCtx + id], unsigned(st.s_cur() + 1), 0u);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      push2(to_shade, to_term, id);
      if constexpr (F & F_SSS) push_w(to_walk, id);
      if constexpr (kX) {
        push_q(QS_XS, to_xs, id);
        push_q(QS_XC, to_xc, id);
      }
      if constexpr (kFork) {
        if (rel_mask != 0) {
          if (release) __atomic_store_n(&ring[QS_K * kQRing + rel_pos], 1u, __ATOMIC_RELAXED);
          // contexts with nothing to do wait in F: wake one per token (they enter stage T as empty contexts and find
          // the tokens there)
          wake_free(unsigned(__popcll(rel_mask)));
        }
      }
    } else if (kX && (stage == QS_XS || stage == QS_XC)) {
      // ================= traversal stages: XS any-hit (shadow rays), XC closest hit (pine_trav.h) =================
      if constexpr (kX) {
        if (stage == QS_XS) trav_stage(std::true_type{}, valid ? id : -1);
        else trav_stage(std::false_type{}, valid ? id : -1);
      }
    } else if ((F & F_SSS) != 0 && stage == QS_W) {
      // ================= one step of a BSSRDF random walk (bxdf.cpp:340-351) =================
      bool to_walk = false, to_shade = false;
      if constexpr (F & F_SSS) {
        if (valid) {
          REGION(14);  // (lanes of a stage-W pass that hold a context)
          float4* const cg4 = reinterpret_cast<float4*>(ctx_global(id));
          const float4 a = cg4[2], b = cg4[3];
          PackedState st{cstu[CF_ST * kQCtx + id]};
          const unsigned pxy = cstu[CF_PXY * kQCtx + id];
          const DShape* shape = &V.shapes[int(cstu[CF_GEOM * kQCtx + id]) & kPrimIndexMask];
          const DMaterial* mat = &V.materials[shape->material];
          const int channel = __float_as_int(a.w);
          DRay wr{f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, 0.0f, b.w};
          walk_count++;
          DSampler sampler;
          sampler.px = int(pxy & 0xffffu);
          sampler.py = int(pxy >> 16);
          sampler.index = st.s_cur();
          sampler.dimension = st.dim();
          if constexpr (kBigDim)
            if (S.tables.kind != 0) sampler.dimension = int(ctx_global(id)[6]);
          bool hh;
          int wprim = 0;
          bool walk_mesh = false;
          if constexpr (F & F_MESH) walk_mesh = shape->kind == SHAPE_MESH;
          if (walk_mesh) {
            const DRayOct oct = make_oct(wr);
            hh = mesh_traverse<false, kStride, F>(V, V.bvhs[as_int(shape->f[2])], wr, oct, stack, 0, wprim);
          } else {
            hh = shape_intersect<F>(shape, wr);
          }
          SEC_MARK(12);  // W: closest hit inside the shape
          if (!hh) {  // sample_p returns nullopt
            st.set_walk(kWalkFailed);
            to_shade = true;
          } else {
            const float t = -plog(1 - sampler_get1d<kSM & kSmSobol>(S.tables, sampler)) * (1 / mat->sigma_s[channel]);
            if (wr.tmax < t) {
              // leaves the shape here: Shape::intersect filled it.p / it.n for meshes only
              DSurface sit;
              sit.p = sit.n = mk3(0.0f);
              if (walk_mesh) mesh_surface_info(V.tri_verts, V.tri_attrs, as_int(shape->f[4]), wprim, ray_at(wr, wr.tmax), sit);
              cg4[2] = make_float4(sit.p.x, sit.p.y, sit.p.z, a.w);
              cg4[4] = make_float4(sit.n.x, sit.n.y, sit.n.z, 0.0f);
              st.set_walk(kWalkExited);
              to_shade = true;
            } else {
              const f3 p = ray_at(wr, t);
              const f3 w = uniform_sphere(sampler_get2d<kSM & kSmSobol>(S.tables, sampler));
              cg4[2] = make_float4(p.x, p.y, p.z, a.w);
              cg4[3] = make_float4(w.x, w.y, w.z, kFloatMax);
              to_walk = true;
            }
          }
          st.set_dim(sampler.dimension & 0x1ff);
          if constexpr (kBigDim) ctx_global(id)[6] = uint32_t(sampler.dimension);
          cstu[CF_ST * kQCtx + id] = st.v;
        }
      }
      SEC_MARK(13);  // W: free flight, scatter / exit, state store
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      push2(to_shade, false, id);
      push_w(to_walk, id);
    } else {
      // ================= terminal: result, backward fold, store, next sample / item =================
      bool need_item = false, have_path = false;
      bool to_shade2 = false, to_term2 = false, to_xc2 = false;
      bool to_free = false;  // (kFork) nothing to do for this context now: it waits in F
      bool fresh_rng = false;  // the context took a new work item this round: its RNG state is in item_rng, not yet in memory
      bool free_item = false;  // (kFork, tile classes) ... and it is an item of the independent class
      bool wake_more = false;  // (kFork, tile classes) independent items are at hand: wake contexts that wait in F
      DRng item_rng{0, 0};
      unsigned pxy = 0, sample_base = 0;
      int s_next = 0;
      uint4 rng_words = make_uint4(0, 0, 0, 0);
      if (valid) {
        REGION(13);  // (lanes of a stage-T pass that hold a context)
        // the pixel's RNG state is needed only for the next camera sample, but its L2 round trip
        // starts here so that it overlaps the fold loop's
        rng_words = *reinterpret_cast<const uint4*>(ctx_global(id));
        const PackedState st{cstu[CF_ST * kQCtx + id]};
        if (st.v == kStFresh) {
          need_item = true;
        } else {
          const int pv_length = st.length();
          const int geom = int(cstu[CF_GEOM * kQCtx + id]);
          f3 Lo = mk3(0.0f);
          bool lp_valid = false;
          float lp = 0.0f;
          if (geom == -2) {
            Lo = f3{cstf[CF_DX * kQCtx + id], cstf[CF_DY * kQCtx + id], cstf[CF_DZ * kQCtx + id]};
          } else if (geom == -1) {  // miss: the environment light, if any (path.cpp:75-81)
            if constexpr (F & F_LIGHTS)
              if (S.env_light >= 0) {
                const f3 ray_d{cstf[CF_DX * kQCtx + id], cstf[CF_DY * kQCtx + id], cstf[CF_DZ * kQCtx + id]};
                Lo = mk3(1.0f) * sky_color_of(ld3(V.lights[S.env_light].color), ray_d);
                if (!st.is_delta()) {
                  lp_valid = true;
                  lp = 1 / (4 * kPi);  // Sky::pdf -- not divided by the light count
                }
              }
          } else if (geom >= 0 && (geom & kPrimEmissiveBit) != 0) {
            const DShape* shape = &V.shapes[geom & kPrimIndexMask];
            const DMaterial* mat = &V.materials[shape->material];
            {  // path.cpp:83-87
              const f3 ray_o{cstf[CF_OX * kQCtx + id], cstf[CF_OY * kQCtx + id], cstf[CF_OZ * kQCtx + id]};
              const f3 ray_d{cstf[CF_DX * kQCtx + id], cstf[CF_DY * kQCtx + id], cstf[CF_DZ * kQCtx + id]};
              const float ray_tmax = cstf[CF_TMAX * kQCtx + id];
              DSurface it;
              it.p = it.n = mk3(0.0f);
              it.uv = f2{0, 0};
              const f3 ph = ray_o + ray_tmax * ray_d;
              bool on_mesh = false;
              if constexpr (F & F_MESH) on_mesh = shape->kind == SHAPE_MESH;
              if (on_mesh) {
                if constexpr (F & F_EMBREE) mesh_surface_info_embree(V.rcpps, V.tri_verts, V.tri_attrs, as_int(shape->f[4]), int(cstu[CF_PRIM * kQCtx + id]), ray_o, ray_d, it);
                else mesh_surface_info(V.tri_verts, V.tri_attrs, as_int(shape->f[4]), int(cstu[CF_PRIM * kQCtx + id]), ph, it);
              } else shape_surface_info<F>(shape, ph, it);
              Lo = mk3(1.0f) * material_le(mat, it.n, -ray_d);
              if (!st.is_delta()) {
                lp_valid = true;
                const DRay ray{ray_o, ray_d, 0.0f, ray_tmax};
                lp = shape_pdf<F>(shape, ray, it.n);
                if (S.num_lights != 1) lp = lp / float(size_t(S.num_lights));
              }
            }
          }
          // backward fold (path.cpp:114-121)
          f3 Li = Lo;
          unsigned long long beta_flags = 0;  // 2 bits per level: the BSSRDF beta channel (bxdf.cpp:335), dead code without F_SSS
          if constexpr (F & F_SSS) {
            const uint2 bf = *reinterpret_cast<const uint2*>(ctx_global(id) + 4);
            beta_flags = (unsigned long long)bf.x | ((unsigned long long)bf.y << 32);
          }
          if (kVlog && W.vertex_log) {  // the terminal invocation's record (a shaded vertex without continuation: stage S wrote its terms)
            float* r = vlog(cstu[CF_PXY * kQCtx + id], st.s_cur(), pv_length);
            if (geom != -2) {
              r[0] = geom == -1 ? 0.0f : (geom & kPrimEmissiveBit) ? 1.0f : 2.0f;
              r[1] = float(pv_length);
            }
            r[12] = lp_valid ? lp : -1.0f;
            r[13] = Lo.x, r[14] = Lo.y, r[15] = Lo.z;
          }
          auto fold_step = [&](const float4& a, const float4& b, int level) {
            const f3 e_nee{a.x, a.y, a.z};
            const f3 e_f{a.w, b.x, b.y};
            const float e_cp = b.z, e_pdf = b.w;
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
            if (kVlog && W.vertex_log) {
              float* r = vlog(cstu[CF_PXY * kQCtx + id], st.s_cur(), level);
              r[11] = mis;
              r[12] = -1.0f;
              r[13] = Li.x, r[14] = Li.y, r[15] = Li.z;
            }
          };
          int level = pv_length - 1;
          for (; level >= 0; level--) {
            const float4* q = fold_entry(id, level);
            fold_step(q[0], q[1], level);
          }
          const int s_now = st.s_cur();
          sample_base = cstu[CF_SBASE * kQCtx + id];
          pxy = cstu[CF_PXY * kQCtx + id];
          samples[size_t(sample_base) + size_t(s_now) * 64u] = make_float4(Li.x, Li.y, Li.z, float(pv_length + 1));
          s_next = s_now + 1;
          // (sealed: the pixel's next sample went out as a token -- or, with tile classes, the path is a one-sample item
          // of the independent class, which never owned its pixel's chain)
          // (an item is samples_per_item consecutive samples: a power of two that divides spp, or -- SobolSampler / HaltonSampler
          //  with another count -- the whole pixel)
          const int spi = W.samples_per_item;
          const bool item_done = kFork ? chain_ends_at(s_next) : (spi & (spi - 1)) == 0 ? (s_next & (spi - 1)) == 0 : s_next == spi;
          if ((kFork && st.sealed()) || item_done) need_item = true;
          else have_path = true;
          if constexpr (kFork)
            if (!st.sealed() && chain_ends_at(s_next)) {
              const unsigned before = atomicSub(&qctl[QC_PIXELS], 1u);
              (void)before;
#ifdef PINE_PROFILE_SECTIONS
              if (before == 1u && blockIdx.x < 1024) counters->wg_t[blockIdx.x][0] = wall_clock64();
#endif
            }
        }
      }
      SEC_MARK(7);  // T: terminal result + backward fold + sample store
      // ---- hand out work items to the contexts that need one (block pool, refilled from the global queue) ----
      for (unsigned rounds = 0;; rounds++) {
        const unsigned long long m = __ballot(need_item);
        if (m == 0) break;
        if (rounds > 4096u) {
          if (lane == 0) bail(4, unsigned(m), unsigned(m >> 32));
          need_item = false;
          break;
        }
        const unsigned want_all = __popcll(m);
        unsigned want = want_all;
        unsigned long long base = 0;
        unsigned got = 0;
        unsigned kbase = 0, kgot = 0;  // (kFork) sample tokens taken: ring positions [kbase, kbase + kgot)
        unsigned want_free = 0;        // (kFork) items wanted beyond the tokens, before the in-flight limit
        unsigned free_items_left = 0;  // (kFork, tile classes) the workgroup's block still has independent items after this round
        if (lane == 0) {
          unsigned tries = 0;
          while (atomicCAS(&qctl[QC_LOCK], 0u, 1u) != 0u) {
            if (++tries > kQSpinLimit) {
              bail(3, 0, 0);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if constexpr (kFork) {
            // waiting sample tokens first (only holders of the lock move K's head); NEW pixels only when none waits:
            // then every pixel in flight has its token in a running path, so there are never more tokens than contexts
            kbase = lds_load(&qctl[QC_HEAD + QS_K]);
            const unsigned kavail = lds_load(&qctl[QC_TAIL + QS_K]) - kbase;
            kgot = want < kavail ? want : kavail;
            __atomic_store_n(&qctl[QC_HEAD + QS_K], kbase + kgot, __ATOMIC_RELAXED);
            want -= kgot;
            want_free = want;
            // pixels in flight per workgroup are bounded (kW(max_pixels)): each one's samples then follow one another
            // quickly, and little is left half-done -- unevenly, workgroup by workgroup -- when the work-item pool runs dry
            const unsigned inflight = lds_load(&qctl[QC_PIXELS]);
            const unsigned room = inflight < unsigned(kW(max_pixels)) ? unsigned(kW(max_pixels)) - inflight : 0u;
            if (want > room) want = room;
          }
          unsigned long long nx = (unsigned long long)lds_load(&qctl[QC_PNEXT]) | ((unsigned long long)lds_load(&qctl[QC_PNEXT + 1]) << 32);
          unsigned long long en = (unsigned long long)lds_load(&qctl[QC_PEND]) | ((unsigned long long)lds_load(&qctl[QC_PEND + 1]) << 32);
          if (want != 0u && nx == en && lds_load(&qctl[QC_EXHAUSTED]) == 0u) {
            const unsigned long long b = atomicAdd(&counters->next_item, (unsigned long long)kW(pool_items));
            if (b >= kW(total_items)) {
#ifdef PINE_PROFILE_SECTIONS
              atomicCAS(&counters->t_pool_dry, 0ull, wall_clock64());
              if (blockIdx.x < 1024) counters->wg_t[blockIdx.x][1] = wall_clock64();
#endif
              __atomic_store_n(&qctl[QC_EXHAUSTED], 1u, __ATOMIC_RELAXED);
            } else {
              post_progress(W, b, 9);
#ifdef PINE_PROFILE_SECTIONS
              if (blockIdx.x < 1024) {
                if (b < (unsigned long long)W.serial_tiles * 64ull) atomicAdd(&counters->wg_t[blockIdx.x][3], 64ull);
              }
#endif
              nx = b;
              en = b + kW(pool_items) < kW(total_items) ? b + kW(pool_items) : kW(total_items);
            }
          }
          const unsigned long long avail = en - nx;
          if constexpr (kFork) {
            // tile classes: the block [nx, en) holds independent one-sample items (no pixel is taken into flight, no limit)
            // when it lies behind the whole-pixel items -- blocks are 64 items and the class boundary a multiple of 64
            if (W.serial_tiles > 0 && nx >= (unsigned long long)W.serial_tiles * 64ull) want = want_free;
          }
          got = want < avail ? want : unsigned(avail);
          if constexpr (kFork) {
            if (!(W.serial_tiles > 0 && nx >= (unsigned long long)W.serial_tiles * 64ull)) atomicAdd(&qctl[QC_PIXELS], got);
            else if (avail > got) free_items_left = 1u;  // independent items at hand: contexts waiting in F can have them
          }
          base = nx;
          nx += got;
          __atomic_store_n(&qctl[QC_PNEXT], unsigned(nx), __ATOMIC_RELAXED);
          __atomic_store_n(&qctl[QC_PNEXT + 1], unsigned(nx >> 32), __ATOMIC_RELAXED);
          __atomic_store_n(&qctl[QC_PEND], unsigned(en), __ATOMIC_RELAXED);
          __atomic_store_n(&qctl[QC_PEND + 1], unsigned(en >> 32), __ATOMIC_RELAXED);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __atomic_store_n(&qctl[QC_LOCK], 0u, __ATOMIC_RELAXED);
        }
        base = __shfl(base, 0);
        got = __shfl(got, 0);
        if constexpr (kFork) {
          kbase = __shfl(kbase, 0);
          kgot = __shfl(kgot, 0);
          wake_more |= __shfl(free_items_left, 0) != 0u;
        }
        if (got == 0 && kgot == 0) {
          // global queue exhausted (and no token waits): these contexts retire -- plain variants: they are simply not
          // pushed anywhere; kFork: they wait in F until a path releases a token
          if constexpr (kFork) to_free = need_item;
          need_item = false;
          break;
        }
        unsigned rank = __popcll(m & ((1ull << lane) - 1ull));
        if (kFork && need_item && rank < kgot) {
          // a waiting sample of a pixel in flight: its token holds the pixel's RNG state as the previous sample left it
          const unsigned pos = (kbase + rank) & (kQRing - 1);
          unsigned* flag = &ring[QS_K * kQRing + pos];
          unsigned tries = 0;
          while (atomicExch(flag, 0u) == 0u) {
            if (++tries > kQSpinLimit) {
              bail(8, pos, kbase);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          const uint4* tok = reinterpret_cast<const uint4*>(token_slot(pos));
          const uint4 g = tok[0], b = tok[1];
          need_item = false;
          have_path = true;
          pxy = b.x;
          sample_base = b.y;
          s_next = int(b.z);
          item_rng = DRng{uint64_t(g.x) | (uint64_t(g.y) << 32), uint64_t(g.z) | (uint64_t(g.w) << 32)};
          fresh_rng = true;
        }
        rank -= kgot;  // (wraps for the lanes served above: they no longer need an item)
        if (need_item && rank < got) {
          const unsigned long long item = base + rank;
          WorkParams Wl;  // (the work decomposition: cold, read from the kernel arguments here)
          kload(kWOffset, &Wl);
          const ItemInfo itf = decode_item(Wl, kS(cam.W), kS(cam.H), kS(spp), item);
          if (itf.valid) {  // else: pixel outside the film (border tile): take another item next round
            need_item = false;
            have_path = true;
            pxy = unsigned(itf.px) | (unsigned(itf.py) << 16);
            sample_base = unsigned(itf.sample_base);
            s_next = itf.chunk * W.samples_per_item;
            DRng g;
            if (itf.serial || W.items_per_pixel == 1) {
              g = rng_seed(hash_pixel(itf.px, itf.py, 0));
            } else {
              const ulonglong2 c = ckpt[itf.ckpt_index];
              g = DRng{c.x, c.y};
            }
            item_rng = g;
            fresh_rng = true;
            if constexpr (kFork) free_item = W.serial_tiles > 0 && !itf.serial;
          } else {
            if constexpr (kFork)
              if (!(W.serial_tiles > 0 && !itf.serial)) atomicSub(&qctl[QC_PIXELS], 1u);  // (counted when claimed)
          }
        }
      }
      if constexpr (kFork)
        if (to_free) cstu[CF_ST * kQCtx + id] = kStFresh;  // (woken through stage T as an empty context)
      SEC_MARK(8);  // T: work-item hand-out
      // ---- start the next camera sample (path.cpp:34-36) ----
      if (valid && have_path) {
        uint4* cg4 = reinterpret_cast<uint4*>(ctx_global(id));
        DRng g = item_rng;
        if (!fresh_rng)
          g = DRng{uint64_t(rng_words.x) | (uint64_t(rng_words.y) << 32), uint64_t(rng_words.z) | (uint64_t(rng_words.w) << 32)};
        const float lx = rng_nextf(g);  // g++ argument order: lens sample first, then pixel jitter
        const float ly = rng_nextf(g);
        const float jx = rng_nextf(g);
        const float jy = rng_nextf(g);
        cg4[0] = make_uint4(uint32_t(g.s0), uint32_t(g.s0 >> 32), uint32_t(g.s1), uint32_t(g.s1 >> 32));
        if constexpr (kBigDim) ctx_global(id)[6] = S.tables.kind == 2 ? 2u : 0u;  // start_next_sample: dimension = 0 (HaltonSampler: 2)
        const int px = int(pxy & 0xffffu), py = int(pxy >> 16);
        const DCamera cam = kS(cam);
        const f2 pf{(float(px) + jx) / float(cam.W), (float(py) + jy) / float(cam.H)};
        const DRay r = camera_gen_ray(cam, pf, f2{lx, ly});
        PackedState st{0};
        st.start_sample(s_next);
        if constexpr (kFork)
          if (free_item) st.set_sealed();  // tile classes: a one-sample item of the independent class never owns its pixel's chain
        if constexpr (F & F_SOBOL)
          if (S.tables.kind == 2) st.set_dim(2);  // HaltonSampler::start_pixel / start_next_sample: dimension = 2
        cstu[CF_PXY * kQCtx + id] = pxy;
        cstu[CF_SBASE * kQCtx + id] = sample_base;
        extend(id, r.o, r.d, r.tmax, st, to_shade2, to_term2, to_xc2);
      }
      SEC_MARK(9);  // state store + classification
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      push2(to_shade2, to_term2, id);
      if constexpr (kX) push_q(QS_XC, to_xc2, id);
      if constexpr (kFork) {
        push_q(QS_F, to_free, id);
        // tile classes: contexts that went to wait while the workgroup's whole-pixel items were at their in-flight limit
        // are woken by token releases only; once independent items are being handed out they can all work again
        if (wake_more) wake_free(64u);
      }
    }
    if (lane == 0) atomicSub(&qctl[QC_BUSY], 1u);
  }

#ifdef PINE_PROFILE_SECTIONS
  if (tid == 0) atomicMax(&counters->t_end, wall_clock64());
  if (tid == 0 && blockIdx.x < 1024) counters->wg_t[blockIdx.x][2] = wall_clock64();
#endif
  SEC_FLUSH();
  unsigned long long sc = shadow_count;
  for (int off = 32; off > 0; off >>= 1) sc += __shfl_down(sc, off);
  if (lane == 0) atomicAdd(&counters->shadow_rays, sc);
  if constexpr (F & F_SSS) {
    unsigned long long wc = walk_count;
    for (int off = 32; off > 0; off >>= 1) wc += __shfl_down(wc, off);
    if (lane == 0) atomicAdd(&counters->walk_steps, wc);
  }
}

template <unsigned F, int CTX = PINE_QCTX>
__global__ void __launch_bounds__(kQBlock, kQBlock / 256)
path_queue_kernel(DeviceScene S, WorkParams W, const ulonglong2* __restrict__ ckpt, float4* __restrict__ samples,
                  float* __restrict__ fold, uint32_t* __restrict__ ctxg, Counters* __restrict__ counters) {
  path_queue_body<F, CTX>(S, W, ckpt, samples, fold, ctxg, counters);
}

}  // namespace pine_gpu
