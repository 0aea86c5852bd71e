// pine_amd/csrc/pine_bvh_build_device.h -- the level-synchronous BVH build of pine_bvh_build.h on the GPU (included by
// pine_kernels.hip).  Same schedule, same arithmetic, same tree: three kernels per level --
//   decide : one wave per open range: centroid bounds (wave min/max), 3 x 16 bucket boxes and counts (LDS atomics on
//            order-preserving integer images of the floats), the SAH sweep by lane 0 (build_decide, shared with the host);
//   scan   : one workgroup: exclusive prefix sum of the "splits" flags -> node index and next-level task slot of every
//            range (nodes must be numbered in task order, not in the order waves finish);
//   split  : one workgroup per splitting range: the predicate per primitive, the RESULT of the reference's Lomuto swap sequence
//            computed in parallel (prefix sum + pointer jumping, see the kernel), the move, the two child boxes, the node,
//            the two new ranges.
// Every reduction is a min / max / count, so the result does not depend on how lanes share the work.
#pragma once
#include "pine_bvh_build.h"

namespace pine_gpu {

constexpr int kBuildWave = 64;
constexpr int kBuildLdsPrims = 6800;  // a range up to this size partitions in LDS (permutation 4 B + chain pointer 4 B + predicate 1 B each)

__device__ __forceinline__ int f2ord(float f) {  // float -> int whose signed order is the float order (no NaNs here)
  const int b = __float_as_int(f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float ord2f(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); }
__device__ __forceinline__ float wave_min(float v) {
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

struct BuildDecision {  // what `decide` leaves for `scan` and `split`
  int splits;           // 1: the range becomes a node; 0: a leaf
  int axis, bucket;
  float clo[3], chi[3];
};

// One wave per task.
__global__ void __launch_bounds__(kBuildWave) bvh_decide_kernel(const BuildPrim* __restrict__ prims, const BuildTask* __restrict__ tasks, int ntasks,
                                                                BuildDecision* __restrict__ dec) {
  __shared__ int s_count[3][kBuildBuckets];
  __shared__ int s_lo[3][kBuildBuckets][3], s_hi[3][kBuildBuckets][3];
  const int t = blockIdx.x;
  if (t >= ntasks) return;
  const int lane = threadIdx.x;
  const BuildTask task = tasks[t];
  const int n = task.end - task.begin;
  const BuildPrim* P = prims + task.begin;
  if (n == 1) {
    if (lane == 0) dec[t].splits = 0;
    return;
  }
  float clo[3] = {kFloatMax, kFloatMax, kFloatMax}, chi[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
  for (int i = lane; i < n; i += kBuildWave)
    for (int k = 0; k < 3; k++) {
      const float c = build_centroid(P[i], k);
      clo[k] = c < clo[k] ? c : clo[k];
      chi[k] = c > chi[k] ? c : chi[k];
    }
  for (int k = 0; k < 3; k++) clo[k] = wave_min(clo[k]), chi[k] = wave_max(chi[k]);
  for (int j = lane; j < 3 * kBuildBuckets; j += kBuildWave) {
    (&s_count[0][0])[j] = 0;
    for (int k = 0; k < 3; k++) (&s_lo[0][0][0])[j * 3 + k] = f2ord(kFloatMax), (&s_hi[0][0][0])[j * 3 + k] = f2ord(-kFloatMax);
  }
  __syncthreads();
  for (int a = 0; a < 3; a++) {
    if (chi[a] <= clo[a]) continue;
    for (int i = lane; i < n; i += kBuildWave) {
      const int j = build_bucket_of(build_centroid(P[i], a), clo[a], chi[a]);
      atomicAdd(&s_count[a][j], 1);
      for (int k = 0; k < 3; k++) {
        atomicMin(&s_lo[a][j][k], f2ord(P[i].lo[k]));
        atomicMax(&s_hi[a][j][k], f2ord(P[i].hi[k]));
      }
    }
  }
  __syncthreads();
  if (lane == 0) {
    BucketStats B;
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < kBuildBuckets; j++) {
        B.count[a][j] = s_count[a][j];
        for (int k = 0; k < 3; k++) B.lo[a][j][k] = ord2f(s_lo[a][j][k]), B.hi[a][j][k] = ord2f(s_hi[a][j][k]);
      }
    SplitDecision d;
    build_decide(B, n, build_area(task.blo, task.bhi), clo, chi, d);
    BuildDecision o;
    o.splits = d.leaf ? 0 : 1;
    o.axis = d.axis, o.bucket = d.bucket;
    for (int k = 0; k < 3; k++) o.clo[k] = clo[k], o.chi[k] = chi[k];
    dec[t] = o;
  }
}

// One workgroup: rank[t] = number of splitting tasks before t; totals -> counts[0] (nodes so far, updated), counts[1] (tasks of
// the next level).  Leaves fill their parent's child slot here (they need nothing else).
__global__ void __launch_bounds__(1024) bvh_scan_kernel(const BuildTask* __restrict__ tasks, int ntasks, const BuildDecision* __restrict__ dec,
                                                        int* __restrict__ rank, DNode* __restrict__ nodes, DBvh* __restrict__ bvhs, int* __restrict__ counts) {
  __shared__ int s_part[1024];
  __shared__ int s_base;
  const int tid = threadIdx.x;
  const int per = (ntasks + 1023) / 1024;
  const int b = tid * per, e = b + per < ntasks ? b + per : ntasks;
  int sum = 0;
  for (int t = b; t < e; t++) sum += dec[t].splits;
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < 1024; i++) {
      const int v = s_part[i];
      s_part[i] = acc;
      acc += v;
    }
    s_base = counts[0];
    counts[0] += acc;      // nodes after this level
    counts[1] = 2 * acc;   // tasks of the next level
  }
  __syncthreads();
  int r = s_part[tid];
  for (int t = b; t < e; t++) {
    rank[t] = r;
    const BuildTask task = tasks[t];
    const int split = dec[t].splits;
    const int child = split ? s_base + r : task.begin;     // node index, or the leaf's first primitive
    const int count = split ? 0 : task.end - task.begin;
    if (task.parent < 0) {
      DBvh bv = bvhs[task.bvh];
      if (count > 0) bv.root = -1, bv.root_start = child, bv.root_count = count;
      else bv.root = child, bv.root_start = 0, bv.root_count = 0;
      bvhs[task.bvh] = bv;
    } else {
      nodes[task.parent].child[task.which] = child;
      nodes[task.parent].count[task.which] = count;
    }
    r += split;
  }
}

// One workgroup per task that splits.
//
// The partition.  The reference's Lomuto loop (`for i: if pred(a[i]) swap(a[tail++], a[i])`, src/psl/algorithm.h:394-402)
// looks sequential, but its result has a closed description.  Elements satisfying the predicate ("trues") end up in front
// in their original order.  The others form a QUEUE between `tail` and `i`: a false element is appended at its back, and a
// true element -- swapped with a[tail], the queue's front -- moves the front element to the back.  Number the queue's
// entries in creation order (op k = the k-th element from the first false on: an append creates an entry holding that
// element; the j-th rotation consumes entry j -- the oldest unconsumed one -- and creates an entry holding the same
// element).  With R rotations in total the final queue is entries R, R+1, ... in order, and the element an entry holds is
// found by following src(k) = (append ? k : j(k)) to a fixed point, where j(k) = number of trues among the ops before k, a
// prefix sum.  Pointer jumping resolves all chains in log2(n) data-parallel rounds.  (tests/test_abi.py checks this
// formulation against the sequential loop on random predicates through pine_gpu_test_lomuto; the trees it builds are
// compared with the host build's in tests/test_gpu_parity.py.)
constexpr int kSplitBlock = 256;
__global__ void __launch_bounds__(kSplitBlock) bvh_split_kernel(BuildPrim* __restrict__ prims, BuildPrim* __restrict__ scratch, const BuildTask* __restrict__ tasks,
                                                                int ntasks, const BuildDecision* __restrict__ dec, const int* __restrict__ rank, int node_base,
                                                                DNode* __restrict__ nodes, BuildTask* __restrict__ next, int* __restrict__ g_perm,
                                                                int* __restrict__ g_src, unsigned char* __restrict__ g_pred, int lds_prims,
                                                                int* __restrict__ max_next) {
  extern __shared__ int s_dyn[];
  __shared__ int s_part[kSplitBlock];
  __shared__ int s_f0, s_left;
  __shared__ float s_box[4][12];
  const int t = blockIdx.x;
  if (t >= ntasks) return;
  const BuildDecision d = dec[t];
  if (!d.splits) return;
  const int tid = threadIdx.x;
  const BuildTask task = tasks[t];
  const int n = task.end - task.begin;
  BuildPrim* P = prims + task.begin;
  const bool in_lds = n <= lds_prims;  // (lds_prims: what this launch's dynamic LDS holds: perm | src | pred)
  int* perm = in_lds ? s_dyn : g_perm + task.begin;
  int* src = in_lds ? s_dyn + lds_prims : g_src + task.begin;
  unsigned char* pred = in_lds ? reinterpret_cast<unsigned char*>(s_dyn + 2 * lds_prims) : g_pred + task.begin;
  // 1. the predicate; a copy of the range to move from; per-thread chunk counts for the prefix sum
  const int per = (n + kSplitBlock - 1) / kSplitBlock;
  const int cb = tid * per < n ? tid * per : n, ce = cb + per < n ? cb + per : n;
  int count = 0, first_false = n;
  for (int i = cb; i < ce; i++) {
    const BuildPrim p = P[i];
    const bool pr = build_bucket_of(build_centroid(p, d.axis), d.clo[d.axis], d.chi[d.axis]) <= d.bucket;
    pred[i] = pr;
    scratch[task.begin + i] = p;
    count += pr ? 1 : 0;
    if (!pr && first_false == n) first_false = i;
  }
  s_part[tid] = count;
  if (tid == 0) s_f0 = n;
  __syncthreads();
  atomicMin(&s_f0, first_false);
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < kSplitBlock; i++) {
      const int v = s_part[i];
      s_part[i] = acc;
      acc += v;
    }
    s_left = acc;
  }
  __syncthreads();
  const int f0 = s_f0, left = s_left;  // first false position; number of trues
  // 2. trues: stable, in front.  ops from the first false on: src(k)
  {
    int T = s_part[tid];  // trues before position cb
    for (int i = cb; i < ce; i++) {
      const bool pr = pred[i] != 0;
      if (pr) perm[T] = i;
      if (i >= f0) src[i - f0] = pr ? T - f0 : i - f0;
      T += pr ? 1 : 0;
    }
  }
  __syncthreads();
  // 3. resolve the chains (in place: a value read mid-update is further along the same chain)
  const int K = n - f0, R = left - f0;
  if (K > 0) {
    int rounds = 1;
    while ((1 << rounds) < K) rounds++;
    for (int r = 0; r <= rounds; r++) {
      for (int k = tid; k < K; k += kSplitBlock) {
        const int s0 = src[k];
        src[k] = src[s0];
      }
      __syncthreads();
    }
    // 4. the falses, in final queue order, behind the trues
    for (int q = tid; q < n - left; q += kSplitBlock) perm[left + q] = f0 + src[R + q];
  }
  __syncthreads();
  // 5. the move and the two child boxes
  float lo0[3] = {kFloatMax, kFloatMax, kFloatMax}, hi0[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
  float lo1[3] = {kFloatMax, kFloatMax, kFloatMax}, hi1[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
  for (int i = tid; i < n; i += kSplitBlock) {
    const BuildPrim p = scratch[task.begin + perm[i]];
    P[i] = p;
    if (i < left) {
      for (int k = 0; k < 3; k++) lo0[k] = p.lo[k] < lo0[k] ? p.lo[k] : lo0[k], hi0[k] = p.hi[k] > hi0[k] ? p.hi[k] : hi0[k];
    } else {
      for (int k = 0; k < 3; k++) lo1[k] = p.lo[k] < lo1[k] ? p.lo[k] : lo1[k], hi1[k] = p.hi[k] > hi1[k] ? p.hi[k] : hi1[k];
    }
  }
  for (int k = 0; k < 3; k++) lo0[k] = wave_min(lo0[k]), hi0[k] = wave_max(hi0[k]), lo1[k] = wave_min(lo1[k]), hi1[k] = wave_max(hi1[k]);
  if ((tid & 63) == 0)
    for (int k = 0; k < 3; k++) s_box[tid >> 6][k] = lo0[k], s_box[tid >> 6][3 + k] = hi0[k], s_box[tid >> 6][6 + k] = lo1[k], s_box[tid >> 6][9 + k] = hi1[k];
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kSplitBlock / 64; w++)
      for (int k = 0; k < 3; k++) {
        lo0[k] = s_box[w][k] < lo0[k] ? s_box[w][k] : lo0[k];
        hi0[k] = s_box[w][3 + k] > hi0[k] ? s_box[w][3 + k] : hi0[k];
        lo1[k] = s_box[w][6 + k] < lo1[k] ? s_box[w][6 + k] : lo1[k];
        hi1[k] = s_box[w][9 + k] > hi1[k] ? s_box[w][9 + k] : hi1[k];
      }
    const int node = node_base + rank[t];
    DNode nd{};
    for (int k = 0; k < 3; k++) nd.lo0[k] = lo0[k], nd.hi0[k] = hi0[k], nd.lo1[k] = lo1[k], nd.hi1[k] = hi1[k];
    nodes[node] = nd;  // (child slots: filled by the next level's scan)
    BuildTask l{task.begin, task.begin + left, {lo0[0], lo0[1], lo0[2]}, {hi0[0], hi0[1], hi0[2]}, node, 0, task.bvh};
    BuildTask r{task.begin + left, task.end, {lo1[0], lo1[1], lo1[2]}, {hi1[0], hi1[1], hi1[2]}, node, 1, task.bvh};
    next[2 * rank[t]] = l;
    next[2 * rank[t] + 1] = r;
    atomicMax(max_next, left > n - left ? left : n - left);  // the next level's largest range (sizes its partition's LDS)
  }
}

// The same formulation on the host, for the CPU test of its equivalence with build_lomuto (serial loops standing for the
// data-parallel ones).
inline int build_lomuto_by_chains(const unsigned char* pred, int* perm, int n) {
  std::vector<int> T(size_t(n) + 1, 0);
  for (int i = 0; i < n; i++) T[size_t(i) + 1] = T[size_t(i)] + (pred[i] ? 1 : 0);
  const int left = T[size_t(n)];
  int f0 = n;
  for (int i = n - 1; i >= 0; i--)
    if (!pred[i]) f0 = i;
  for (int i = 0; i < n; i++)
    if (pred[i]) perm[T[size_t(i)]] = i;
  const int K = n - f0, R = left - f0;
  std::vector<int> src(size_t(K > 0 ? K : 0));
  for (int k = 0; k < K; k++) src[size_t(k)] = pred[f0 + k] ? T[size_t(f0 + k)] - f0 : k;
  int rounds = 1;
  while ((1 << rounds) < K) rounds++;
  for (int r = 0; r <= rounds; r++)
    for (int k = 0; k < K; k++) src[size_t(k)] = src[size_t(src[size_t(k)])];
  for (int q = 0; q < n - left; q++) perm[left + q] = f0 + src[size_t(R + q)];
  return left;
}

}  // namespace pine_gpu
