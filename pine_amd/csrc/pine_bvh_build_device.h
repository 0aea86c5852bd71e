// pine_amd/csrc/pine_bvh_build_device.h -- the level-synchronous BVH build of pine_bvh_build.h on the GPU (included by
// pine_kernels.hip).  Same schedule, same arithmetic, same tree: three kernels per level --
//   decide : one wave per open range: centroid bounds (wave min/max), 3 x 16 bucket boxes and counts (LDS atomics on
//            order-preserving integer images of the floats), the SAH sweep by lane 0 (build_decide, shared with the host);
//   scan   : one workgroup: exclusive prefix sum of the "splits" flags -> node index and next-level task slot of every
//            range (nodes must be numbered in task order, not in the order waves finish);
//   split  : one wave per splitting range: the predicate per primitive, the reference's Lomuto swap sequence on an index
//            permutation by lane 0 (in LDS when the range fits), the move, the two child boxes, the node, the two new ranges.
// Every reduction is a min / max / count, so the result does not depend on how lanes share the work; the partition is
// sequential by nature (see pine_bvh_build.h) and is the critical path: n dependent LDS round trips for a range of n.
#pragma once
#include "pine_bvh_build.h"

namespace pine_gpu {

constexpr int kBuildWave = 64;
constexpr int kBuildLdsPrims = 12000;  // a range up to this size partitions in LDS (4-byte permutation + 1-byte predicate each)

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

// One wave per task that splits.
__global__ void __launch_bounds__(kBuildWave) bvh_split_kernel(BuildPrim* __restrict__ prims, BuildPrim* __restrict__ scratch, const BuildTask* __restrict__ tasks,
                                                               int ntasks, const BuildDecision* __restrict__ dec, const int* __restrict__ rank, int node_base,
                                                               DNode* __restrict__ nodes, BuildTask* __restrict__ next, int* __restrict__ g_perm,
                                                               unsigned char* __restrict__ g_pred, int lds_prims, int* __restrict__ max_next) {
  extern __shared__ int s_dyn[];
  const int t = blockIdx.x;
  if (t >= ntasks) return;
  const BuildDecision d = dec[t];
  if (!d.splits) return;
  const int lane = threadIdx.x;
  const BuildTask task = tasks[t];
  const int n = task.end - task.begin;
  BuildPrim* P = prims + task.begin;
  const bool in_lds = n <= lds_prims;  // (lds_prims: what this launch's dynamic LDS holds)
  int* perm = in_lds ? s_dyn : g_perm + task.begin;
  unsigned char* pred = in_lds ? reinterpret_cast<unsigned char*>(s_dyn + lds_prims) : g_pred + task.begin;
  for (int i = lane; i < n; i += kBuildWave) {
    pred[i] = build_bucket_of(build_centroid(P[i], d.axis), d.clo[d.axis], d.chi[d.axis]) <= d.bucket;
    perm[i] = i;
    scratch[task.begin + i] = P[i];
  }
  __syncthreads();
  __shared__ int s_left;
  if (lane == 0) s_left = build_lomuto(pred, perm, n);  // the reference's swap sequence: sequential by nature
  __syncthreads();
  const int left = s_left;
  float lo0[3] = {kFloatMax, kFloatMax, kFloatMax}, hi0[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
  float lo1[3] = {kFloatMax, kFloatMax, kFloatMax}, hi1[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
  for (int i = lane; i < n; i += kBuildWave) {
    const BuildPrim p = scratch[task.begin + perm[i]];
    P[i] = p;
    if (i < left) {
      for (int k = 0; k < 3; k++) lo0[k] = p.lo[k] < lo0[k] ? p.lo[k] : lo0[k], hi0[k] = p.hi[k] > hi0[k] ? p.hi[k] : hi0[k];
    } else {
      for (int k = 0; k < 3; k++) lo1[k] = p.lo[k] < lo1[k] ? p.lo[k] : lo1[k], hi1[k] = p.hi[k] > hi1[k] ? p.hi[k] : hi1[k];
    }
  }
  for (int k = 0; k < 3; k++) lo0[k] = wave_min(lo0[k]), hi0[k] = wave_max(hi0[k]), lo1[k] = wave_min(lo1[k]), hi1[k] = wave_max(hi1[k]);
  if (lane == 0) {
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

}  // namespace pine_gpu
