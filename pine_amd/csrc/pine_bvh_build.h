// pine_amd/csrc/pine_bvh_build.h -- the BVH build, level-synchronous (host code; pine_bvh_build_device.h runs the same
// schedule on the GPU).
//
// What must come out is pine's tree exactly (src/pine/impl/accel/bvh.cpp:43-147): the same splits (binned SAH, 16
// buckets, first minimum over axes x y z and buckets in order), the same leaves (a range is a leaf when it has one
// primitive or when no split costs less than testing them all) and -- because leaf primitives are tested in stored order
// and a scaled Box is order dependent -- the same primitive order, which the reference produces with a Lomuto partition
// (src/psl/algorithm.h:394-402) at every split.  How it is computed is this repository's: not a recursion but a
// breadth-first sweep over a list of open ranges ("tasks"), one level of every BVH of the scene at a time -- the
// top-level BVH and all mesh BVHs together -- so that nodes are numbered level by level across the whole two-level
// structure as they are created (the numbering the stage-queued kernel caches in LDS from the front), and so that a
// level is a data-parallel step: every reduction a task needs (centroid bounds, 3 x 16 bucket boxes and counts, the
// two child boxes) is a min / max / count, independent of evaluation order, hence identical on any number of lanes;
// the one sequential piece, the partition, runs over a one-byte predicate per primitive and an index permutation.
#pragma once
#include <vector>

#include "pine_host.h"

namespace pine_gpu {

struct BuildPrim {  // 32 bytes
  float lo[3], hi[3];
  int index;  // what the leaf lists: geometry index (top level) or triangle index within its mesh
  int pad;
};
struct BuildTask {
  int begin, end;      // range of BuildPrim (absolute positions in the scene-wide primitive array)
  float blo[3], bhi[3];  // bounds of the range's primitives
  int parent;          // node whose child slot this range fills; -1: the root of BVH `bvh`
  int which;           // child slot 0 / 1, or (parent == -1) nothing
  int bvh;             // index into FlatAccel::bvhs
};
constexpr int kBuildBuckets = 16;

struct SplitDecision {
  bool leaf;
  int axis, bucket;  // split: primitives whose bucket along `axis` is <= `bucket` go left
  float clo[3], chi[3];  // centroid bounds of the range (bucket of a primitive = f(centroid, these))
};
// bucket of a centroid coordinate: int(16 * relative_position) clamped (bvh.cpp:66-68, :105-107)
PINE_HD int build_bucket_of(float c, float lo, float hi) {
  const float o = c - lo, d = hi - lo;
  const float rel = d > 0.0f ? o / d : o;  // AABB::relative_position bbox.cpp:55-59
  // (bounds so large that hi - lo or a centroid overflows make `rel` NaN, infinite or negative: the reference's int() of that
  //  is undefined behaviour and indexes its bucket array with it; here such a primitive lands in an end bucket.  For finite
  //  geometry 0 <= rel <= 1 and nothing changes.)
  if (!(rel >= 0.0f)) return 0;
  if (rel >= 1.0f) return kBuildBuckets - 1;
  int b = int(float(kBuildBuckets) * rel);
  return b >= kBuildBuckets ? kBuildBuckets - 1 : b;
}
PINE_HD float build_centroid(const BuildPrim& p, int axis) { return (p.lo[axis] + p.hi[axis]) / 2; }
PINE_HD float build_area(const float lo[3], const float hi[3]) {  // AABB::surface_area bbox.cpp:60-63
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return 2.0f * (dx * dy + dx * dz + dy * dz);
}
// The SAH sweep over the bucket statistics of one range (bvh.cpp:71-103): counts and boxes per bucket and axis in,
// the decision out.  `n` primitives, `area` = surface area of the range's bounds.
struct BucketStats {
  int count[3][kBuildBuckets];
  float lo[3][kBuildBuckets][3], hi[3][kBuildBuckets][3];
};
PINE_HD void build_decide(const BucketStats& B, int n, float area, const float clo[3], const float chi[3], SplitDecision& out) {
  float min_cost = kFloatMax;
  int best_axis = -1, best_bucket = -1;
  for (int axis = 0; axis < 3; axis++) {
    if (chi[axis] <= clo[axis]) continue;  // degenerated axis
    float cost[kBuildBuckets - 1];
    {
      float lo[3] = {kFloatMax, kFloatMax, kFloatMax}, hi[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
      int c = 0;
      for (int i = 0; i < kBuildBuckets - 1; i++) {
        for (int k = 0; k < 3; k++) {
          lo[k] = B.lo[axis][i][k] < lo[k] ? B.lo[axis][i][k] : lo[k];
          hi[k] = B.hi[axis][i][k] > hi[k] ? B.hi[axis][i][k] : hi[k];
        }
        c += B.count[axis][i];
        cost[i] = float(c) * build_area(lo, hi);
      }
    }
    {
      float lo[3] = {kFloatMax, kFloatMax, kFloatMax}, hi[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
      int c = 0;
      for (int i = kBuildBuckets - 1; i >= 1; i--) {
        for (int k = 0; k < 3; k++) {
          lo[k] = B.lo[axis][i][k] < lo[k] ? B.lo[axis][i][k] : lo[k];
          hi[k] = B.hi[axis][i][k] > hi[k] ? B.hi[axis][i][k] : hi[k];
        }
        c += B.count[axis][i];
        cost[i - 1] += float(c) * build_area(lo, hi);
      }
    }
    float axis_min = kFloatMax;
    int axis_split = -1;
    for (int i = 0; i < kBuildBuckets - 1; i++) {
      const float v = 1.0f + cost[i] / area;
      if (v < axis_min) {
        axis_min = v;
        axis_split = i;
      }
    }
    if (axis_min < min_cost) {
      min_cost = axis_min;
      best_axis = axis;
      best_bucket = axis_split;
    }
  }
  out.leaf = min_cost > float(n);
  out.axis = best_axis;
  out.bucket = best_bucket;
  for (int k = 0; k < 3; k++) out.clo[k] = clo[k], out.chi[k] = chi[k];
}
// The reference's partition (psl::partition, src/psl/algorithm.h:394-402) on an index permutation: `pred[i]` for
// i in [0, n), `perm` starts as the identity; on return the element that belongs at position k is perm[k].  Returns
// the number of elements for which the predicate holds (they come first, in their original order; the others are
// rotated by the swaps, which is why this cannot be a stable partition).
PINE_HD int build_lomuto(const unsigned char* pred, int* perm, int n) {
  int tail = 0;
  for (int i = 0; i < n; i++)
    if (pred[i]) {
      const int t = perm[tail];
      perm[tail] = perm[i];
      perm[i] = t;
      tail++;
    }
  return tail;
}

// Host build of every BVH of a scene: `prims` holds the primitives of all BVHs back to back (meshes in geometry order,
// the top level last), `roots` one task per BVH covering its range, in the order the BVHs' roots are to be numbered.
// Appends nodes to A.nodes (breadth-first across all BVHs), permutes `prims` in place, fills A.bvhs[*].root / root_start /
// root_count.  A.prims is written by the caller from the permuted array.
void build_level_synchronous(std::vector<BuildPrim>& prims, const std::vector<BuildTask>& roots, FlatAccel& A);

}  // namespace pine_gpu
