// pine_amd/csrc/pine_embree_order.h -- host side of PINE_GPU_FLAG_ORDER_EMBREE: the hierarchy the reference's DEFAULT accel
// walks (included by pine_kernels.hip; the device side is scene_traverse_embree in pine_kernels_device.h).
//
// What is reproduced, and from where.  EmbreeAccel registers every non-mesh shape as ONE Embree user primitive whose bounds and
// whose intersect callback are pine's own (src/pine/impl/accel/embree.cpp:12-40, :88-99).  So the reference's images depend on
// Embree only through WHICH primitives a closest-hit query hands to the callback, in WHICH order, with which tfar -- and pine has
// shapes whose answer depends on that (bbox.cpp:149-171, geometry.cpp:52, :171-244).  That order is a pure function of the
// primitives' boxes and the ray, defined by the vendored Embree 4.3.1 (src/contrib/embree) on an AVX2 x86 host:
//   * a BVH8 with one `Object` per leaf (kernels/common/scene.cpp:453-467, state.cpp:76-77), built by the AVX build of the binned
//     SAH builder: BVH8VirtualSceneBuilderSAH, SAH block size 8 (kernels/bvh/bvh_builder_sah.cpp:509-513);
//     GeneralBVHBuilder::BuilderT::recurse (kernels/builders/bvh_builder_sah.h:220-330): split, then keep splitting the child
//     with the largest half area until the node has eight, children sorted by size; HeuristicArrayBinningSAH
//     (heuristic_binning_array_aligned.h:100-181) over BinMapping / BinInfoT (heuristic_binning.h:16-120, :210-385);
//   * walked by BVHNIntersector1<8, BVH_AN1, false, ...> (kernels/bvh/bvh_intersector1.cpp:30-107): that is the device's part.
// This file builds the same tree from the same boxes: same bins, same sums, same float operations in the same order (the AVX
// builder's madd is a * b + c with two roundings, common/math/emath.h:328).  tests/test_embree_order.py pins the result against
// the call sequences of the REAL Embree (tests/golden/embree_order.npz, made by oracle/embree_probe.cpp).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <string>
#include <vector>

#include "pine_types.h"  // EmbreeNode, kEmbreeNoChild, kEmbreeStackEntries

namespace pine_gpu {


struct EmbreeOrderTree {
  std::vector<EmbreeNode> nodes;
  int root = kEmbreeNoChild;
  int stack_needed = 1;

  struct Ref {  // PrimRef: the callback's bounds and what the leaf stands for
    float lo[3], hi[3];
    int id, leaf;
  };
  struct Range {  // PrimInfoRange: a run of `refs` with the union of its boxes and of its doubled centres
    size_t first = 0, last = 0;
    float glo[3], ghi[3], clo[3], chi[3];
    Range() {
      for (int a = 0; a < 3; a++) glo[a] = clo[a] = std::numeric_limits<float>::infinity(), ghi[a] = chi[a] = -std::numeric_limits<float>::infinity();
    }
    size_t size() const { return last - first; }
    void take(const Ref& r) {
      for (int a = 0; a < 3; a++) {
        glo[a] = std::min(glo[a], r.lo[a]), ghi[a] = std::max(ghi[a], r.hi[a]);
        const float c2 = r.lo[a] + r.hi[a];
        clo[a] = std::min(clo[a], c2), chi[a] = std::max(chi[a], c2);
      }
    }
    float half_area() const {
      const float x = ghi[0] - glo[0], y = ghi[1] - glo[1], z = ghi[2] - glo[2];
      return x * (y + z) + y * z;
    }
  };
  struct Cut {  // BinSplit with its BinMapping
    int axis = -1, at = 0;
    float origin[3], per_unit[3];
    int slot(const Ref& r, int a) const { return int(std::floor(((r.lo[a] + r.hi[a]) - origin[a]) * per_unit[a])); }
  };

  std::vector<Ref> refs;

  static float half_area_of(const float* lo, const float* hi) {
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * (y + z) + y * z;
  }

  Cut choose(const Range& rg) const {
    constexpr int kBins = 32;
    Cut cut;
    const int bins = int(std::min<size_t>(kBins, size_t(4.0f + 0.05f * float(rg.size()))));
    for (int a = 0; a < 3; a++) {
      const float tiny = 1E-34f, extent = rg.chi[a] - rg.clo[a];
      const float width = tiny > extent ? tiny : extent;
      cut.per_unit[a] = width > tiny ? (0.99f * float(bins)) / width : 0.0f;
      cut.origin[a] = rg.clo[a];
    }
    struct Bin {
      float lo[3], hi[3];
      unsigned n = 0;
      Bin() {
        for (int a = 0; a < 3; a++) lo[a] = std::numeric_limits<float>::infinity(), hi[a] = -std::numeric_limits<float>::infinity();
      }
      void grow(const float* l, const float* h) {
        for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], l[a]), hi[a] = std::max(hi[a], h[a]);
      }
    };
    float best_cost = std::numeric_limits<float>::infinity();
    for (int a = 0; a < 3; a++) {  // (the three axes are independent in the reference's 4-wide sweep; the winner is the first axis with the strictly lowest cost)
      Bin bin[kBins];
      for (size_t i = rg.first; i < rg.last; i++) {
        const int b = std::clamp(cut.slot(refs[i], a), 0, bins - 1);
        bin[b].grow(refs[i].lo, refs[i].hi);
        bin[b].n++;
      }
      float right_area[kBins];
      unsigned right_n[kBins];
      {
        Bin acc;
        for (int i = bins - 1; i > 0; i--) {
          acc.n += bin[i].n;
          acc.grow(bin[i].lo, bin[i].hi);
          right_n[i] = acc.n;
          right_area[i] = half_area_of(acc.lo, acc.hi);
        }
      }
      float axis_cost = std::numeric_limits<float>::infinity();
      int axis_at = 0;
      Bin acc;
      for (int i = 1; i < bins; i++) {
        acc.n += bin[i - 1].n;
        acc.grow(bin[i - 1].lo, bin[i - 1].hi);
        // primitive counts in blocks of eight (sahBlockSize 8)
        const float cost = half_area_of(acc.lo, acc.hi) * float((acc.n + 7u) >> 3) + right_area[i] * float((right_n[i] + 7u) >> 3);
        if (cost < axis_cost) axis_at = i, axis_cost = cost;
      }
      if (cut.per_unit[a] == 0.0f) continue;
      if (axis_cost < best_cost && axis_at != 0) cut.axis = a, cut.at = axis_at, best_cost = axis_cost;
    }
    return cut;
  }
  void divide(const Range& rg, Range& left, Range& right) {
    const Cut cut = choose(rg);
    size_t mid;
    if (cut.axis < 0) {  // no usable split: ordered by id, halved (performFallbackSplit after deterministic_order)
      std::sort(refs.begin() + long(rg.first), refs.begin() + long(rg.last), [](const Ref& x, const Ref& y) { return x.id < y.id; });
      mid = (rg.first + rg.last) / 2;
    } else {  // (the order inside the halves never reaches the tree: halves are re-binned, the fallback sorts first)
      mid = size_t(std::stable_partition(refs.begin() + long(rg.first), refs.begin() + long(rg.last), [&](const Ref& r) { return cut.slot(r, cut.axis) < cut.at; }) - refs.begin());
    }
    left.first = rg.first, left.last = mid, right.first = mid, right.last = rg.last;
    for (size_t i = left.first; i < left.last; i++) left.take(refs[i]);
    for (size_t i = right.first; i < right.last; i++) right.take(refs[i]);
  }
  // -> child word; `held`: stack entries a traversal may hold above this subtree's parent
  int grow(const Range& rg, int level, int held, std::string& err) {
    if (rg.size() == 1) {
      stack_needed = std::max(stack_needed, held + 1);
      return ~refs[rg.first].leaf;
    }
    if (level + 8 >= 40) {  // (GeneralBVHBuilder switches to large leaves there: never reached by a hierarchy the device stack accepts)
      err = "the EmbreeAccel order's hierarchy is too deep";
      return kEmbreeNoChild;
    }
    Range part[8];
    int n = 2;
    divide(rg, part[0], part[1]);
    while (n < 8) {
      float widest = -std::numeric_limits<float>::infinity();
      int pick = -1;
      for (int i = 0; i < n; i++)
        if (part[i].size() > 1 && part[i].half_area() > widest) pick = i, widest = part[i].half_area();
      if (pick < 0) break;
      Range l, r;
      divide(part[pick], l, r);
      part[pick] = l;
      part[n++] = r;
    }
    std::stable_sort(part, part + n, [](const Range& x, const Range& y) { return x.size() > y.size(); });
    stack_needed = std::max(stack_needed, held + n);  // (the device sorts a node's hit children in place on the stack before it descends)
    const int me = int(nodes.size());
    nodes.emplace_back();
    {
      EmbreeNode& nd = nodes.back();
      nd.count = n;
      for (int i = 0; i < 8; i++) {
        nd.child[i] = kEmbreeNoChild;
        for (int a = 0; a < 3; a++) nd.lo[a][i] = i < n ? part[i].glo[a] : std::numeric_limits<float>::infinity(), nd.hi[a][i] = i < n ? part[i].ghi[a] : -std::numeric_limits<float>::infinity();
      }
      for (int i = 0; i < 7; i++) nd.pad[i] = 0;
    }
    for (int i = 0; i < n; i++) {
      const int c = grow(part[i], level + 1, held + n - 1, err);
      if (!err.empty()) return kEmbreeNoChild;
      nodes[size_t(me)].child[i] = c;
    }
    return me;
  }
  // boxes: 6 floats per primitive (lower, upper) in geometry order; leaf_of[i]: primitive i's place in SceneView::leaf
  bool build(const std::vector<float>& boxes, const std::vector<int>& leaf_of, std::string& err) {
    nodes.clear(), refs.clear();
    root = kEmbreeNoChild, stack_needed = 1;
    Range all;
    for (size_t i = 0; i < leaf_of.size(); i++) {
      Ref r;
      bool usable = true;  // isvalid_non_empty (common/math/bbox.h:100-102): finite below FLT_LARGE and not inverted, or Embree leaves the primitive out
      for (int a = 0; a < 3; a++) {
        r.lo[a] = boxes[6 * i + size_t(a)], r.hi[a] = boxes[6 * i + 3 + size_t(a)];
        usable = usable && r.lo[a] > -1.844E18f && r.hi[a] < 1.844E18f && r.lo[a] <= r.hi[a];
      }
      if (!usable) continue;
      r.id = int(i), r.leaf = leaf_of[i];
      refs.push_back(r);
      all.take(r);
    }
    all.first = 0, all.last = refs.size();
    if (refs.empty()) return true;
    root = grow(all, 1, 0, err);
    if (err.empty() && stack_needed > kEmbreeStackEntries) err = "the EmbreeAccel order's hierarchy needs " + std::to_string(stack_needed) + " stack entries per ray; the device keeps " + std::to_string(kEmbreeStackEntries);
    return err.empty();
  }
};

}  // namespace pine_gpu
