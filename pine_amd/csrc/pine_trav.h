// pine_amd/csrc/pine_trav.h -- the flat BVH traversal of the stage-queued kernel's F_LDS_TOP variants (included by
// pine_kernels.hip after SceneView / fetch_node; used by pine_queue_kernel.h).
//
// Why: in scenes with a real BVH (10 000 cones, triangle meshes) the rays of one wave need wildly
// different numbers of node visits; with nested node / leaf loops the wave runs until its longest ray
// is done and lanes wait for their neighbours' inner loops.  Measured on the round-1 structure: 12.8 % VALU lane
// utilisation on the Subsurface-icosphere scene, 27.6 % on the 10 000-cone scene.  Here a traversal is an explicit
// per-lane state machine -- per loop trip at most one node visit and one leaf primitive -- whose state is a handful of
// registers (TravState) and that can stop at any trip boundary: pine_queue_kernel.h either runs it to the end inside
// stages S / T, or (F_XSTAGE variants) makes it stages of its own in which a wave retires its finished lanes and hands
// them new rays from the queue while the others keep travelling.
//
// The order of operations per ray is pine's exactly (src/pine/impl/accel/bvh.cpp:321-451, :497-548):
// both child boxes are tested against the tmax captured when the node is visited, leaf children are
// tested in stored order before the next node, the child with the larger exit distance is pushed.
#pragma once

namespace pine_gpu {

struct TravState {
  int next;      // node to visit, -1 = none (pop)
  int sp;        // entries of this lane on its LDS stack
  int pa, pan;   // pending leaf primitives [pa, pa + pan), tested before anything else
  int pb, pbn;   // ... then these
  int mesh_base; // -1: in the top-level BVH; >= 0: inside a mesh BVH whose stack entries start here
  int mesh_word; // packed primitive word of that mesh (reported as the hit geometry)
  int r_next, r_pa, r_pan, r_pb, r_pbn;  // the top-level state that continues after the mesh
  int hit_geom;  // closest hit so far: packed primitive word, -1 = none
  int hit_prim;  // triangle index of a mesh hit
  int done;      // 0 travelling, 1 finished
};
constexpr int kTravRecordDwords = 8;  // a context's shadow-ray record in its global record: origin, direction, tmax, flags
// flags word of a shadow-ray record
enum : unsigned {
  kTravTerminalAfterShadow = 1u,  // the vertex has no continuation: after the shadow ray the context goes to stage T
  kTravClosestResolved = 2u,      // (top level baked, pine_specialize.h) the vertex's new ray ended in the top-level code: result in the context
};

__device__ __forceinline__ void trav_begin(const SceneView& S, TravState& ts) {
  ts.next = -1;
  ts.sp = 0;
  ts.pa = ts.pan = ts.pb = ts.pbn = 0;
  ts.mesh_base = -1;
  ts.mesh_word = 0;
  ts.r_next = -1;
  ts.r_pa = ts.r_pan = ts.r_pb = ts.r_pbn = 0;
  ts.hit_geom = -1;
  ts.hit_prim = 0;
  ts.done = 0;
  if (S.num_shapes == 0) {
    ts.done = 1;
    return;
  }
  const DBvh top = S.bvhs[0];
  if (top.root_count > 0) {  // the root itself is a leaf (bvh.cpp:331-334)
    ts.pa = top.root_start;
    ts.pan = top.root_count;
  } else if (top.root >= 0) {
    ts.next = top.root;
  } else {
    ts.done = 1;  // geometries exist but none has primitives
  }
}

// Advance the travelling lanes (`ts.done == 0`) one step per trip until every one is finished, or until fewer than
// `keep_lanes` of the wave are still travelling after at least `min_trips` trips (the caller then retires the finished
// lanes and gives them new rays).  ANY: BVH::hit (first hit ends the ray, hit_geom = 0); else BVH::intersect.
//
// A trip runs the node arm for the lanes at a node, THEN the primitive arm for the lanes with a pending leaf primitive --
// among them the lanes whose node visit has just found a leaf child, which so test its first primitive in the same trip
// (both arms are executed by the wave in nearly every trip anyway) -- and then the cheap bookkeeping that decides the
// lane's next step (second leaf range, pop, return from a mesh, finished).  The bookkeeping used to be trips of its own:
// a lane that only had to pop waited a whole trip of the other lanes' node and primitive tests for it.  The order of
// the steps per ray is unchanged: primitives pending > second pending range > node > pop > mesh return.
template <bool ANY, unsigned F, int STRIDE, class StackT>
__device__ __forceinline__ void trav_trips(const SceneView& S, DRay& ray, const DRayOct& oct, TravState& ts, StackT* stack, int keep_lanes,
                                           int min_trips, int* lane_trips = nullptr, TravLog* log = nullptr) {
  for (int trip = 0;; trip++) {
    const unsigned long long travelling = __ballot(ts.done == 0);
    if (travelling == 0) break;
    if (trip >= min_trips && __popcll(travelling) < keep_lanes) break;
    if (ts.done) continue;
    if (lane_trips) ++*lane_trips;  // (diagnostic builds: trips this lane took part in)
    REGION(ANY ? 5 : 1);  // (per-arm lane counts of -DPINE_PROFILE_REGIONS builds: a trip / its triangle, top-level primitive and node arms)
    // what a lane without a pending primitive does next: its second leaf range, a pop, the return from a mesh, or the end
    auto next_step = [&]() {
      if (!ts.done && ts.pan == 0) {
        if (ts.pbn > 0) {
          ts.pa = ts.pb, ts.pan = ts.pbn;
          ts.pbn = 0;
        } else if (ts.next < 0) {
          if (ts.sp > (ts.mesh_base >= 0 ? ts.mesh_base : 0)) {
            ts.next = int(stack[(--ts.sp) * STRIDE]);
          } else if (ts.mesh_base >= 0) {
#ifdef PINE_BAKED_TOP  /* scene-specialised build, one mesh: the top level is code (pine_specialize.h) and the r_ fields carry its results */
            ts.done = 1;
          } else if (false) {
#endif
            // mesh exhausted: back to the top-level leaf it was a primitive of (whatever that state needs next --
            // its pending primitives, its node, or a pop -- the next step does)
            ts.next = ts.r_next, ts.pa = ts.r_pa, ts.pan = ts.r_pan, ts.pb = ts.r_pb, ts.pbn = ts.r_pbn;
            ts.mesh_base = -1;
            if (ts.pan == 0 && ts.pbn > 0) {
              ts.pa = ts.pb, ts.pan = ts.pbn;
              ts.pbn = 0;
            }
          } else {
            ts.done = 1;
          }
        }
      }
    };
    // Node steps per trip: a ray visits 2.5 nodes per primitive it tests, and the primitive arm is the expensive one -- so
    // where the traversal runs inside stages S / T (every lane to its end, no refill) a trip runs the node arm TWICE (with
    // the bookkeeping in between: a lane that only has to pop does, and visits the popped node in the same trip) before the
    // primitive arm runs once for whoever has found a leaf meanwhile; those lanes sit the second node step out.  The order of
    // a ray's steps is unchanged.  10 000 cones: 7.10 -> 6.82 ms (three steps: 6.83).  In the traversal STAGES (lanes refilled;
    // mesh scenes) two steps are neutral (C5 131.9 -> 131.4 ms) and three or four cost 5 - 9 %: one step there.
    // ... and two again where the top level is code and the trips are the mesh's alone (pine_specialize.h: node arm and
    // triangle arm only): C5 100.7 -> 94.8 ms (three: 96.5).
#ifndef PINE_NODE_REPS_X
#ifdef PINE_BAKED_TOP
#define PINE_NODE_REPS_X 2
#else
#define PINE_NODE_REPS_X 1
#endif
#endif
#ifdef PINE_NODE_ADAPT  /* experiment: up to PINE_NODE_ADAPT_MAX node steps per trip, ended as soon as PINE_NODE_ADAPT lanes wait with a primitive or no lane has a node to visit */
    constexpr int kNodeReps = PINE_NODE_ADAPT_MAX;
#else
    constexpr int kNodeReps = (F & F_XSTAGE) != 0 ? PINE_NODE_REPS_X : 2;
#endif
#pragma unroll
    for (int rep = 0; rep < kNodeReps; rep++) {
    if (rep > 0) {
      next_step();
#ifdef PINE_NODE_ADAPT
      if (rep >= PINE_NODE_ADAPT_MIN) {
        const unsigned long long waiting = __ballot(ts.pan > 0 && !ts.done);
        const unsigned long long at_node = __ballot(ts.pan == 0 && ts.next >= 0 && !ts.done);
        if (at_node == 0 || __popcll(waiting) >= PINE_NODE_ADAPT) break;
      }
#endif
    }
    if (ts.pan == 0 && ts.next >= 0 && !ts.done) {
      // ---- one node: both child boxes against the tmax of this moment (bvh.cpp:405-446) ----
      REGION(ANY ? 6 : 2);
      const DNode nd = fetch_node<F>(S, ts.next);
      int l = -1, r = -1;
      float t0 = ray.tmax, t1 = ray.tmax;
#ifdef PINE_DUP_TRAV_NODES  /* cost measurement only: both box tests once more on opaque copies of EVERY operand (same film; the extra time is their cost) */
      {
        float q0 = ray.tmax, q1 = ray.tmax, tm = ray.tmin;
        DRayOct oo = oct;
        asm volatile("" : "+v"(tm), "+v"(oo.dir_inv.x), "+v"(oo.dir_inv.y), "+v"(oo.dir_inv.z), "+v"(oo.org_div_dir.x), "+v"(oo.org_div_dir.y), "+v"(oo.org_div_dir.z));
        const bool b0 = box_hit_oct(nd.lo0, nd.hi0, oo, tm, q0);
        const bool b1 = box_hit_oct(nd.lo1, nd.hi1, oo, tm, q1);
        float sink = (b0 ? q0 : 0.0f) + (b1 ? q1 : 0.0f);
        asm volatile("" : : "v"(sink));
      }
#endif
      if (box_hit_oct(nd.lo0, nd.hi0, oct, ray.tmin, t0)) {
        if (nd.count[0] == 0) l = nd.child[0];
        else ts.pa = nd.child[0], ts.pan = nd.count[0];
      }
      if (box_hit_oct(nd.lo1, nd.hi1, oct, ray.tmin, t1)) {
        if (nd.count[1] == 0) r = nd.child[1];
        else if (ts.pan > 0) ts.pb = nd.child[1], ts.pbn = nd.count[1];
        else ts.pa = nd.child[1], ts.pan = nd.count[1];
      }
      if (l != -1) {
        if (r != -1) {
          if (t0 > t1) {
            stack[ts.sp * STRIDE] = StackT(l);
            ts.next = r;
          } else {
            stack[ts.sp * STRIDE] = StackT(r);
            ts.next = l;
          }
          ts.sp++;
        } else ts.next = l;
      } else ts.next = r;  // (-1 when neither child is an inner node to visit)
    }
    }
    // (a node whose child is a leaf: its first primitive is tested in this same trip)
    if (ts.pan > 0) {
      // ---- one pending leaf primitive ----
      const int i = ts.pa++;
      ts.pan--;
      if (ts.mesh_base >= 0) {
        if constexpr (F & F_MESH) {
          REGION(ANY ? 8 : 4);
          float v[9];
          int tri;
          fetch_triangle<F>(S, i, v, tri);
          if (log) log->put(0x40000000u | unsigned(tri - S.bvhs[as_int(S.shapes[ts.mesh_word & kPrimIndexMask].f[2])].prim_base));  // (index within its mesh)
          if (ANY) {
            if (tri_hit(v, ray)) {
              ts.hit_geom = 0;
              ts.done = 1;
            }
          } else if (tri_intersect(v, ray)) {
            ts.hit_geom = ts.mesh_word;
            ts.hit_prim = tri;
          }
        }
      } else {
        REGION(ANY ? 7 : 3);
        const DShape* sh = &S.leaf[i];
        // The whole 128-byte record is fetched in ONE batch of loads before the kind is looked at: reading the kind word
        // first and the kind's fields after the dispatch is two dependent round trips per primitive test -- to L2 for the
        // 10 000-cone scene, whose records live in global memory (C4 7.70 -> 7.11 ms), to LDS elsewhere (C5 134.5 -> 132.1).
        // (Quads that no kind of the variant reads are dead loads the compiler drops.)
        DShape rec;
        {
          const uint4* src = reinterpret_cast<const uint4*>(sh);
          uint4* dst = reinterpret_cast<uint4*>(&rec);
#pragma unroll
          for (int q = 0; q < 8; q++) dst[q] = src[q];
          sh = &rec;
        }
        const int word = sh->kind;  // (the packed word rides in the copy's kind field)
        const int kind = word >> kPrimKindShift;
        bool is_mesh = false;
        if constexpr (F & F_MESH) is_mesh = kind == SHAPE_MESH;
        if (log) log->put(unsigned(word & kPrimIndexMask));
        if (is_mesh) {
          if constexpr (F & F_MESH) {
            // descend into the mesh's BVH; the top-level traversal continues when it is exhausted
            const DBvh mb = S.bvhs[as_int(sh->f[2])];
            ts.r_next = ts.next, ts.r_pa = ts.pa, ts.r_pan = ts.pan, ts.r_pb = ts.pb, ts.r_pbn = ts.pbn;
            ts.mesh_base = ts.sp;
            ts.mesh_word = word;
            ts.pb = ts.pbn = 0;
            if (mb.root_count > 0) {
              ts.next = -1;
              ts.pa = mb.root_start;
              ts.pan = mb.root_count;
            } else {
              ts.next = mb.root;
              ts.pa = ts.pan = 0;
            }
          }
        } else if (ANY) {
#ifdef PINE_DUP_TRAV_PRIMS  /* cost measurement only: the shape test once more on an opaque copy of the ray */
          {
            DRay rr = ray;
            asm volatile("" : "+v"(rr.tmin), "+v"(rr.o.x), "+v"(rr.d.x));
            float sink = shape_hit<F>(kind, sh, rr) ? 1.0f : 0.0f;
            asm volatile("" : : "v"(sink));
          }
#endif
          if (shape_hit<F>(kind, sh, ray)) {
            ts.hit_geom = 0;
            ts.done = 1;
          }
        } else {
#ifdef PINE_DUP_TRAV_PRIMS
          {
            DRay rr = ray;
            asm volatile("" : "+v"(rr.tmin), "+v"(rr.o.x), "+v"(rr.d.x));
            float sink = shape_intersect<F>(kind, sh, rr) ? rr.tmax : 0.0f;
            asm volatile("" : : "v"(sink));
          }
#endif
          if (shape_intersect<F>(kind, sh, ray)) ts.hit_geom = word;
        }
      }
    }
    // ---- what this lane does in the next trip ----
    next_step();
  }
}

}  // namespace pine_gpu
