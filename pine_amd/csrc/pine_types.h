// pine_amd/csrc/pine_types.h -- plain-old-data records shared by the host scene builder and the
// gfx950 kernels.  Layouts are chosen for the GPU: fixed-size 16-byte-aligned records that one lane
// fetches with a few dwordx4 loads (from LDS when the scene is staged there, else L1/L2).
//
// Reference counterparts (paths relative to /root/reference/): Shape variants
// src/pine/core/geometry.h:297-349, Material variants src/pine/core/material.h:112-131, BVH node
// src/pine/impl/accel/bvh.h:14-30, ThinLenCamera src/pine/core/camera.h:12-30.
#pragma once
#include <stdint.h>

namespace pine_gpu {

enum ShapeKind : int {
  SHAPE_RECT = 0,
  SHAPE_AABB = 1,
  SHAPE_OBB = 2,
  SHAPE_SPHERE = 3,
  SHAPE_DISK = 4,
  SHAPE_CONE = 5,
  SHAPE_MESH = 6,
  SHAPE_PLANE = 7,
  SHAPE_LINE = 8,
  SHAPE_CYLINDER = 9,
  SHAPE_TRIANGLE = 10,
};

// 128-byte shape record.  Field use per kind:
//  RECT  : f[0..2] position, [3..5] ex, [6..8] ey, [9..11] n, [12] lx, [13] ly, [14..16] rx,
//          [17..19] ry, [20] area (= lx*ly)
//  AABB  : f[0..2] lower, [3..5] upper
//  OBB   : f[0..2] lower, [3..5] upper, [6..17] m  (columns x,y,z,w; 3 rows each),
//          [18..29] m_inv (same layout)
//  SPHERE: f[0..2] c, [3] r
//  DISK  : f[0..2] position, [3..5] n, [6..8] u, [9..11] v, [12] r, [13] area
//  CONE  : f[0..2] apex p, [3..5] n, [6] r, [7] h, [8] A, [9] A2, [10] S, [11] area
//  MESH  : i(0) first_tri, i(1) num_tri, i(2) bvh index, f[3] area (first triangle x count),
//          i(4) attribute flags: 1 = per-vertex normals, 2 = per-vertex texcoords (FlatAccel::tri_attrs)
//  PLANE : f[0..2] position, [3..5] n, [6..8] u, [9..11] v
//  LINE  : f[0..2] p0, [3..5] p1, [6..8] tbn.x, [9..11] tbn.y, [12..14] tbn.z, [15] thickness, [16] area
//  CYLINDER: f[0..2] p0, [3..5] p1, [6..8] n, [9] r
//  TRIANGLE: f[0..8] v0 v1 v2, [9..11] n, [12] area
struct alignas(16) DShape {
  float f[30];
  int kind;
  int material;
};
static_assert(sizeof(DShape) == 128, "DShape must be 128 bytes");


enum MaterialKind : int {
  MAT_EMISSIVE = 0,
  MAT_DIFFUSE = 1,
  MAT_UBER = 2,
  MAT_SUBSURFACE = 3,
  MAT_METAL = 4,   // ConductorBSDF(albedo, max(roughness, min_roughness))              material.h:39-50
  MAT_GLOSSY = 5,  // DiffusiveDielectricBSDF(albedo, max(roughness, min_roughness), ior) material.h:52-64
  MAT_GLASS = 6,   // RefractiveDielectricBSDF(...)                                      material.h:66-78
};

// 80-byte material record.  Parameters are literals (constant shading nodes and node subtrees that do
// not read the surface are folded on the host) unless prog[k] >= 0: then
// parameter k (0 albedo, 1 roughness, 2 metallic, 3 transmission / ior of Glossy and Glass) is the
// result of the node program starting at DNodeOp index prog[k].
struct alignas(16) DMaterial {
  float color[3];  // albedo or emission
  int kind;
  float roughness, metallic, transmission, ior;
  float sigma_s[3];
  int pad;
  float color_over_pi[3];  // albedo / Pi, the Lambertian f (bxdf.cpp:21,27): same IEEE division, done once on the host
  int pad2;
  int prog[4];
};
static_assert(sizeof(DMaterial) == 80, "DMaterial must be 80 bytes");

// Shading-node program (node.h:13-297 flattened to postfix form): a stack machine over vec3 values; a
// float node is carried as a splat vec3 (every node operation is componentwise, so that is exact).
enum NodeOpCode : int {
  N_END = 0, N_CONST, N_POS, N_NORMAL, N_UV, N_ADD, N_SUB, N_MUL, N_DIV, N_POW, N_NEG, N_ABS, N_SQR, N_SQRT, N_FRACT,
  N_COMP,     // x = component index (as float): pop v, push splat(v[n])
  N_TOVEC3,   // pop z, y, x (splats), push (x, y, z)
  N_CHECKER,  // x = ratio: pop p, push splat(float(prod(fract(p) - ratio) > 0))  node.cpp:15-18
};
struct alignas(16) DNodeOp {
  int op;
  float x, y, z;
};
constexpr int kNodeStack = 8;  // evaluation stack depth the host guarantees

// 64-byte BVH node: the two child boxes live in the parent (as in pine's BVH) so one fetch decides
// both children.  child[i] >= 0 with count[i] == 0: inner node index.  count[i] > 0: child i is a
// leaf whose primitives are prim_index[child[i] .. child[i]+count[i]) (stored order = test order).
struct alignas(16) DNode {
  float lo0[3], hi0[3];
  float lo1[3], hi1[3];
  int child[2];
  int count[2];
};
static_assert(sizeof(DNode) == 64, "DNode must be 64 bytes");

// Top-level leaf primitive words on the DEVICE are packed: geometry index | emissive flag | kind, so
// that the leaf loop knows which intersection routine to run (and the path kernel whether the hit is
// a light) without first fetching the shape / material records -- one dependent LDS/L2 round trip
// less per primitive.  (The host-side FlatAccel::prims and the accel dump keep plain indices.)
constexpr int kPrimIndexMask = 0x03ffffff;
constexpr int kPrimEmissiveBit = 1 << 26;
constexpr int kPrimKindShift = 27;  // kinds 0..15 in bits 27..30 -> the packed word stays non-negative (-1 = miss)
static_assert(SHAPE_TRIANGLE < 16, "the shape kind must fit the 4 bits below the sign bit");

// One BVH (top level or per mesh).  If root_count > 0 the root itself is a leaf
// (bvh.cpp:331-334,396-399): test prim_index[root_start .. root_start+root_count) and stop.
struct DBvh {
  int root;        // inner root node index (valid when root_count == 0)
  int root_start;  // leaf-root primitive range
  int root_count;
  int prim_base;   // added to primitive ids of this BVH (mesh: first triangle; top level: 0)
};

struct DCamera {
  float position[3];
  float c2w[9];  // columns x,y,z
  float fov2d[2];
  float len_radius, focus_distance;
  int W, H;
};

// One entry of the light sampler's list (UniformLightSampler::build lightsampler.cpp:6-10): Scene::lights
// in add order -- an area light per emissive geometry, Point / Spot / Directional lights -- then the
// environment light, if any, last.  light.h:21-67.
enum LightKind : int { LIGHT_AREA = 0, LIGHT_POINT = 1, LIGHT_SPOT = 2, LIGHT_DIRECTIONAL = 3, LIGHT_SKY = 4 };
struct alignas(16) DLight {
  float position[3];
  int kind;
  float direction[3];  // normalised (Spot, Directional)
  int geom;            // LIGHT_AREA: geometry index
  float color[3];      // Point / Spot / Directional: colour; Sky: sun_color
  float falloff_cos;
  float cutoff_cos;
  int pad[3];
};
static_assert(sizeof(DLight) == 64, "DLight must be 64 bytes");

// Fold-stack entry (SURVEY.md 8(d) "FoldEntry 32 B"): what a non-terminal path vertex must keep
// until its child subtree has resolved, for the backward per-level clamp (path.cpp:114-121).
struct FoldEntry {
  float nee[3];  // direct lighting at this vertex
  float f[3];    // bs.f
  float cp;      // cosine / bs.pdf
  float pdf;     // bs.pdf (for the MIS weight against the child's light_pdf)
};
static_assert(sizeof(FoldEntry) == 32, "FoldEntry must be 32 bytes");

// PINE_GPU_FLAG_ORDER_EMBREE: the BVH8 the reference's EmbreeAccel walks (built by pine_embree_order.h on the host, read by
// scene_traverse_embree in the kernels).
// One node of that BVH8 as the device reads it: 16 quads.  Planes of the eight children per axis, then the child words:
// >= 0 a node, < 0 the complement of a primitive's place in SceneView::leaf, kEmbreeNoChild for an unused slot
// (whose planes are +inf / -inf: no ray enters it -- AABBNode::clear, vendored embree kernels/bvh/bvh_node_aabb.h:97-101).
struct EmbreeNode {
  float lo[3][8];
  float hi[3][8];
  int child[8];
  int count, pad[7];
};
static_assert(sizeof(EmbreeNode) == 256, "EmbreeNode is read as 16 quads");
constexpr int kEmbreeNoChild = -2147483647 - 1;
constexpr int kEmbreeStackEntries = 192;  // per-ray stack of scene_traverse_embree (the host checks the tree against it)

}  // namespace pine_gpu
