// pine_amd/csrc/pine_host.h -- host-side scene model behind the C ABI (include/pine_gpu.h).
// Mirrors what the reference computes on the host before rendering: shape constructors
// (derived quantities), camera setup, area-light list, and pine's BVH build, flattened into the
// GPU records of pine_types.h.
#pragma once
#include <string>
#include <vector>

#include "pine_math.h"
#include "pine_types.h"

namespace pine_gpu {

// Full 4x4 host matrix in the reference's storage order: m[c][r] (column vectors).
struct Mat4 {
  float m[4][4];
  static Mat4 identity();
  static Mat4 from_rows(const float r[16]);  // the reference's scalar ctor (row-major arguments)
  static Mat4 from_storage(const float s[16]);
  void to_storage(float s[16]) const;
};
Mat4 mat4_mul(const Mat4& l, const Mat4& r);
Mat4 mat4_inverse(const Mat4& m);
Mat4 mat4_translate(f3 v);
Mat4 mat4_scale(f3 v);
Mat4 mat4_rotate_x(float rad);
Mat4 mat4_rotate_y(float rad);
Mat4 mat4_rotate_z(float rad);
Mat4 mat4_look_at(f3 from, f3 at);

struct HostAABB {
  f3 lower{kFloatMax, kFloatMax, kFloatMax};
  f3 upper{-kFloatMax, -kFloatMax, -kFloatMax};
  void extend(f3 p) {
    lower = vmin(lower, p);
    upper = vmax(upper, p);
  }
  void extend(const HostAABB& b) {
    lower = vmin(lower, b.lower);
    upper = vmax(upper, b.upper);
  }
  f3 centroid() const { return (lower + upper) / 2.0f; }
  float centroid(int d) const { return (get(lower, d) + get(upper, d)) / 2; }
  float surface_area() const {
    f3 d = upper - lower;
    return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
  }
  bool degenerated(int d) const { return get(upper, d) <= get(lower, d); }
  float relative_position(float p, int dim) const {
    float o = p - get(lower, dim);
    float d = get(upper, dim) - get(lower, dim);
    return d > 0.0f ? o / d : o;
  }
};

struct HostMesh {
  std::vector<float> vertices;    // 3 per vertex
  std::vector<uint32_t> indices;  // 3 per triangle
  std::vector<float> normals;     // 3 per vertex, or empty (Mesh::normals geometry.h:215)
  std::vector<float> texcoords;   // 2 per vertex, or empty (Mesh::texcoords :216)
};

struct HostGeometry {
  DShape shape;
  std::string describe;  // the .pscene line that recreates it (user-level parameters, hexfloat)
  int mesh = -1;         // index into SceneHost::meshes for SHAPE_MESH
};

struct BuildPrim;
struct BuildTask;
struct FlatAccel;
// The GPU build (pine_bvh_build_device.h), installed by the kernels' translation unit; same contract as
// build_level_synchronous.  Returns 0 on success.
using DeviceBuilderFn = int (*)(std::vector<BuildPrim>& prims, const std::vector<BuildTask>& roots, FlatAccel& A, int device);
extern DeviceBuilderFn g_device_builder;

struct FlatAccel {
  std::vector<DNode> nodes;
  std::vector<int> prims;  // leaf primitive ids (top level: geometry index; mesh: triangle index)
  int top_prim_begin = 0;  // prims[top_prim_begin ..) belong to the top-level BVH
  std::vector<DBvh> bvhs;  // [0] = top level, then one per mesh in lbvh order
  std::vector<float> tri_verts;    // all meshes: 9 floats per triangle (v0,v1,v2), mesh after mesh
  // device traversal copy: 12 floats per entry of `prims` (mesh entries only): v0,v1,v2, the triangle's index into
  // tri_verts (as int bits), 2 pad -- in LEAF order and 16-byte aligned, so that a leaf's triangles are consecutive
  // 48-byte records read with three wide loads, without the prims[i] -> triangle indirection
  std::vector<float> tri_leaf;
  // per-vertex attributes of meshes that have them, expanded per triangle like tri_verts: 16 floats per triangle
  // (n0 n1 n2, t0 t1 t2, pad); empty when no mesh of the scene carries normals or texcoords
  std::vector<float> tri_attrs;
  // the top-level primitives' own boxes in the reference's listing order (bvh.cpp:470-488: one per non-empty mesh, then the
  // other shapes, each in geometry order), 8 floats each: lower, geometry index (int bits), upper, 0 -- what
  // PINE_GPU_FLAG_ORDER_EMBREE builds its hierarchy from
  std::vector<float> top_boxes;
  bool built = false;
};

// One shading node of the scene's node table (node.h:13-297); ids are indices into SceneHost::nodes.
struct HostNode {
  enum Kind { ConstF, Const3, Position, Normal, UV, BinF, Bin3, UnF, Un3, Comp, ToVec3, Checker, Splat } kind = ConstF;
  char op = 0;
  int a = -1, b = -1, c = -1, n = 0;
  float f = 0;
  f3 v{0, 0, 0};
  bool is_vec3() const {
    return kind == Const3 || kind == Position || kind == Normal || kind == UV || kind == Bin3 || kind == Un3 || kind == ToVec3 || kind == Splat;
  }
};
// node ids of a material's parameters (-1: the literal in DMaterial is used)
struct MaterialNodes {
  int id[4] = {-1, -1, -1, -1};
};

struct SceneHost {
  std::vector<HostNode> nodes;
  std::vector<MaterialNodes> material_nodes;  // parallel to `materials`
  std::vector<DMaterial> materials;
  std::vector<std::string> material_names;
  std::vector<std::string> material_describe;
  std::vector<HostGeometry> geometries;
  std::vector<HostMesh> meshes;
  std::vector<DLight> lights;  // Scene::lights in add order (area lights of emissive geometry + explicit lights)
  bool has_env = false;
  DLight env{};                // environment light (appended to the sampler's list at plan build)
  std::vector<std::string> light_describe;
  std::string env_describe;
  // what describe() prints after the materials, in add order: (0, geometry index) or (1, light index)
  std::vector<std::pair<int, int>> item_order;
  DCamera camera{};
  bool has_camera = false;
  int tonemapper = 0;
  std::string camera_describe;
  FlatAccel accel;

  int add_node(const HostNode& n);  // returns the id, or -1 with the error set
  bool node_reads_surface(int id) const;
  f3 node_fold(int id) const;       // value of a node subtree that does not read the surface
  // flatten the programs of every material into `ops`; fills DMaterial::prog / folded literals of `out`
  bool compile_node_programs(std::vector<DMaterial>& out, std::vector<DNodeOp>& ops) const;
  int find_material(const char* name) const;
  int add_material(const char* name, const DMaterial& m, const std::string& desc);
  int add_geometry(HostGeometry g);
  HostAABB geometry_aabb(int g) const;
  void build_accel();
  int build_on_device = -1;  // >= 0: run the BVH build on this HIP device (pine_bvh_build_device.h) instead of on the host
  bool built_on_device = false;
  std::string describe() const;
};

void set_error(const std::string& msg);

}  // namespace pine_gpu
