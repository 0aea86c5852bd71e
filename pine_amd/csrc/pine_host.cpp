// pine_amd/csrc/pine_host.cpp -- host side of the C ABI: scene container, shape constructors,
// host matrix math, pine's BVH build flattened for the GPU, film finalize.  No GPU calls here.
#include "pine_host.h"
#include "pine_specialize.h"
#include "pine_bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>

#include "../../include/pine_gpu.h"

namespace pine_gpu {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

static std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}
static std::string hex3(const float* v) { return fmt("%a %a %a", v[0], v[1], v[2]); }
static std::string hex3(f3 v) { return fmt("%a %a %a", v.x, v.y, v.z); }

// ------------------------------------------------------------------------------------------------
// Mat4 (src/pine/core/vecmath.h:575-640,1102-1180; src/pine/core/vecmath.cpp:103-132)
// ------------------------------------------------------------------------------------------------
Mat4 Mat4::identity() {
  Mat4 r{};
  for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
  return r;
}
Mat4 Mat4::from_rows(const float a[16]) {  // scalar ctor takes row-major args (vecmath.h:579-581)
  Mat4 r{};
  for (int row = 0; row < 4; row++)
    for (int col = 0; col < 4; col++) r.m[col][row] = a[row * 4 + col];
  return r;
}
Mat4 Mat4::from_storage(const float s[16]) {
  Mat4 r{};
  for (int c = 0; c < 4; c++)
    for (int rr = 0; rr < 4; rr++) r.m[c][rr] = s[c * 4 + rr];
  return r;
}
void Mat4::to_storage(float s[16]) const {
  for (int c = 0; c < 4; c++)
    for (int rr = 0; rr < 4; rr++) s[c * 4 + rr] = m[c][rr];
}
Mat4 mat4_mul(const Mat4& l, const Mat4& r) {  // vecmath.h:617-624: accumulate from zero, i = 0..3
  Mat4 ret{};
  for (int c = 0; c < 4; c++)
    for (int rr = 0; rr < 4; rr++) {
      float acc = 0.0f;
      for (int i = 0; i < 4; i++) acc += l.m[i][rr] * r.m[c][i];
      ret.m[c][rr] = acc;
    }
  return ret;
}
Mat4 mat4_inverse(const Mat4& M) {  // cofactor expansion with the reference's index pattern
  const auto& m = M.m;
  Mat4 R = Mat4::identity();
  float det = 0;
  for (int i = 0; i < 4; i++)
    det += (m[(1 + i) % 4][0] *
                (m[(2 + i) % 4][1] * m[(3 + i) % 4][2] - m[(3 + i) % 4][1] * m[(2 + i) % 4][2]) +
            m[(2 + i) % 4][0] *
                (m[(3 + i) % 4][1] * m[(1 + i) % 4][2] - m[(1 + i) % 4][1] * m[(3 + i) % 4][2]) +
            m[(3 + i) % 4][0] *
                (m[(1 + i) % 4][1] * m[(2 + i) % 4][2] - m[(2 + i) % 4][1] * m[(1 + i) % 4][2])) *
           m[i % 4][3] * (i % 2 ? -1 : 1);
  if (det == 0) return R;
  for (int v = 0; v < 4; v++)
    for (int i = 0; i < 4; i++)
      R.m[v][i] = (m[(1 + i) % 4][(1 + v) % 4] *
                       (m[(2 + i) % 4][(2 + v) % 4] * m[(3 + i) % 4][(3 + v) % 4] -
                        m[(3 + i) % 4][(2 + v) % 4] * m[(2 + i) % 4][(3 + v) % 4]) +
                   m[(2 + i) % 4][(1 + v) % 4] *
                       (m[(3 + i) % 4][(2 + v) % 4] * m[(1 + i) % 4][(3 + v) % 4] -
                        m[(1 + i) % 4][(2 + v) % 4] * m[(3 + i) % 4][(3 + v) % 4]) +
                   m[(3 + i) % 4][(1 + v) % 4] *
                       (m[(1 + i) % 4][(2 + v) % 4] * m[(2 + i) % 4][(3 + v) % 4] -
                        m[(2 + i) % 4][(2 + v) % 4] * m[(1 + i) % 4][(3 + v) % 4])) *
                  ((v + i) % 2 ? 1 : -1);
  for (int c = 0; c < 4; c++)
    for (int rr = 0; rr < 4; rr++) R.m[c][rr] /= det;
  return R;
}
Mat4 mat4_translate(f3 v) {
  const float a[16] = {1, 0, 0, v.x, 0, 1, 0, v.y, 0, 0, 1, v.z, 0, 0, 0, 1};
  return Mat4::from_rows(a);
}
Mat4 mat4_scale(f3 v) {
  const float a[16] = {v.x, 0, 0, 0, 0, v.y, 0, 0, 0, 0, v.z, 0, 0, 0, 0, 1};
  return Mat4::from_rows(a);
}
// The host trig goes through the system libm exactly as the reference's does (psl::cos -> std::cos).
Mat4 mat4_rotate_x(float r) {
  const float a[16] = {1, 0, 0, 0, 0, std::cos(r), -std::sin(r), 0, 0, std::sin(r), std::cos(r), 0, 0, 0, 0, 1};
  return Mat4::from_rows(a);
}
Mat4 mat4_rotate_y(float r) {
  const float a[16] = {std::cos(r), 0, std::sin(r), 0, 0, 1, 0, 0, -std::sin(r), 0, std::cos(r), 0, 0, 0, 0, 1};
  return Mat4::from_rows(a);
}
Mat4 mat4_rotate_z(float r) {
  const float a[16] = {std::cos(r), -std::sin(r), 0, 0, std::sin(r), std::cos(r), 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  return Mat4::from_rows(a);
}
Mat4 mat4_look_at(f3 from, f3 at) {  // vecmath.h:1172-1180, up = (0,1,0)
  f3 up = mk3(0, 1, 0);
  f3 z = normalize(at - from);
  if (std::abs(dot(z, up)) > 0.999f) z = normalize(z + mk3(0.0f, 0.0f, 1e-5f));
  f3 x = normalize(cross(up, z));
  f3 y = cross(z, x);
  Mat4 r{};
  const f3 cols[3] = {x, y, z};
  for (int c = 0; c < 3; c++) {
    r.m[c][0] = cols[c].x;
    r.m[c][1] = cols[c].y;
    r.m[c][2] = cols[c].z;
    r.m[c][3] = 0.0f;
  }
  r.m[3][0] = from.x;
  r.m[3][1] = from.y;
  r.m[3][2] = from.z;
  r.m[3][3] = 1.0f;
  return r;
}

// ------------------------------------------------------------------------------------------------
// Scene
// ------------------------------------------------------------------------------------------------
int SceneHost::find_material(const char* name) const {
  for (int i = int(material_names.size()) - 1; i >= 0; i--)
    if (material_names[i] == name) return i;
  return -1;
}
// ---- shading nodes (node.h:13-297) ----------------------------------------------------------------
int SceneHost::add_node(const HostNode& n) {
  auto ok = [&](int id, bool want_vec3) {
    if (id < 0 || id >= int(nodes.size())) {
      set_error("node: operand id out of range");
      return false;
    }
    if (nodes[size_t(id)].is_vec3() != want_vec3) {
      set_error(want_vec3 ? "node: a Node3f operand is required" : "node: a Nodef operand is required");
      return false;
    }
    return true;
  };
  switch (n.kind) {
    case HostNode::BinF:
      if (!ok(n.a, false) || !ok(n.b, false)) return -1;
      break;
    case HostNode::Bin3:
      if (!ok(n.a, true) || !ok(n.b, true)) return -1;
      break;
    case HostNode::UnF:
    case HostNode::Splat:
      if (!ok(n.a, false)) return -1;
      break;
    case HostNode::Un3:
    case HostNode::Checker:
      if (!ok(n.a, true)) return -1;
      break;
    case HostNode::Comp:
      if (!ok(n.a, true)) return -1;
      if (n.n < 0 || n.n > 2) {  // node.h:181-182
        set_error("NodeComponent's second parameter should be 0, 1, or 2");
        return -1;
      }
      break;
    case HostNode::ToVec3:
      if (!ok(n.a, false)) return -1;
      if (n.b >= 0 || n.c >= 0)
        if (!ok(n.b, false) || !ok(n.c, false)) return -1;
      break;
    default: break;
  }
  if ((n.kind == HostNode::BinF || n.kind == HostNode::Bin3) && !strchr("+-*/^", n.op)) {
    set_error("node: unknown binary operator");
    return -1;
  }
  if ((n.kind == HostNode::UnF || n.kind == HostNode::Un3) && !strchr("-asrf", n.op)) {
    set_error("node: unknown unary operator");
    return -1;
  }
  nodes.push_back(n);
  return int(nodes.size()) - 1;
}
bool SceneHost::node_reads_surface(int id) const {
  const HostNode& k = nodes[size_t(id)];
  if (k.kind == HostNode::Position || k.kind == HostNode::Normal || k.kind == HostNode::UV) return true;
  for (int c : {k.a, k.b, k.c})
    if (c >= 0 && node_reads_surface(c)) return true;
  return false;
}
static float node_un(char op, float x) {  // NodeUnary::eval node.h:158-175
  switch (op) {
    case '-': return -x;
    case 'a': return fabsf(x);
    case 's': return x * x;
    case 'r': return sqrtf(x);
    default: return x - floorf(x);  // psl::fract math.h:152-154
  }
}
static float node_bin(char op, float x, float y) {  // NodeBinary::eval node.h:134-150
  switch (op) {
    case '+': return x + y;
    case '-': return x - y;
    case '*': return x * y;
    case '/': return x / y;
    default: return powf(x, y);
  }
}
f3 SceneHost::node_fold(int id) const {  // floats come back as splats
  const HostNode& k = nodes[size_t(id)];
  switch (k.kind) {
    case HostNode::ConstF: return f3{k.f, k.f, k.f};
    case HostNode::Const3: return k.v;
    case HostNode::BinF:
    case HostNode::Bin3: {
      const f3 x = node_fold(k.a), y = node_fold(k.b);
      return f3{node_bin(k.op, x.x, y.x), node_bin(k.op, x.y, y.y), node_bin(k.op, x.z, y.z)};
    }
    case HostNode::UnF:
    case HostNode::Un3: {
      const f3 x = node_fold(k.a);
      return f3{node_un(k.op, x.x), node_un(k.op, x.y), node_un(k.op, x.z)};
    }
    case HostNode::Comp: {
      const float v = get(node_fold(k.a), k.n);
      return f3{v, v, v};
    }
    case HostNode::ToVec3:
      if (k.b < 0) return node_fold(k.a);
      return f3{node_fold(k.a).x, node_fold(k.b).x, node_fold(k.c).x};
    case HostNode::Checker: {
      const f3 q = node_fold(k.a);
      const f3 x{node_un('f', q.x) - k.f, node_un('f', q.y) - k.f, node_un('f', q.z) - k.f};
      const float v = float(x.x * x.y * x.z > 0);
      return f3{v, v, v};
    }
    case HostNode::Splat: return node_fold(k.a);
    default: return f3{0, 0, 0};
  }
}
// postfix flattening of one node tree; returns the maximum stack depth, or -1
static int emit_program(const SceneHost& S, int id, std::vector<DNodeOp>& ops, int depth) {
  const HostNode& k = S.nodes[size_t(id)];
  auto push = [&](int op, float x = 0, float y = 0, float z = 0) { ops.push_back(DNodeOp{op, x, y, z}); };
  if (!S.node_reads_surface(id)) {  // constant subtree: one literal
    const f3 v = S.node_fold(id);
    push(N_CONST, v.x, v.y, v.z);
    return depth + 1;
  }
  auto binop = [](char c) { return c == '+' ? N_ADD : c == '-' ? N_SUB : c == '*' ? N_MUL : c == '/' ? N_DIV : N_POW; };
  auto unop = [](char c) { return c == '-' ? N_NEG : c == 'a' ? N_ABS : c == 's' ? N_SQR : c == 'r' ? N_SQRT : N_FRACT; };
  int m = depth + 1;
  switch (k.kind) {
    case HostNode::Position: push(N_POS); break;
    case HostNode::Normal: push(N_NORMAL); break;
    case HostNode::UV: push(N_UV); break;
    case HostNode::BinF:
    case HostNode::Bin3: {
      const int m1 = emit_program(S, k.a, ops, depth), m2 = emit_program(S, k.b, ops, depth + 1);
      if (m1 < 0 || m2 < 0) return -1;
      push(binop(k.op));
      m = std::max(m1, m2);
      break;
    }
    case HostNode::UnF:
    case HostNode::Un3:
      m = emit_program(S, k.a, ops, depth);
      push(unop(k.op));
      break;
    case HostNode::Comp:
      m = emit_program(S, k.a, ops, depth);
      push(N_COMP, float(k.n));
      break;
    case HostNode::ToVec3:
      if (k.b < 0) {
        m = emit_program(S, k.a, ops, depth);  // splat already
      } else {
        const int m1 = emit_program(S, k.a, ops, depth), m2 = emit_program(S, k.b, ops, depth + 1),
                  m3 = emit_program(S, k.c, ops, depth + 2);
        if (m1 < 0 || m2 < 0 || m3 < 0) return -1;
        push(N_TOVEC3);
        m = std::max(m1, std::max(m2, m3));
      }
      break;
    case HostNode::Checker:
      m = emit_program(S, k.a, ops, depth);
      push(N_CHECKER, k.f);
      break;
    case HostNode::Splat: m = emit_program(S, k.a, ops, depth); break;
    default: return -1;
  }
  return m;
}
bool SceneHost::compile_node_programs(std::vector<DMaterial>& out, std::vector<DNodeOp>& ops) const {
  out = materials;
  ops.clear();
  for (size_t i = 0; i < out.size(); i++) {
    DMaterial& m = out[i];
    for (int k = 0; k < 4; k++) {
      m.prog[k] = -1;
      const int id = material_nodes[i].id[k];
      if (id < 0) continue;
      if (!node_reads_surface(id)) {  // fold to a literal (exactly the arithmetic the node tree would do)
        const f3 v = node_fold(id);
        if (k == 0) {
          m.color[0] = v.x, m.color[1] = v.y, m.color[2] = v.z;
          for (int c = 0; c < 3; c++) m.color_over_pi[c] = m.color[c] / kPi;
        } else if (k == 1) m.roughness = v.x;
        else if (k == 2) m.metallic = v.x;
        else if (m.kind == MAT_GLOSSY || m.kind == MAT_GLASS) m.ior = v.x;
        else m.transmission = v.x;
        continue;
      }
      m.prog[k] = int(ops.size());
      const int depth = emit_program(*this, id, ops, 0);
      ops.push_back(DNodeOp{N_END, 0, 0, 0});
      if (depth < 0 || depth > kNodeStack) {
        set_error("shading-node tree too deep for the device evaluator");
        return false;
      }
    }
  }
  return true;
}

int SceneHost::add_material(const char* name, const DMaterial& m_, const std::string& desc) {
  DMaterial m = m_;
  for (int i = 0; i < 3; i++) m.color_over_pi[i] = m.color[i] / kPi;
  for (int& p : m.prog) p = -1;
  materials.push_back(m);
  material_nodes.emplace_back();
  material_names.push_back(name ? name : "");
  material_describe.push_back(desc);
  accel.built = false;
  return int(materials.size()) - 1;
}
int SceneHost::add_geometry(HostGeometry g) {
  if (g.shape.material < 0 || g.shape.material >= int(materials.size())) {
    set_error("scene.add: invalid material id");
    return -1;
  }
  geometries.push_back(std::move(g));
  int id = int(geometries.size()) - 1;
  // Scene::add_geometry (scene.cpp:19-20): emissive geometry becomes an AreaLight
  item_order.push_back({0, id});
  if (materials[geometries[id].shape.material].kind == MAT_EMISSIVE) {
    DLight L{};
    L.kind = LIGHT_AREA;
    L.geom = id;
    lights.push_back(L);
    light_describe.push_back("");
  }
  accel.built = false;
  return id;
}

static HostAABB sphere_aabb(f3 c, float r) {
  HostAABB b;
  b.lower = c - mk3(r);
  b.upper = c + mk3(r);
  return b;
}

HostAABB SceneHost::geometry_aabb(int gi) const {
  const DShape& s = geometries[gi].shape;
  const float* f = s.f;
  HostAABB b;
  switch (s.kind) {
    case SHAPE_RECT: {  // Rect::get_aabb geometry.cpp:401-408
      f3 position = ld3(f), ex = ld3(f + 3), ey = ld3(f + 6);
      float lx = f[12], ly = f[13];
      b.extend(position - ex * lx / 2.0f - ey * ly / 2.0f);
      b.extend(position - ex * lx / 2.0f + ey * ly / 2.0f);
      b.extend(position + ex * lx / 2.0f - ey * ly / 2.0f);
      b.extend(position + ex * lx / 2.0f + ey * ly / 2.0f);
      break;
    }
    case SHAPE_AABB:  // bbox.h:77: padded by epsilon
      b.lower = ld3(f) - mk3(kEpsilon);
      b.upper = ld3(f + 3) + mk3(kEpsilon);
      break;
    case SHAPE_OBB: {  // AABB(OBB) bbox.cpp:8-16, no pad (bbox.h:93)
      f3 lo = ld3(f), hi = ld3(f + 3);
      m34 m = ld34(f + 6);
      for (int i = 0; i < 8; i++) {
        f3 p = lo;
        if (i % 2 >= 1) p.x = hi.x;
        if (i % 4 >= 2) p.y = hi.y;
        if (i % 8 >= 4) p.z = hi.z;
        b.extend(mul_point(m, p));
      }
      break;
    }
    case SHAPE_SPHERE: b = sphere_aabb(ld3(f), f[3]); break;
    case SHAPE_DISK: b = sphere_aabb(ld3(f), f[12]); break;  // Disk::get_aabb geometry.cpp:169
    case SHAPE_CONE: {  // geometry.h:129: bottom.get_aabb().extend(apex); bottom centre = p - n_*h
      // the bottom disk centre is kept in f[12..14]
      b = sphere_aabb(ld3(f + 12), f[6]);
      b.extend(ld3(f));
      break;
    }
    case SHAPE_PLANE:  // geometry.cpp:52
      b.lower = ld3(f) + mk3(-100.0f);
      b.upper = ld3(f) + mk3(100.0f);
      break;
    case SHAPE_LINE: {  // geometry.cpp:237-244
      const f3 p0 = ld3(f), p1 = ld3(f + 3), th = mk3(f[15]);
      b.extend(p0 - th);
      b.extend(p1 - th);
      b.extend(p0 + th);
      b.extend(p1 + th);
      break;
    }
    case SHAPE_CYLINDER:  // geometry.h:141,147: both cap disks are built at p0, so the box is the one of Sphere(p0, r)
      b = sphere_aabb(ld3(f), f[9]);
      break;
    case SHAPE_TRIANGLE:  // geometry.cpp:588-594
      b.extend(ld3(f));
      b.extend(ld3(f + 3));
      b.extend(ld3(f + 6));
      break;
    case SHAPE_MESH: {
      const HostMesh& m = meshes[geometries[gi].mesh];
      for (size_t i = 0; i < m.vertices.size() / 3; i++) b.extend(ld3(&m.vertices[3 * i]));
      break;
    }
  }
  return b;
}

DeviceBuilderFn g_device_builder = nullptr;

// ---- pine's BVH build, level-synchronous (pine_bvh_build.h) ------------------------------------------------------
void build_level_synchronous(std::vector<BuildPrim>& prims, const std::vector<BuildTask>& roots, FlatAccel& A) {
  std::vector<BuildTask> level = roots, next;
  std::vector<unsigned char> pred;
  std::vector<int> perm;
  std::vector<BuildPrim> scratch;
  auto set_slot = [&](const BuildTask& t, int child, int count) {  // the parent's child slot, or the BVH's root
    if (t.parent < 0) {
      DBvh& b = A.bvhs[size_t(t.bvh)];
      if (count > 0) b.root = -1, b.root_start = child, b.root_count = count;
      else b.root = child, b.root_start = 0, b.root_count = 0;
    } else {
      A.nodes[size_t(t.parent)].child[t.which] = child;
      A.nodes[size_t(t.parent)].count[t.which] = count;
    }
  };
  while (!level.empty()) {
    next.clear();
    for (const BuildTask& t : level) {
      const int n = t.end - t.begin;
      BuildPrim* P = prims.data() + t.begin;
      if (n == 1) {
        set_slot(t, t.begin, 1);
        continue;
      }
      // reductions of the range: centroid bounds, then counts and boxes per bucket and axis
      float clo[3] = {kFloatMax, kFloatMax, kFloatMax}, chi[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
      for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
          const float c = build_centroid(P[i], k);
          clo[k] = c < clo[k] ? c : clo[k];
          chi[k] = c > chi[k] ? c : chi[k];
        }
      BucketStats B;
      for (int a = 0; a < 3; a++)
        for (int j = 0; j < kBuildBuckets; j++) {
          B.count[a][j] = 0;
          for (int k = 0; k < 3; k++) B.lo[a][j][k] = kFloatMax, B.hi[a][j][k] = -kFloatMax;
        }
      for (int a = 0; a < 3; a++) {
        if (chi[a] <= clo[a]) continue;
        for (int i = 0; i < n; i++) {
          const int j = build_bucket_of(build_centroid(P[i], a), clo[a], chi[a]);
          B.count[a][j]++;
          for (int k = 0; k < 3; k++) {
            B.lo[a][j][k] = P[i].lo[k] < B.lo[a][j][k] ? P[i].lo[k] : B.lo[a][j][k];
            B.hi[a][j][k] = P[i].hi[k] > B.hi[a][j][k] ? P[i].hi[k] : B.hi[a][j][k];
          }
        }
      }
      SplitDecision d;
      build_decide(B, n, build_area(t.blo, t.bhi), clo, chi, d);
      if (d.leaf) {
        set_slot(t, t.begin, n);
        continue;
      }
      // the partition: predicate per primitive, the swap sequence on an index permutation, then the move
      pred.resize(size_t(n));
      perm.resize(size_t(n));
      for (int i = 0; i < n; i++) {
        pred[size_t(i)] = build_bucket_of(build_centroid(P[i], d.axis), clo[d.axis], chi[d.axis]) <= d.bucket;
        perm[size_t(i)] = i;
      }
      const int left = build_lomuto(pred.data(), perm.data(), n);
      scratch.assign(P, P + n);
      for (int i = 0; i < n; i++) P[i] = scratch[size_t(perm[size_t(i)])];
      // the node, with its two child boxes; its child slots are filled when the two new ranges are processed
      BuildTask l{t.begin, t.begin + left, {kFloatMax, kFloatMax, kFloatMax}, {-kFloatMax, -kFloatMax, -kFloatMax}, int(A.nodes.size()), 0, t.bvh};
      BuildTask r{t.begin + left, t.end, {kFloatMax, kFloatMax, kFloatMax}, {-kFloatMax, -kFloatMax, -kFloatMax}, int(A.nodes.size()), 1, t.bvh};
      for (int i = 0; i < n; i++) {
        BuildTask& c = i < left ? l : r;
        for (int k = 0; k < 3; k++) {
          c.blo[k] = P[i].lo[k] < c.blo[k] ? P[i].lo[k] : c.blo[k];
          c.bhi[k] = P[i].hi[k] > c.bhi[k] ? P[i].hi[k] : c.bhi[k];
        }
      }
      DNode nd{};
      for (int k = 0; k < 3; k++) nd.lo0[k] = l.blo[k], nd.hi0[k] = l.bhi[k], nd.lo1[k] = r.blo[k], nd.hi1[k] = r.bhi[k];
      set_slot(t, int(A.nodes.size()), 0);
      A.nodes.push_back(nd);
      next.push_back(l);
      next.push_back(r);
    }
    level.swap(next);
  }
}

void SceneHost::build_accel() {  // BVH::build bvh.cpp:453-495
  accel = FlatAccel();
  accel.bvhs.push_back(DBvh{-1, 0, 0, 0});  // slot 0 = top level
  std::vector<BuildPrim> prims;   // every BVH's primitives back to back: meshes in geometry order, the top level last
  std::vector<BuildTask> roots;   // one open range per BVH
  std::vector<BuildPrim> top;
  auto make_prim = [](const HostAABB& b, int index) {
    // (+ 0.0f: a box coordinate of -0 becomes +0.  Every use of a box coordinate is a comparison or feeds one -- the sign
    //  of a zero never decides anything -- but min / max reductions pick between -0 and +0 by evaluation order, and the
    //  device build must produce the host build's bits on any number of lanes.)
    BuildPrim p{};
    p.lo[0] = b.lower.x + 0.0f, p.lo[1] = b.lower.y + 0.0f, p.lo[2] = b.lower.z + 0.0f;
    p.hi[0] = b.upper.x + 0.0f, p.hi[1] = b.upper.y + 0.0f, p.hi[2] = b.upper.z + 0.0f;
    p.index = index;
    return p;
  };
  auto root_task = [&](int begin, int end, int bvh) {
    BuildTask t{begin, end, {kFloatMax, kFloatMax, kFloatMax}, {-kFloatMax, -kFloatMax, -kFloatMax}, -1, 0, bvh};
    for (int i = begin; i < end; i++)
      for (int k = 0; k < 3; k++) {
        t.blo[k] = prims[size_t(i)].lo[k] < t.blo[k] ? prims[size_t(i)].lo[k] : t.blo[k];
        t.bhi[k] = prims[size_t(i)].hi[k] > t.bhi[k] ? prims[size_t(i)].hi[k] : t.bhi[k];
      }
    return t;
  };
  // 1. one BVH per non-empty mesh, in geometry order
  for (size_t gi = 0; gi < geometries.size(); gi++) {
    if (geometries[gi].shape.kind != SHAPE_MESH) continue;
    const HostMesh& m = meshes[geometries[gi].mesh];
    size_t nt = m.indices.size() / 3;
    if (nt == 0) continue;
    int first_tri = int(accel.tri_verts.size() / 9);
    const int begin = int(prims.size());
    const bool has_n = !m.normals.empty(), has_t = !m.texcoords.empty();
    if (has_n || has_t) accel.tri_attrs.resize(size_t(first_tri) * 16, 0.0f);  // (earlier meshes without attributes: zeros, never read)
    for (size_t t = 0; t < nt; t++) {
      HostAABB box;
      for (int k = 0; k < 3; k++) {
        const float* v = &m.vertices[3 * m.indices[3 * t + k]];
        box.extend(ld3(v));
        accel.tri_verts.insert(accel.tri_verts.end(), v, v + 3);
      }
      if (has_n || has_t || !accel.tri_attrs.empty()) {
        float a[16] = {};
        for (int k = 0; k < 3; k++) {
          const uint32_t vi = m.indices[3 * t + k];
          if (has_n) memcpy(a + 3 * k, &m.normals[3 * vi], 12);
          if (has_t) memcpy(a + 9 + 2 * k, &m.texcoords[2 * vi], 8);
        }
        accel.tri_attrs.insert(accel.tri_attrs.end(), a, a + 16);
      }
      prims.push_back(make_prim(box, int(t)));
    }
    accel.bvhs.push_back(DBvh{-1, 0, 0, first_tri});
    const int bvh_id = int(accel.bvhs.size()) - 1;
    roots.push_back(root_task(begin, int(prims.size()), bvh_id));
    int32_t tmp;
    tmp = first_tri;
    memcpy(&geometries[gi].shape.f[0], &tmp, 4);
    tmp = int(nt);
    memcpy(&geometries[gi].shape.f[1], &tmp, 4);
    tmp = bvh_id;
    memcpy(&geometries[gi].shape.f[2], &tmp, 4);
    tmp = (has_n ? 1 : 0) | (has_t ? 2 : 0);
    memcpy(&geometries[gi].shape.f[4], &tmp, 4);
    // BVHImpl::get_aabb (bvh.h:50-52): the mesh as a top-level primitive is bounded by all its triangles
    HostAABB bounds;
    const BuildTask& rt = roots.back();
    bounds.lower = f3{rt.blo[0], rt.blo[1], rt.blo[2]};
    bounds.upper = f3{rt.bhi[0], rt.bhi[1], rt.bhi[2]};
    top.push_back(make_prim(bounds, int(gi)));
  }
  // 2. then every non-mesh geometry, in geometry order
  for (size_t gi = 0; gi < geometries.size(); gi++) {
    if (geometries[gi].shape.kind == SHAPE_MESH) continue;
    top.push_back(make_prim(geometry_aabb(int(gi)), int(gi)));
  }
  accel.top_prim_begin = int(prims.size());
  for (const BuildPrim& t : top) {
    float r[8] = {t.lo[0], t.lo[1], t.lo[2], 0.0f, t.hi[0], t.hi[1], t.hi[2], 0.0f};
    memcpy(&r[3], &t.index, 4);
    accel.top_boxes.insert(accel.top_boxes.end(), r, r + 8);
  }
  prims.insert(prims.end(), top.begin(), top.end());
  // the top-level root is numbered first, then the mesh roots, then level by level across all of them
  if (!top.empty()) roots.insert(roots.begin(), root_task(accel.top_prim_begin, int(prims.size()), 0));
  built_on_device = false;
  if (build_on_device >= 0 && g_device_builder && !prims.empty()) {
    // same schedule, same arithmetic, same tree, on the GPU; a failure there (no device) falls through to the host build
    // of the SAME thing -- not a different algorithm -- and is remembered in built_on_device
    std::vector<BuildPrim> dp = prims;
    FlatAccel tmp = accel;
    if (g_device_builder(dp, roots, tmp, build_on_device) == 0) {
      prims.swap(dp);
      accel = std::move(tmp);
      built_on_device = true;
    }
  }
  if (!built_on_device) build_level_synchronous(prims, roots, accel);
  accel.prims.resize(prims.size());
  for (size_t i = 0; i < prims.size(); i++) accel.prims[i] = prims[i].index;
  // leaf-ordered triangle records for the device (FlatAccel::tri_leaf)
  accel.tri_leaf.assign(size_t(accel.top_prim_begin) * 12, 0.0f);
  for (size_t b = 1; b < accel.bvhs.size(); b++) {
    // the prims entries of mesh BVH b are those between its predecessor's and its own end: walk its nodes instead
    std::vector<int> todo;
    auto emit = [&](int start, int count) {
      for (int i = start; i < start + count; i++) {
        const int tri = accel.prims[size_t(i)] + accel.bvhs[b].prim_base;
        float* r = &accel.tri_leaf[size_t(i) * 12];
        memcpy(r, &accel.tri_verts[size_t(tri) * 9], 36);
        memcpy(r + 9, &tri, 4);
      }
    };
    if (accel.bvhs[b].root_count > 0) emit(accel.bvhs[b].root_start, accel.bvhs[b].root_count);
    else todo.push_back(accel.bvhs[b].root);
    while (!todo.empty()) {
      const DNode nd = accel.nodes[size_t(todo.back())];
      todo.pop_back();
      for (int c = 0; c < 2; c++) {
        if (nd.count[c] > 0) emit(nd.child[c], nd.count[c]);
        else todo.push_back(nd.child[c]);
      }
    }
  }
  accel.built = true;
}

std::string SceneHost::describe() const {
  std::string s = "# pine scene description v1 (hexfloat); written by libpine_gpu\n";
  // materials and geometry interleaved in creation order is not needed: materials first
  for (size_t i = 0; i < nodes.size(); i++) {
    const HostNode& k = nodes[i];
    std::string l = fmt("node %d ", int(i));
    switch (k.kind) {
      case HostNode::ConstF: l += fmt("constf %a", k.f); break;
      case HostNode::Const3: l += "const3 " + hex3(k.v); break;
      case HostNode::Position: l += "position"; break;
      case HostNode::Normal: l += "normal"; break;
      case HostNode::UV: l += "uv"; break;
      case HostNode::BinF: l += fmt("binf %c %d %d", k.op, k.a, k.b); break;
      case HostNode::Bin3: l += fmt("bin3 %c %d %d", k.op, k.a, k.b); break;
      case HostNode::UnF: l += fmt("unf %c %d", k.op, k.a); break;
      case HostNode::Un3: l += fmt("un3 %c %d", k.op, k.a); break;
      case HostNode::Comp: l += fmt("comp %d %d", k.a, k.n); break;
      case HostNode::ToVec3: l += k.b < 0 ? fmt("tovec3 %d", k.a) : fmt("tovec3 %d %d %d", k.a, k.b, k.c); break;
      case HostNode::Checker: l += fmt("checker %d %a", k.a, k.f); break;
      case HostNode::Splat: l += fmt("splat %d", k.a); break;
    }
    s += l + "\n";
  }
  for (auto& d : material_describe) s += d + "\n";
  for (auto& it : item_order)
    s += (it.first == 0 ? geometries[size_t(it.second)].describe : light_describe[size_t(it.second)]) + "\n";
  if (has_env) s += env_describe + "\n";
  if (has_camera) s += camera_describe + "\n";
  return s;
}

}  // namespace pine_gpu

// ================================================================================================
// C ABI (host part)
// ================================================================================================
using namespace pine_gpu;

struct pine_gpu_scene {
  SceneHost host;
};

extern "C" {

const char* pine_gpu_last_error(void) { return g_last_error.c_str(); }
int pine_gpu_abi_version(void) { return PINE_GPU_ABI_VERSION; }

void pine_gpu_mat4_identity(float out[16]) { Mat4::identity().to_storage(out); }
void pine_gpu_mat4_translate(const float v[3], float out[16]) { mat4_translate(ld3(v)).to_storage(out); }
void pine_gpu_mat4_scale(const float v[3], float out[16]) { mat4_scale(ld3(v)).to_storage(out); }
void pine_gpu_mat4_rotate_x(float rad, float out[16]) { mat4_rotate_x(rad).to_storage(out); }
void pine_gpu_mat4_rotate_y(float rad, float out[16]) { mat4_rotate_y(rad).to_storage(out); }
void pine_gpu_mat4_rotate_z(float rad, float out[16]) { mat4_rotate_z(rad).to_storage(out); }
void pine_gpu_mat4_mul(const float a[16], const float b[16], float out[16]) {
  mat4_mul(Mat4::from_storage(a), Mat4::from_storage(b)).to_storage(out);
}
void pine_gpu_mat4_inverse(const float m[16], float out[16]) {
  mat4_inverse(Mat4::from_storage(m)).to_storage(out);
}
void pine_gpu_mat4_look_at(const float from[3], const float at[3], float out[16]) {
  mat4_look_at(ld3(from), ld3(at)).to_storage(out);
}
// q2m(a, b, c, d) fileio.cpp:127-144 (quaternion w, x, y, z -> rotation; the glTF loader's node rotations), the matrix's
// scalar constructor (row-major arguments, vecmath.h:579-581) and transpose (vecmath.h:640-648)
void pine_gpu_mat4_from_quaternion(float a, float b, float c, float d, float out[16]) {
  const float r[16] = {a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c, 0,
                       2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b, 0,
                       2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d, 0,
                       0, 0, 0, 1};
  Mat4::from_rows(r).to_storage(out);
}
void pine_gpu_mat4_from_rows(const float rows[16], float out[16]) { Mat4::from_rows(rows).to_storage(out); }
void pine_gpu_mat4_transpose(const float m[16], float out[16]) {
  const Mat4 a = Mat4::from_storage(m);
  Mat4 t{};
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) t.m[c][r] = a.m[r][c];
  t.to_storage(out);
}

pine_gpu_scene* pine_gpu_scene_create(void) { return new pine_gpu_scene(); }
void pine_gpu_scene_destroy(pine_gpu_scene* s) { delete s; }

static bool check(pine_gpu_scene* s, const void* p = (const void*)1) {
  if (!s || !p) {
    set_error("null argument");
    return false;
  }
  return true;
}
static std::string mat_name(pine_gpu_scene* s, const char* name) {
  if (name && *name) return name;
  return fmt("_anon%d", int(s->host.materials.size()));
}

int pine_gpu_scene_add_material_emissive(pine_gpu_scene* s, const char* name, const float c[3]) {
  if (!check(s, c)) return -1;
  DMaterial m{};
  m.kind = MAT_EMISSIVE;
  memcpy(m.color, c, 12);
  std::string n = mat_name(s, name);
  return s->host.add_material(n.c_str(), m, "material " + n + " emissive " + hex3(c));
}
int pine_gpu_scene_add_material_diffuse(pine_gpu_scene* s, const char* name, const float c[3]) {
  if (!check(s, c)) return -1;
  DMaterial m{};
  m.kind = MAT_DIFFUSE;
  memcpy(m.color, c, 12);
  std::string n = mat_name(s, name);
  return s->host.add_material(n.c_str(), m, "material " + n + " diffuse " + hex3(c));
}
int pine_gpu_scene_add_material_uber(pine_gpu_scene* s, const char* name, const float c[3],
                                     float roughness, float metallic, float transmission, float ior) {
  if (!check(s, c)) return -1;
  DMaterial m{};
  m.kind = MAT_UBER;
  memcpy(m.color, c, 12);
  m.roughness = roughness;
  m.metallic = metallic;
  m.transmission = transmission;
  m.ior = ior;
  std::string n = mat_name(s, name);
  return s->host.add_material(n.c_str(), m,
                              "material " + n + " uber " + hex3(c) +
                                  fmt(" %a %a %a %a", roughness, metallic, transmission, ior));
}
int pine_gpu_scene_add_material_subsurface(pine_gpu_scene* s, const char* name, const float c[3],
                                           float roughness, const float sigma_s[3]) {
  if (!check(s, c) || !check(s, sigma_s)) return -1;
  DMaterial m{};
  m.kind = MAT_SUBSURFACE;
  memcpy(m.color, c, 12);
  m.roughness = roughness;
  m.ior = 1.4f;  // material.h:110
  memcpy(m.sigma_s, sigma_s, 12);
  std::string n = mat_name(s, name);
  return s->host.add_material(
      n.c_str(), m, "material " + n + " subsurface " + hex3(c) + fmt(" %a ", roughness) + hex3(sigma_s));
}
// ---- lights other than emissive geometry (light.h:21-67, light.cpp:11-84, scene.cpp:29-46) ----
static int add_light(pine_gpu_scene* s, const DLight& L, const std::string& desc) {
  s->host.lights.push_back(L);
  s->host.light_describe.push_back(desc);
  s->host.item_order.push_back({1, int(s->host.lights.size()) - 1});
  return int(s->host.lights.size()) - 1;
}
int pine_gpu_scene_add_light_point(pine_gpu_scene* s, const float position[3], const float color[3]) {
  if (!check(s, position) || !check(s, color)) return -1;
  DLight L{};
  L.kind = LIGHT_POINT;
  memcpy(L.position, position, 12);
  memcpy(L.color, color, 12);
  return add_light(s, L, "light point " + hex3(position) + " " + hex3(color));
}
int pine_gpu_scene_add_light_spot(pine_gpu_scene* s, const float position[3], const float direction[3], const float color[3],
                                  float falloff_radian, float cutoff_additional_radian) {
  if (!check(s, position) || !check(s, direction) || !check(s, color)) return -1;
  const float pi2 = kPi * 2;  // Pi2 (math.h)
  if (falloff_radian <= 0.0f) return set_error("`SpotLight` invalid falloff angle"), -1;  // light.cpp:25-32
  if (falloff_radian > pi2) return set_error("`SpotLight` invalid falloff angle(please use radian, not degree)"), -1;
  if (cutoff_additional_radian < 0.0f) return set_error("`SpotLight` invalid cutoff angle"), -1;
  if (falloff_radian + cutoff_additional_radian > pi2) return set_error("`SpotLight` invalid cutoff angle(please use radian, not degree)"), -1;
  DLight L{};
  L.kind = LIGHT_SPOT;
  memcpy(L.position, position, 12);
  const f3 d = normalize(ld3(direction));
  L.direction[0] = d.x, L.direction[1] = d.y, L.direction[2] = d.z;
  memcpy(L.color, color, 12);
  L.falloff_cos = cosf(falloff_radian);
  L.cutoff_cos = cosf(falloff_radian + cutoff_additional_radian);
  return add_light(s, L, "light spot " + hex3(position) + " " + hex3(direction) + " " + hex3(color) +
                            fmt(" %a %a", falloff_radian, cutoff_additional_radian));
}
int pine_gpu_scene_add_light_directional(pine_gpu_scene* s, const float direction[3], const float color[3]) {
  if (!check(s, direction) || !check(s, color)) return -1;
  DLight L{};
  L.kind = LIGHT_DIRECTIONAL;
  const f3 d = normalize(ld3(direction));
  L.direction[0] = d.x, L.direction[1] = d.y, L.direction[2] = d.z;
  memcpy(L.color, color, 12);
  return add_light(s, L, "light directional " + hex3(direction) + " " + hex3(color));
}
int pine_gpu_scene_set_env_sky(pine_gpu_scene* s, const float sun_color[3]) {
  if (!check(s, sun_color)) return -1;
  DLight L{};
  L.kind = LIGHT_SKY;
  memcpy(L.color, sun_color, 12);
  s->host.env = L;
  s->host.has_env = true;
  s->host.env_describe = "envlight sky " + hex3(sun_color);
  return 0;
}

// ---- shading nodes + node-parameterised materials ----
static int add_node(pine_gpu_scene* s, HostNode n) {
  if (!check(s)) return -1;
  return s->host.add_node(n);
}
int pine_gpu_scene_node_constf(pine_gpu_scene* s, float v) {
  HostNode n;
  n.kind = HostNode::ConstF;
  n.f = v;
  return add_node(s, n);
}
int pine_gpu_scene_node_const3(pine_gpu_scene* s, const float v[3]) {
  if (!check(s, v)) return -1;
  HostNode n;
  n.kind = HostNode::Const3;
  n.v = ld3(v);
  return add_node(s, n);
}
int pine_gpu_scene_node_input(pine_gpu_scene* s, int which) {
  if (which < 0 || which > 2) {
    if (check(s)) set_error("node input: 0 Position, 1 Normal, 2 UV");
    return -1;
  }
  HostNode n;
  n.kind = which == 0 ? HostNode::Position : which == 1 ? HostNode::Normal : HostNode::UV;
  return add_node(s, n);
}
int pine_gpu_scene_node_is_vec3(pine_gpu_scene* s, int id) {
  if (!check(s)) return -1;
  if (id < 0 || id >= int(s->host.nodes.size())) {
    set_error("node id out of range");
    return -1;
  }
  return s->host.nodes[size_t(id)].is_vec3() ? 1 : 0;
}
int pine_gpu_scene_node_binary(pine_gpu_scene* s, int op, int a, int b) {
  if (!check(s)) return -1;
  const int va = pine_gpu_scene_node_is_vec3(s, a), vb = pine_gpu_scene_node_is_vec3(s, b);
  if (va < 0 || vb < 0) return -1;
  if (va != vb) {
    set_error("node binary: operands must both be Nodef or both Node3f (wrap the Nodef with node_splat)");
    return -1;
  }
  HostNode n;
  n.kind = va ? HostNode::Bin3 : HostNode::BinF;
  n.op = char(op);
  n.a = a;
  n.b = b;
  return add_node(s, n);
}
int pine_gpu_scene_node_unary(pine_gpu_scene* s, int op, int a) {
  if (!check(s)) return -1;
  const int va = pine_gpu_scene_node_is_vec3(s, a);
  if (va < 0) return -1;
  HostNode n;
  n.kind = va ? HostNode::Un3 : HostNode::UnF;
  n.op = char(op);
  n.a = a;
  return add_node(s, n);
}
int pine_gpu_scene_node_component(pine_gpu_scene* s, int a, int comp) {
  HostNode n;
  n.kind = HostNode::Comp;
  n.a = a;
  n.n = comp;
  return add_node(s, n);
}
int pine_gpu_scene_node_to_vec3(pine_gpu_scene* s, int x, int y, int z) {
  HostNode n;
  n.kind = HostNode::ToVec3;
  n.a = x;
  n.b = y;
  n.c = z;
  return add_node(s, n);
}
int pine_gpu_scene_node_checkerboard(pine_gpu_scene* s, int p, float ratio) {
  HostNode n;
  n.kind = HostNode::Checker;
  n.a = p;
  n.f = ratio;
  return add_node(s, n);
}
int pine_gpu_scene_node_splat(pine_gpu_scene* s, int a) {
  HostNode n;
  n.kind = HostNode::Splat;
  n.a = a;
  return add_node(s, n);
}
static int add_node_material(pine_gpu_scene* s, const char* name, int kind, const int ids[4], const bool used[4],
                             float ior, const std::string& desc_kind, const std::string& desc_tail) {
  if (!check(s)) return -1;
  for (int k = 0; k < 4; k++) {
    if (!used[k]) continue;
    const int v = pine_gpu_scene_node_is_vec3(s, ids[k]);
    if (v < 0) return -1;
    if ((k == 0) != (v == 1)) {
      set_error(k == 0 ? "material: albedo must be a Node3f" : "material: roughness / metallic / transmission / ior must be a Nodef");
      return -1;
    }
  }
  DMaterial m{};
  m.kind = kind;
  m.ior = ior;
  std::string n = mat_name(s, name);
  std::string desc = "material " + n + " " + desc_kind;
  for (int k = 0; k < 4; k++)
    if (used[k]) desc += fmt(" %d", ids[k]);
  desc += desc_tail;
  const int id = s->host.add_material(n.c_str(), m, desc);
  for (int k = 0; k < 4; k++) s->host.material_nodes[size_t(id)].id[k] = used[k] ? ids[k] : -1;
  return id;
}
int pine_gpu_scene_add_material_diffuse_n(pine_gpu_scene* s, const char* name, int albedo) {
  const int ids[4] = {albedo, -1, -1, -1};
  const bool used[4] = {true, false, false, false};
  return add_node_material(s, name, MAT_DIFFUSE, ids, used, 1.45f, "diffuse_n", "");
}
int pine_gpu_scene_add_material_uber_n(pine_gpu_scene* s, const char* name, int albedo, int roughness, int metallic,
                                       int transmission, float ior) {
  const int ids[4] = {albedo, roughness, metallic, transmission};
  const bool used[4] = {true, true, true, true};
  return add_node_material(s, name, MAT_UBER, ids, used, ior, "uber_n", fmt(" %a", ior));
}
int pine_gpu_scene_add_material_metal(pine_gpu_scene* s, const char* name, int albedo, int roughness) {
  const int ids[4] = {albedo, roughness, -1, -1};
  const bool used[4] = {true, true, false, false};
  return add_node_material(s, name, MAT_METAL, ids, used, 1.0f, "metal", "");
}
int pine_gpu_scene_add_material_glossy(pine_gpu_scene* s, const char* name, int albedo, int roughness, int ior) {
  const int ids[4] = {albedo, roughness, -1, ior};
  const bool used[4] = {true, true, false, true};
  return add_node_material(s, name, MAT_GLOSSY, ids, used, 1.4f, "glossy", "");
}
int pine_gpu_scene_add_material_glass(pine_gpu_scene* s, const char* name, int albedo, int roughness, int ior) {
  const int ids[4] = {albedo, roughness, -1, ior};
  const bool used[4] = {true, true, false, true};
  return add_node_material(s, name, MAT_GLASS, ids, used, 1.4f, "glass", "");
}

int pine_gpu_scene_find_material(pine_gpu_scene* s, const char* name) {
  if (!check(s, name)) return -1;
  int id = s->host.find_material(name);
  if (id < 0) set_error(std::string("Can't find material `") + name + "`");  // scene.cpp:53
  return id;
}

static std::string matref(pine_gpu_scene* s, int material) {
  if (material < 0 || material >= int(s->host.material_names.size())) return "?";
  return s->host.material_names[material];
}

int pine_gpu_scene_add_rect(pine_gpu_scene* s, const float position[3], const float ex_[3],
                            const float ey_[3], int flip, int material) {
  if (!check(s, position) || !check(s, ex_) || !check(s, ey_)) return -1;
  HostGeometry g{};
  DShape& d = g.shape;
  d.kind = SHAPE_RECT;
  d.material = material;
  // Rect::Rect geometry.cpp:255-267
  f3 ex = normalize(ld3(ex_)), ey = normalize(ld3(ey_));
  f3 n = normalize(cross(ex, ey)) * float(flip ? -1 : 1);
  float lx = length(ld3(ex_)), ly = length(ld3(ey_));
  f3 rx = ex / lx, ry = ey / ly;
  if (std::abs(length(n) - 1.0f) > 1e-6f) {
    set_error("`Rect` has degenerated shape");
    return -1;
  }
  float* f = d.f;
  memcpy(f, position, 12);
  f[3] = ex.x, f[4] = ex.y, f[5] = ex.z;
  f[6] = ey.x, f[7] = ey.y, f[8] = ey.z;
  f[9] = n.x, f[10] = n.y, f[11] = n.z;
  f[12] = lx, f[13] = ly;
  f[14] = rx.x, f[15] = rx.y, f[16] = rx.z;
  f[17] = ry.x, f[18] = ry.y, f[19] = ry.z;
  f[20] = lx * ly;  // Rect::area geometry.h:97
  g.describe = "shape rect " + matref(s, material) + " " + hex3(position) + " " + hex3(ex_) + " " +
               hex3(ey_) + fmt(" %d", flip ? 1 : 0);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_aabb(pine_gpu_scene* s, const float lo[3], const float hi[3], int material) {
  if (!check(s, lo) || !check(s, hi)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_AABB;
  g.shape.material = material;
  memcpy(g.shape.f, lo, 12);
  memcpy(g.shape.f + 3, hi, 12);
  g.describe = "shape box " + matref(s, material) + " " + hex3(lo) + " " + hex3(hi);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_obb(pine_gpu_scene* s, const float lo[3], const float hi[3],
                           const float m[16], int material) {
  if (!check(s, lo) || !check(s, hi) || !check(s, m)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_OBB;
  g.shape.material = material;
  float* f = g.shape.f;
  memcpy(f, lo, 12);
  memcpy(f + 3, hi, 12);
  Mat4 M = Mat4::from_storage(m);
  Mat4 Mi = mat4_inverse(M);  // OBB::OBB bbox.cpp:144
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 3; r++) {
      f[6 + c * 3 + r] = M.m[c][r];
      f[18 + c * 3 + r] = Mi.m[c][r];
    }
  // describe with row-major constructor arguments (what `mat4(...)` takes)
  std::string ms;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) ms += fmt(" %a", M.m[c][r]);
  g.describe = "shape obb " + matref(s, material) + " " + hex3(lo) + " " + hex3(hi) + ms;
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_sphere(pine_gpu_scene* s, const float c[3], float r, int material) {
  if (!check(s, c)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_SPHERE;
  g.shape.material = material;
  memcpy(g.shape.f, c, 12);
  g.shape.f[3] = r;
  g.describe = "shape sphere " + matref(s, material) + " " + hex3(c) + fmt(" %a", r);
  return s->host.add_geometry(std::move(g));
}
static bool fill_disk(float* f, f3 position, f3 normal, float r) {  // Disk::Disk geometry.cpp:123-127
  if (r < 0.0f) {
    set_error("`Disk` can't have negative radius");
    return false;
  }
  f3 n = normalize(normal);
  if (length(n) == 0.0f) {
    set_error("`Disk` can't have degenerated normal");
    return false;
  }
  f3 u, v;
  coordinate_system(n, u, v);
  f[0] = position.x, f[1] = position.y, f[2] = position.z;
  f[3] = n.x, f[4] = n.y, f[5] = n.z;
  f[6] = u.x, f[7] = u.y, f[8] = u.z;
  f[9] = v.x, f[10] = v.y, f[11] = v.z;
  f[12] = r;
  f[13] = kPi * r * r;  // Disk::area geometry.h:54
  return true;
}
int pine_gpu_scene_add_disk(pine_gpu_scene* s, const float p[3], const float n[3], float r, int material) {
  if (!check(s, p) || !check(s, n)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_DISK;
  g.shape.material = material;
  if (!fill_disk(g.shape.f, ld3(p), ld3(n), r)) return -1;
  g.describe = "shape disk " + matref(s, material) + " " + hex3(p) + " " + hex3(n) + fmt(" %a", r);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_cone(pine_gpu_scene* s, const float p_[3], const float n_[3], float r, float h,
                            int material) {
  if (!check(s, p_) || !check(s, n_)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_CONE;
  g.shape.material = material;
  // Cone::Cone geometry.cpp:409-414: apex = p + n*h (n as given), n normalised afterwards
  float bottom[14];
  if (!fill_disk(bottom, ld3(p_), ld3(n_), r)) return -1;
  f3 apex = ld3(p_) + ld3(n_) * h;
  f3 n = normalize(ld3(n_));
  float A2 = sqr(r / h) + 1;
  float A = std::sqrt(A2);
  float S = r / std::sqrt(r * r + h * h);
  float area = std::sqrt(r * r + h * h) * kPi * r + bottom[13];  // geometry.h:132
  float* f = g.shape.f;
  f[0] = apex.x, f[1] = apex.y, f[2] = apex.z;
  f[3] = n.x, f[4] = n.y, f[5] = n.z;
  f[6] = r, f[7] = h, f[8] = A, f[9] = A2, f[10] = S, f[11] = area;
  f[12] = p_[0], f[13] = p_[1], f[14] = p_[2];  // bottom disk centre (AABB only)
  g.describe = "shape cone " + matref(s, material) + " " + hex3(p_) + " " + hex3(n_) + fmt(" %a %a", r, h);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_plane(pine_gpu_scene* s, const float p_[3], const float n_[3], int material) {
  if (!check(s, p_) || !check(s, n_)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_PLANE;
  g.shape.material = material;
  f3 n = normalize(ld3(n_));  // Plane::Plane geometry.cpp:31-34
  if (length(n) == 0.0f) {
    set_error("`Plane` can't have degenerated normal");
    return -1;
  }
  f3 u, v;
  coordinate_system(n, u, v);
  float* f = g.shape.f;
  memcpy(f, p_, 12);
  f[3] = n.x, f[4] = n.y, f[5] = n.z;
  f[6] = u.x, f[7] = u.y, f[8] = u.z;
  f[9] = v.x, f[10] = v.y, f[11] = v.z;
  g.describe = "shape plane " + matref(s, material) + " " + hex3(p_) + " " + hex3(n_);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_line(pine_gpu_scene* s, const float p0_[3], const float p1_[3], float thickness,
                            int material) {
  if (!check(s, p0_) || !check(s, p1_)) return -1;
  if (thickness <= 0.0f) {  // Line::Line geometry.cpp:171-179
    set_error("`Line` should have positive thickness");
    return -1;
  }
  f3 p0 = ld3(p0_), p1 = ld3(p1_);
  if (p0.x == p1.x && p0.y == p1.y && p0.z == p1.z) {
    set_error("`Line` shouldn't have identical begin and end point");
    return -1;
  }
  HostGeometry g{};
  g.shape.kind = SHAPE_LINE;
  g.shape.material = material;
  m3 tbn = coordinate_system(normalize(p1 - p0));
  float len = length(p1 - p0);
  float* f = g.shape.f;
  memcpy(f, p0_, 12);
  memcpy(f + 3, p1_, 12);
  f[6] = tbn.x.x, f[7] = tbn.x.y, f[8] = tbn.x.z;
  f[9] = tbn.y.x, f[10] = tbn.y.y, f[11] = tbn.y.z;
  f[12] = tbn.z.x, f[13] = tbn.z.y, f[14] = tbn.z.z;
  f[15] = thickness;
  f[16] = thickness * 2 * kPi * len;  // Line::area geometry.h:70
  g.describe = "shape line " + matref(s, material) + " " + hex3(p0_) + " " + hex3(p1_) + fmt(" %a", thickness);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_cylinder(pine_gpu_scene* s, const float p0_[3], const float p1_[3], float r,
                                int material) {
  if (!check(s, p0_) || !check(s, p1_)) return -1;
  if (material >= 0 && material < int(s->host.materials.size()) &&
      s->host.materials[size_t(material)].kind == MAT_EMISSIVE) {
    // Cylinder::sample / pdf / area are PINE_UNREACHABLE in the reference (geometry.h:148-150)
    set_error("`Cylinder` can't be emissive: the reference has no sampling routine for it");
    return -1;
  }
  if (r < 0.0f) {  // the two cap Disks of the constructor (geometry.h:141, geometry.cpp:125)
    set_error("`Disk` can't have negative radius");
    return -1;
  }
  HostGeometry g{};
  g.shape.kind = SHAPE_CYLINDER;
  g.shape.material = material;
  f3 n = normalize(ld3(p1_) - ld3(p0_));  // geometry.h:140-141
  if (length(n) == 0.0f) {
    set_error("`Disk` can't have degenerated normal");
    return -1;
  }
  float* f = g.shape.f;
  memcpy(f, p0_, 12);
  memcpy(f + 3, p1_, 12);
  f[6] = n.x, f[7] = n.y, f[8] = n.z;
  f[9] = r;
  g.describe = "shape cylinder " + matref(s, material) + " " + hex3(p0_) + " " + hex3(p1_) + fmt(" %a", r);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_triangle(pine_gpu_scene* s, const float v0_[3], const float v1_[3], const float v2_[3],
                                int material) {
  if (!check(s, v0_) || !check(s, v1_) || !check(s, v2_)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_TRIANGLE;
  g.shape.material = material;
  f3 v0 = ld3(v0_), v1 = ld3(v1_), v2 = ld3(v2_);
  f3 n = normalize(cross(v0 - v1, v0 - v2));  // Triangle(v0,v1,v2) geometry.cpp:528-531
  if (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f) n = mk3(0.0f, 0.0f, 1.0f);
  float* f = g.shape.f;
  memcpy(f, v0_, 12);
  memcpy(f + 3, v1_, 12);
  memcpy(f + 6, v2_, 12);
  f[9] = n.x, f[10] = n.y, f[11] = n.z;
  f[12] = length(cross(v1 - v0, v2 - v0)) / 2;  // Triangle::area geometry.h:112
  g.describe = "shape triangle " + matref(s, material) + " " + hex3(v0_) + " " + hex3(v1_) + " " + hex3(v2_);
  return s->host.add_geometry(std::move(g));
}
// ---- state-level entry points: the reference's CONSTRUCTED objects, member for member ---------------------------
// A binding inside pine's tree walks an already-built pine::Scene; the members a shape keeps (normalised axes,
// derived lengths) do not invert to its constructor arguments bit-exactly (normalize(normalize(n)) != normalize(n) in
// floats), so these take the stored state as it is.  Shapes whose members ARE their constructor arguments (Sphere, Box,
// Line, Cylinder, Mesh) go through the constructor-level calls above.
static std::string hexn(const float* v, int n) {
  std::string s;
  for (int i = 0; i < n; i++) s += fmt(i ? " %a" : "%a", v[i]);
  return s;
}
int pine_gpu_scene_add_rect_state(pine_gpu_scene* s, const float position[3], const float ex[3], const float ey[3], const float n[3],
                                  float lx, float ly, const float rx[3], const float ry[3], int material) {
  if (!check(s, position) || !check(s, ex) || !check(s, ey) || !check(s, n) || !check(s, rx) || !check(s, ry)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_RECT;
  g.shape.material = material;
  float* f = g.shape.f;  // members of Rect, geometry.h:92-96
  memcpy(f, position, 12), memcpy(f + 3, ex, 12), memcpy(f + 6, ey, 12), memcpy(f + 9, n, 12);
  f[12] = lx, f[13] = ly;
  memcpy(f + 14, rx, 12), memcpy(f + 17, ry, 12);
  f[20] = lx * ly;  // Rect::area geometry.h:90
  g.describe = "shape rect_state " + matref(s, material) + " " + hexn(f, 20);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_disk_state(pine_gpu_scene* s, const float position[3], const float n[3], const float u[3], const float v[3],
                                  float r, int material) {
  if (!check(s, position) || !check(s, n) || !check(s, u) || !check(s, v)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_DISK;
  g.shape.material = material;
  float* f = g.shape.f;  // members of Disk, geometry.h:56-60
  memcpy(f, position, 12), memcpy(f + 3, n, 12), memcpy(f + 6, u, 12), memcpy(f + 9, v, 12);
  f[12] = r;
  f[13] = kPi * r * r;  // Disk::area geometry.h:54
  g.describe = "shape disk_state " + matref(s, material) + " " + hexn(f, 13);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_plane_state(pine_gpu_scene* s, const float position[3], const float n[3], const float u[3], const float v[3],
                                   int material) {
  if (!check(s, position) || !check(s, n) || !check(s, u) || !check(s, v)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_PLANE;
  g.shape.material = material;
  float* f = g.shape.f;  // members of Plane, geometry.h:23-25
  memcpy(f, position, 12), memcpy(f + 3, n, 12), memcpy(f + 6, u, 12), memcpy(f + 9, v, 12);
  g.describe = "shape plane_state " + matref(s, material) + " " + hexn(f, 12);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_cone_state(pine_gpu_scene* s, const float apex[3], const float n[3], float r, float h, float A, float A2, float S,
                                  const float bottom_position[3], int material) {
  if (!check(s, apex) || !check(s, n) || !check(s, bottom_position)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_CONE;
  g.shape.material = material;
  float* f = g.shape.f;  // members of Cone, geometry.h:134-140 (bottom: the Disk's centre, its radius is r)
  memcpy(f, apex, 12), memcpy(f + 3, n, 12);
  f[6] = r, f[7] = h, f[8] = A, f[9] = A2, f[10] = S;
  f[11] = std::sqrt(r * r + h * h) * kPi * r + kPi * r * r;  // Cone::area geometry.h:132
  memcpy(f + 12, bottom_position, 12);
  g.describe = "shape cone_state " + matref(s, material) + " " + hexn(f, 11) + " " + hexn(f + 12, 3);
  return s->host.add_geometry(std::move(g));
}
int pine_gpu_scene_add_triangle_state(pine_gpu_scene* s, const float v0[3], const float v1[3], const float v2[3], const float n[3],
                                      int material) {
  if (!check(s, v0) || !check(s, v1) || !check(s, v2) || !check(s, n)) return -1;
  HostGeometry g{};
  g.shape.kind = SHAPE_TRIANGLE;
  g.shape.material = material;
  float* f = g.shape.f;  // members of Triangle, geometry.h:115-117
  memcpy(f, v0, 12), memcpy(f + 3, v1, 12), memcpy(f + 6, v2, 12), memcpy(f + 9, n, 12);
  f[12] = length(cross(ld3(v1) - ld3(v0), ld3(v2) - ld3(v0))) / 2;  // Triangle::area geometry.h:112
  g.describe = "shape triangle_state " + matref(s, material) + " " + hexn(f, 12);
  return s->host.add_geometry(std::move(g));
}

int pine_gpu_scene_add_mesh(pine_gpu_scene* s, const float* vertices, int nv, const uint32_t* indices,
                            int nt, int material) {
  if (!check(s, vertices) || !check(s, indices)) return -1;
  for (int i = 0; i < 3 * nt; i++)
    if (indices[i] >= uint32_t(nv)) {
      set_error("mesh index out of range");
      return -1;
    }
  HostMesh m;
  m.vertices.assign(vertices, vertices + 3 * size_t(nv));
  m.indices.assign(indices, indices + 3 * size_t(nt));
  HostGeometry g{};
  g.shape.kind = SHAPE_MESH;
  g.shape.material = material;
  if (nt > 0) {  // Mesh::area geometry.h:167-169: first triangle's area x count
    f3 a = ld3(&vertices[3 * indices[0]]), b = ld3(&vertices[3 * indices[1]]), c = ld3(&vertices[3 * indices[2]]);
    g.shape.f[3] = length(cross(b - a, c - a)) / 2 * float(size_t(nt));
  }
  std::string d = "shape mesh " + matref(s, material) + fmt(" %d %d", nv, nt);
  for (int i = 0; i < 3 * nv; i++) d += fmt(" %a", vertices[i]);
  for (int i = 0; i < 3 * nt; i++) d += fmt(" %u", indices[i]);
  g.describe = std::move(d);
  s->host.meshes.push_back(std::move(m));
  g.mesh = int(s->host.meshes.size()) - 1;
  return s->host.add_geometry(std::move(g));
}

// Mesh(vertices, indices, texcoords, normals) geometry.cpp:596-604: per-vertex normals and / or texture coordinates,
// as a glTF import produces them (fileio.cpp:146-311).  Either may be null.
int pine_gpu_scene_add_mesh_full(pine_gpu_scene* s, const float* vertices, int nv, const uint32_t* indices, int nt,
                                 const float* normals, const float* texcoords, int material) {
  const int id = pine_gpu_scene_add_mesh(s, vertices, nv, indices, nt, material);
  if (id < 0 || (!normals && !texcoords)) return id;
  HostGeometry& g = s->host.geometries[size_t(id)];
  HostMesh& m = s->host.meshes[size_t(g.mesh)];
  std::string d = "shape mesh_full " + g.describe.substr(strlen("shape mesh "));
  d += fmt(" %d %d", normals ? 1 : 0, texcoords ? 1 : 0);
  if (normals) {
    m.normals.assign(normals, normals + 3 * size_t(nv));
    for (int i = 0; i < 3 * nv; i++) d += fmt(" %a", normals[i]);
  }
  if (texcoords) {
    m.texcoords.assign(texcoords, texcoords + 2 * size_t(nv));
    for (int i = 0; i < 2 * nv; i++) d += fmt(" %a", texcoords[i]);
  }
  g.describe = std::move(d);
  return id;
}

// Mesh::apply(mat4) geometry.cpp:647-653, in place: v = m * v (affine point transform, vecmath.h:705-707);
// n = normalize(transpose(inverse(mat3(m))) * n).  `normals` may be null.
int pine_gpu_mesh_apply(float* vertices, int nv, float* normals, const float m_[16]) {
  if (!vertices || !m_ || nv < 0) {
    set_error("bad argument");
    return -1;
  }
  const Mat4 M = Mat4::from_storage(m_);
  const f3 cx{M.m[0][0], M.m[0][1], M.m[0][2]}, cy{M.m[1][0], M.m[1][1], M.m[1][2]}, cz{M.m[2][0], M.m[2][1], M.m[2][2]},
      cw{M.m[3][0], M.m[3][1], M.m[3][2]};
  for (int i = 0; i < nv; i++) {
    const f3 q = ld3(vertices + 3 * i);
    const f3 v = cx * q.x + cy * q.y + cz * q.z + cw;  // operator*(mat4, vec3) vecmath.h:705-707
    vertices[3 * i] = v.x, vertices[3 * i + 1] = v.y, vertices[3 * i + 2] = v.z;
  }
  if (normals) {
    const m3 a{cx, cy, cz};  // mat3(mat4): the upper-left 3x3
    const m3 t = transpose(inverse(a));
    for (int i = 0; i < nv; i++) {
      const f3 n = normalize(mul(t, ld3(normals + 3 * i)));
      normals[3 * i] = n.x, normals[3 * i + 1] = n.y, normals[3 * i + 2] = n.z;
    }
  }
  return 0;
}

int pine_gpu_scene_set_camera_thinlens(pine_gpu_scene* s, int w, int h, int tonemapper, const float from[3],
                                       const float to[3], float fov, float len_radius, float focus) {
  if (!check(s, from) || !check(s, to)) return -1;
  if (w <= 0 || h <= 0) {
    set_error("film size must be positive");
    return -1;
  }
  DCamera& c = s->host.camera;
  memcpy(c.position, from, 12);
  Mat4 L = mat4_look_at(ld3(from), ld3(to));  // camera.cpp:9-10
  for (int col = 0; col < 3; col++)
    for (int r = 0; r < 3; r++) c.c2w[col * 3 + r] = L.m[col][r];
  float aspect = float(w) / h;  // Film::aspect film.h:32
  c.fov2d[0] = fov * aspect;
  c.fov2d[1] = fov;
  c.len_radius = len_radius;
  c.focus_distance = focus;
  c.W = w;
  c.H = h;
  s->host.has_camera = true;
  s->host.tonemapper = tonemapper;
  s->host.camera_describe = fmt("camera thinlens %d %d ", w, h) + hex3(from) + " " + hex3(to) +
                            fmt(" %a %a %a", fov, len_radius, focus);
  return 0;
}

// state-level form: the members of a constructed ThinLenCamera (camera.h:21-26), c2w as 9 floats, columns x, y, z
int pine_gpu_scene_set_camera_thinlens_state(pine_gpu_scene* s, int w, int h, int tonemapper, const float position[3],
                                             const float c2w[9], const float fov2d[2], float len_radius, float focus) {
  if (!check(s, position) || !check(s, c2w) || !check(s, fov2d)) return -1;
  if (w <= 0 || h <= 0) {
    set_error("film size must be positive");
    return -1;
  }
  DCamera& c = s->host.camera;
  memcpy(c.position, position, 12);
  memcpy(c.c2w, c2w, 36);
  c.fov2d[0] = fov2d[0], c.fov2d[1] = fov2d[1];
  c.len_radius = len_radius;
  c.focus_distance = focus;
  c.W = w;
  c.H = h;
  s->host.has_camera = true;
  s->host.tonemapper = tonemapper;
  s->host.camera_describe = fmt("camera thinlens_state %d %d ", w, h) + hex3(position) + " " + hexn(c2w, 9) + " " + hexn(fov2d, 2) +
                            fmt(" %a %a", len_radius, focus);
  return 0;
}

// test hook: the 128-byte device record of geometry `index` (30 floats, kind, material) and the camera record
int pine_gpu_scene_shape_record(pine_gpu_scene* s, int index, float out[32]) {
  if (!check(s, out)) return -1;
  if (index < 0 || size_t(index) >= s->host.geometries.size()) {
    set_error("geometry index out of range");
    return -1;
  }
  memcpy(out, &s->host.geometries[size_t(index)].shape, sizeof(DShape));
  return 0;
}
int pine_gpu_scene_camera_record(pine_gpu_scene* s, float out[20]) {
  if (!check(s, out)) return -1;
  static_assert(sizeof(DCamera) <= 80, "DCamera grew");
  memset(out, 0, 80);
  memcpy(out, &s->host.camera, sizeof(DCamera));
  return 0;
}

int pine_gpu_shard_of_pixel(int film_w, int x, int y, int world) {
  if (film_w <= 0 || x < 0 || y < 0 || world < 1) {
    set_error("bad argument");
    return -1;
  }
  const int tiles_x = (film_w + 7) / 8;
  return ((y / 8) * tiles_x + x / 8) % world;
}

int64_t pine_gpu_scene_describe(pine_gpu_scene* s, char* buf, int64_t capacity) {
  if (!check(s)) return -1;
  std::string d = s->host.describe();
  if (buf && capacity > 0) {
    int64_t n = std::min<int64_t>(capacity - 1, int64_t(d.size()));
    memcpy(buf, d.data(), size_t(n));
    buf[n] = 0;
  }
  return int64_t(d.size());
}

int pine_gpu_scene_build_accel(pine_gpu_scene* s) {
  if (!check(s)) return -1;
  s->host.build_on_device = -1;
  s->host.build_accel();
  return int(s->host.accel.nodes.size());
}
int pine_gpu_scene_build_accel_device(pine_gpu_scene* s, int device) {
  if (!check(s)) return -1;
  if (device < 0) {
    set_error("bad device ordinal");
    return -1;
  }
  s->host.build_on_device = device;
  s->host.build_accel();
  s->host.build_on_device = -1;
  if (!s->host.built_on_device && !s->host.geometries.empty()) {
    set_error("the BVH build did not run on the device (no HIP device?); the host built the same tree");
    return -1;
  }
  return int(s->host.accel.nodes.size());
}
int64_t pine_gpu_scene_accel_dump(pine_gpu_scene* s, void* nodes_out, int64_t node_cap, int32_t* prims_out,
                                  int64_t prim_cap) {
  if (!check(s)) return -1;
  if (!s->host.accel.built) s->host.build_accel();
  const FlatAccel& a = s->host.accel;
  if (nodes_out && node_cap >= int64_t(a.nodes.size() * sizeof(DNode)))
    memcpy(nodes_out, a.nodes.data(), a.nodes.size() * sizeof(DNode));
  if (prims_out && prim_cap >= int64_t(a.prims.size()))
    memcpy(prims_out, a.prims.data(), a.prims.size() * 4);
  return int64_t(a.prims.size());
}

int pine_gpu_scene_accel_bvhs(pine_gpu_scene* s, int32_t* out, int64_t cap) {
  if (!check(s) || !out) return -1;
  if (!s->host.accel.built) s->host.build_accel();
  const FlatAccel& a = s->host.accel;
  if (cap < int64_t(a.bvhs.size()) * 5) {
    set_error("capacity too small");
    return -1;
  }
  std::vector<int> mesh_geom(a.bvhs.size(), -1);
  for (size_t g = 0; g < s->host.geometries.size(); g++) {
    const DShape& sh = s->host.geometries[g].shape;
    if (sh.kind == SHAPE_MESH) {
      int b;
      memcpy(&b, &sh.f[2], 4);
      if (b > 0 && size_t(b) < a.bvhs.size()) mesh_geom[size_t(b)] = int(g);
    }
  }
  for (size_t b = 0; b < a.bvhs.size(); b++) {
    out[5 * b + 0] = a.bvhs[b].root, out[5 * b + 1] = a.bvhs[b].root_start, out[5 * b + 2] = a.bvhs[b].root_count;
    out[5 * b + 3] = a.bvhs[b].prim_base, out[5 * b + 4] = mesh_geom[b];
  }
  return int(a.bvhs.size());
}

int64_t pine_gpu_scene_specialized_source(pine_gpu_scene* s, char* out, int64_t cap) {
  if (!check(s)) return -1;
  if (!s->host.accel.built) s->host.build_accel();
  const FlatAccel& a = s->host.accel;
  std::vector<DShape> shapes;
  for (auto& g : s->host.geometries) shapes.push_back(g.shape);
  std::vector<int> words = a.prims;
  for (size_t i = size_t(a.top_prim_begin); i < words.size(); i++) {
    const DShape& sh = shapes[size_t(words[i])];
    words[i] |= (s->host.materials[size_t(sh.material)].kind == MAT_EMISSIVE ? kPrimEmissiveBit : 0) | (sh.kind << kPrimKindShift);
  }
  // (no mesh: the whole scene; exactly one mesh: its top level, for the traversal-stage variants)
  const std::string text = a.top_prim_begin == 0 ? generate_baked_scene(a, shapes, words) : a.bvhs.size() == 2 ? generate_baked_scene(a, shapes, words, true) : std::string();
  if (out && cap > int64_t(text.size())) memcpy(out, text.c_str(), text.size() + 1);
  return int64_t(text.size());
}

int pine_gpu_test_specialize_compile(pine_gpu_scene* s, uint32_t features, int ctx, const char* arch, char* path_out, int64_t cap) {
  if (!arch || (s && !check(s))) return -1;
  std::string text;  // (no scene: level 1, the feature set alone with the generic traversal)
  if (s) {
    const int64_t n = pine_gpu_scene_specialized_source(s, nullptr, 0);
    if (n <= 0) {
      if (n == 0) set_error("the scene does not qualify for specialisation");
      return -1;
    }
    text.assign(size_t(n) + 1, '\0');
    pine_gpu_scene_specialized_source(s, &text[0], n + 1);
    text.resize(size_t(n));
    features |= 1u << 17;  // F_BAKED (pine_device.h)
  }
  std::string path, err;
  bool hit = false;
  if (!compile_baked_kernel(text, features, ctx, arch, path, err, &hit)) {
    set_error(err);
    return -1;
  }
  if (path_out && cap > int64_t(path.size())) memcpy(path_out, path.c_str(), path.size() + 1);
  return hit ? 1 : 0;
}

// Film::finalize + tone mapping + to_uint8_array (film.cpp:21-27,66-68; color.cpp:6-23;
// fileio.cpp:42-54 with flip_y = true, apply_gamma = true)
int pine_gpu_film_finalize_u8(const float* film, int w, int h, int tonemapper, uint8_t* out) {
  if (!film || !out || w <= 0 || h <= 0) {
    set_error("bad argument");
    return -1;
  }
  auto unch = [](float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return (x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F) - E / F;
  };
  const float white = unch(11.2f);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int ys = h - 1 - y;  // y flip
      const float* p = film + (size_t(ys) * w + x) * 4;
      float v[4];
      for (int c = 0; c < 3; c++) {
        float t;
        if (tonemapper == 1) {  // ACES color.cpp:14-23
          const float a = 2.51f, b = 0.03f, cc = 2.43f, d = 0.59f, e = 0.14f;
          float xx = p[c];
          float r = (xx * (a * xx + b)) / (xx * (cc * xx + d) + e);
          t = pmin(pmax(r, 0.0f), 1.0f);
        } else {  // Uncharted2 color.cpp:6-13
          t = unch(p[c] * 2.0f) * 1.0f / white;
        }
        v[c] = t;
      }
      v[3] = 1.0f;  // finalize sets w = 1
      for (int c = 0; c < 4; c++) {
        float g = std::pow(v[c], 1 / 2.2f);
        out[(size_t(y) * w + x) * 4 + c] = uint8_t(pmin(pmax(g * 256.0f, 0.0f), 255.0f));
      }
    }
  return 0;
}

}  // extern "C"

// accessor used by the GPU side (pine_kernels.hip)
namespace pine_gpu {
SceneHost& scene_host(pine_gpu_scene* s) { return s->host; }
}  // namespace pine_gpu
