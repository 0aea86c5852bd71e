// pine_amd/host/pine.hpp -- C++ host facade over the C ABI (include/pine_gpu.h).
//
// Keeps the names a pine user writes (src/pine/core/program_context.cpp:23-125 and the *_context
// functions): Scene / add / set, Diffuse, Emissive, Uber, Subsurface, Rect, Box, Sphere, Disk, Cone,
// Mesh, Film, Uncharted2, ThinLenCamera, BlueSampler, PathIntegrator(sampler, depth).render(scene),
// scene.camera.film().save(...).  Header-only; link with -lpine_gpu.  Errors throw pine::Error where
// the reference would SEVERE()/abort (src/pine/core/log.h:45-51).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/pine_gpu.h"
#include "gltf_import.hpp"

namespace pine {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline int check(int rc, const char* what) {
  if (rc < 0) throw Error(std::string(what) + ": " + pine_gpu_last_error());
  return rc;
}

struct vec3 {
  float x = 0, y = 0, z = 0;
  vec3() = default;
  vec3(float x, float y, float z) : x(x), y(y), z(z) {}
  const float* data() const { return &x; }
};
inline vec3 operator*(float s, vec3 v) { return {s * v.x, s * v.y, s * v.z}; }
struct vec2i {
  int x = 0, y = 0;
};

struct mat4 {  // storage order of the reference: m[c*4 + r]
  float m[16];
  mat4() { pine_gpu_mat4_identity(m); }
  friend mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r;
    pine_gpu_mat4_mul(a.m, b.m, r.m);
    return r;
  }
};
inline mat4 translate(vec3 v) { mat4 r; pine_gpu_mat4_translate(v.data(), r.m); return r; }
inline mat4 scale(vec3 v) { mat4 r; pine_gpu_mat4_scale(v.data(), r.m); return r; }
inline mat4 rotate_x(float a) { mat4 r; pine_gpu_mat4_rotate_x(a, r.m); return r; }
inline mat4 rotate_y(float a) { mat4 r; pine_gpu_mat4_rotate_y(a, r.m); return r; }
inline mat4 rotate_z(float a) { mat4 r; pine_gpu_mat4_rotate_z(a, r.m); return r; }
inline mat4 inverse(const mat4& a) { mat4 r; pine_gpu_mat4_inverse(a.m, r.m); return r; }
inline mat4 look_at(vec3 from, vec3 to) { mat4 r; pine_gpu_mat4_look_at(from.data(), to.data(), r.m); return r; }

// ---- materials ----
struct Emissive { vec3 color; };
struct Diffuse { vec3 albedo; };
struct Uber { vec3 albedo; float roughness; float metallic = 0.0f; float transmission = 0.0f; float ior = 1.45f; };
struct Subsurface { vec3 albedo; float roughness; vec3 sigma_s; };
// constant-parameter forms of material.h:39-78 (node-graph parameters: use the C ABI's node calls)
struct Metal { vec3 albedo; float roughness; };
struct Glossy { vec3 albedo; float roughness; float ior = 1.4f; };
struct Glass { vec3 albedo; float roughness; float ior = 1.4f; };

// ---- lights other than emissive geometry (light.h:21-67) ----
struct PointLight { vec3 position, color; };
struct SpotLight { vec3 position, direction, color; float falloff_radian, cutoff_additional_radian = 0.0f; };
struct DirectionalLight { vec3 direction, color; };
struct Sky { vec3 sun_color; };

// ---- shapes ----
struct Rect { vec3 position, ex, ey; bool flip_normal = false; };
struct AABB { vec3 lower, upper; };
struct OBB { AABB base; mat4 m; };
inline AABB Box(vec3 lower, vec3 upper) { return {lower, upper}; }
inline OBB Box(AABB aabb, mat4 m) { return {aabb, m}; }
struct Sphere { vec3 center; float radius; };
struct Disk { vec3 position, normal; float radius; };
struct Cone { vec3 position, normal; float radius, height; };
struct Plane { vec3 position, normal; };
struct Line { vec3 p0, p1; float thickness; };
struct Cylinder { vec3 p0, p1; float radius; };
struct Triangle { vec3 v0, v1, v2; };
struct Mesh { std::vector<vec3> vertices; std::vector<std::array<uint32_t, 3>> indices; };

template <class T> struct is_shape : std::false_type {};
template <> struct is_shape<Rect> : std::true_type {};
template <> struct is_shape<AABB> : std::true_type {};
template <> struct is_shape<OBB> : std::true_type {};
template <> struct is_shape<Sphere> : std::true_type {};
template <> struct is_shape<Disk> : std::true_type {};
template <> struct is_shape<Cone> : std::true_type {};
template <> struct is_shape<Plane> : std::true_type {};
template <> struct is_shape<Line> : std::true_type {};
template <> struct is_shape<Cylinder> : std::true_type {};
template <> struct is_shape<Triangle> : std::true_type {};
template <> struct is_shape<Mesh> : std::true_type {};

struct Uncharted2 { static constexpr int code = 0; };
struct ACES { static constexpr int code = 1; };

struct Film {
  vec2i size_;
  int tone_mapper = 0;
  std::vector<float> pixels;  // W*H*4, row 0 first (Array2d<vec4>)
  Film() = default;
  Film(vec2i size, Uncharted2 = {}) : size_(size), tone_mapper(0), pixels(size_t(size.x) * size.y * 4, 0.0f) {}
  Film(vec2i size, ACES) : size_(size), tone_mapper(1), pixels(size_t(size.x) * size.y * 4, 0.0f) {}
  vec2i size() const { return size_; }
  // tone-mapped, gamma 2.2, y-flipped RGBA8 exactly as save() produces it
  std::vector<uint8_t> finalize_u8() const {
    std::vector<uint8_t> out(size_t(size_.x) * size_.y * 4);
    check(pine_gpu_film_finalize_u8(pixels.data(), size_.x, size_.y, tone_mapper, out.data()), "film.finalize");
    return out;
  }
  void save_raw(const std::string& path) const;  // raw float32 dump (PNG encoding lives in the Python layer)
};

struct ThinLenCamera {
  Film film_;
  vec3 from, to;
  float fov, len_radius = 0.0f, focus_distance = 1.0f;
  ThinLenCamera() = default;
  ThinLenCamera(Film film, vec3 from, vec3 to, float fov, float len_radius = 0.0f, float focus_distance = 1.0f)
      : film_(std::move(film)), from(from), to(to), fov(fov), len_radius(len_radius), focus_distance(focus_distance) {}
  Film& film() { return film_; }
};

struct BlueSampler {  // sampler.h:166-201
  int requested;
  explicit BlueSampler(int spp) : requested(spp) {
    if (spp <= 0) throw Error("`BlueSampler` should have positive samples per pixel");
  }
};
struct SobolSampler {  // sampler.h:83-164: spp as given (on the device any count up to 4096)
  int requested;
  explicit SobolSampler(int spp) : requested(spp) {}
};
struct HaltonSampler {  // sampler.h:40-81: spp as given (on the device any count up to 4096)
  int requested;
  explicit HaltonSampler(int spp) : requested(spp) {
    if (spp <= 0) throw Error("`HaltonSampler` should have positive samples per pixel");
  }
};
struct Sampler {  // the variant PathIntegrator takes (sampler.h:275-; UniformSampler is not reproducible)
  int requested, kind;
  Sampler(BlueSampler s) : requested(s.requested), kind(PINE_GPU_SAMPLER_BLUE) {}
  Sampler(SobolSampler s) : requested(s.requested), kind(PINE_GPU_SAMPLER_SOBOL) {}
  Sampler(HaltonSampler s) : requested(s.requested), kind(PINE_GPU_SAMPLER_HALTON) {}
};

class Scene {
 public:
  Scene() : h_(pine_gpu_scene_create()) {}
  ~Scene() { pine_gpu_scene_destroy(h_); }
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;

  int add(const std::string& name, Emissive m) { return check(pine_gpu_scene_add_material_emissive(h_, name.c_str(), m.color.data()), "Emissive"); }
  int add(const std::string& name, Diffuse m) { return check(pine_gpu_scene_add_material_diffuse(h_, name.c_str(), m.albedo.data()), "Diffuse"); }
  int add(const std::string& name, Uber m) {
    return check(pine_gpu_scene_add_material_uber(h_, name.c_str(), m.albedo.data(), m.roughness, m.metallic, m.transmission, m.ior), "Uber");
  }
  int add(const std::string& name, Subsurface m) {
    return check(pine_gpu_scene_add_material_subsurface(h_, name.c_str(), m.albedo.data(), m.roughness, m.sigma_s.data()), "Subsurface");
  }
  int add(const std::string& name, Metal m) {
    return check(pine_gpu_scene_add_material_metal(h_, name.c_str(), node(m.albedo), node(m.roughness)), "Metal");
  }
  int add(const std::string& name, Glossy m) {
    return check(pine_gpu_scene_add_material_glossy(h_, name.c_str(), node(m.albedo), node(m.roughness), node(m.ior)), "Glossy");
  }
  int add(const std::string& name, Glass m) {
    return check(pine_gpu_scene_add_material_glass(h_, name.c_str(), node(m.albedo), node(m.roughness), node(m.ior)), "Glass");
  }
  int add(PointLight l) { return check(pine_gpu_scene_add_light_point(h_, l.position.data(), l.color.data()), "PointLight"); }
  int add(SpotLight l) {
    return check(pine_gpu_scene_add_light_spot(h_, l.position.data(), l.direction.data(), l.color.data(), l.falloff_radian,
                                               l.cutoff_additional_radian), "SpotLight");
  }
  int add(DirectionalLight l) { return check(pine_gpu_scene_add_light_directional(h_, l.direction.data(), l.color.data()), "DirectionalLight"); }
  void set(Sky sky) { check(pine_gpu_scene_set_env_sky(h_, sky.sun_color.data()), "scene.set(Sky)"); }
  // scene.add(shape, "material name") and scene.add(shape, Material)
  template <class S, class = std::enable_if_t<is_shape<S>::value>>
  int add(const S& shape, const std::string& material) {
    return add_shape(shape, check(pine_gpu_scene_find_material(h_, material.c_str()), "scene.add"));
  }
  template <class S, class M, class = std::enable_if_t<is_shape<S>::value && !std::is_convertible<M, std::string>::value>>
  int add(const S& shape, M material) {
    return add_shape(shape, add(std::string(), material));
  }
  ThinLenCamera& set(ThinLenCamera cam) {
    camera = std::move(cam);
    check(pine_gpu_scene_set_camera_thinlens(h_, camera.film_.size_.x, camera.film_.size_.y, camera.film_.tone_mapper,
                                             camera.from.data(), camera.to.data(), camera.fov, camera.len_radius,
                                             camera.focus_distance), "scene.set");
    return camera;
  }
  pine_gpu_scene* handle() const { return h_; }
  ThinLenCamera camera;

 private:
  int add_shape(const Rect& s, int m) { return check(pine_gpu_scene_add_rect(h_, s.position.data(), s.ex.data(), s.ey.data(), s.flip_normal, m), "Rect"); }
  int add_shape(const AABB& s, int m) { return check(pine_gpu_scene_add_aabb(h_, s.lower.data(), s.upper.data(), m), "Box"); }
  int add_shape(const OBB& s, int m) { return check(pine_gpu_scene_add_obb(h_, s.base.lower.data(), s.base.upper.data(), s.m.m, m), "Box"); }
  int add_shape(const Sphere& s, int m) { return check(pine_gpu_scene_add_sphere(h_, s.center.data(), s.radius, m), "Sphere"); }
  int add_shape(const Disk& s, int m) { return check(pine_gpu_scene_add_disk(h_, s.position.data(), s.normal.data(), s.radius, m), "Disk"); }
  int add_shape(const Cone& s, int m) { return check(pine_gpu_scene_add_cone(h_, s.position.data(), s.normal.data(), s.radius, s.height, m), "Cone"); }
  int add_shape(const Plane& s, int m) { return check(pine_gpu_scene_add_plane(h_, s.position.data(), s.normal.data(), m), "Plane"); }
  int add_shape(const Line& s, int m) { return check(pine_gpu_scene_add_line(h_, s.p0.data(), s.p1.data(), s.thickness, m), "Line"); }
  int add_shape(const Cylinder& s, int m) { return check(pine_gpu_scene_add_cylinder(h_, s.p0.data(), s.p1.data(), s.radius, m), "Cylinder"); }
  int add_shape(const Triangle& s, int m) { return check(pine_gpu_scene_add_triangle(h_, s.v0.data(), s.v1.data(), s.v2.data(), m), "Triangle"); }
  int add_shape(const Mesh& s, int m) {
    return check(pine_gpu_scene_add_mesh(h_, &s.vertices[0].x, int(s.vertices.size()), &s.indices[0][0], int(s.indices.size()), m), "Mesh");
  }
  int node(vec3 v) { return check(pine_gpu_scene_node_const3(h_, v.data()), "node"); }
  int node(float v) { return check(pine_gpu_scene_node_constf(h_, v), "node"); }
  pine_gpu_scene* h_;
};

class PathIntegrator {
 public:
  PathIntegrator(Sampler sampler, int max_path_length, int device = 0)
      : sampler_(sampler), max_path_length_(max_path_length), device_(device) {
    if (max_path_length <= 0) throw Error("`PathIntegrator` expect `max_path_length` to be positive");
  }
  // render on several devices of the node from this one process (tiles dealt round-robin, slabs gathered with peer copies)
  PathIntegrator& on_devices(std::vector<int> devices) {
    devices_ = std::move(devices);
    return *this;
  }
  // The scene's own path kernel (same film; pine_gpu.h): by default the library uses it when it is in its cache and compiles it
  // in the background otherwise.  specialize(true): wait for the compiler at render() and fail if the kernel cannot be built
  // (PINE_GPU_FLAG_SPECIALIZE); specialize(false): precompiled kernels only (PINE_GPU_FLAG_NO_SPECIALIZE).
  // closest hits in the order of the reference's default accel, EmbreeAccel (PINE_GPU_FLAG_ORDER_EMBREE), instead of pine-BVH
  // order (the default: Accel(BVH())); only scenes with order-dependent shapes -- a scaled Box(AABB, mat4), Plane, Line,
  // Cylinder -- show the difference
  PathIntegrator& order_embree(bool on = true) {
    flags_ = on ? (flags_ | PINE_GPU_FLAG_ORDER_EMBREE) : (flags_ & ~PINE_GPU_FLAG_ORDER_EMBREE);
    return *this;
  }
  PathIntegrator& order_nearest(bool on = true) { return order_embree(on); }  // (former name)
  PathIntegrator& specialize(bool on = true) {
    flags_ &= ~(PINE_GPU_FLAG_SPECIALIZE | PINE_GPU_FLAG_NO_SPECIALIZE);
    flags_ |= on ? PINE_GPU_FLAG_SPECIALIZE : PINE_GPU_FLAG_NO_SPECIALIZE;
    return *this;
  }
  void render(Scene& scene) {
    pine_gpu_render_params p{};
    p.flags = flags_;
    p.spp = sampler_.requested;
    p.max_path_length = max_path_length_;
    p.device = device_;
    p.shard_rank = 0;
    p.shard_world = 1;
    p.sampler = sampler_.kind;
    if (devices_.size() > 1)
      check(pine_gpu_path_render_devices(scene.handle(), &p, devices_.data(), int(devices_.size()), scene.camera.film_.pixels.data()),
            "PathIntegrator::render");
    else
      check(pine_gpu_path_render(scene.handle(), &p, scene.camera.film_.pixels.data()), "PathIntegrator::render");
  }

 private:
  Sampler sampler_;
  int max_path_length_, device_;
  int flags_ = 0;
  std::vector<int> devices_;
};

inline float get_progress() { return pine_gpu_progress(); }

// load(scene, "file.glb" [, mat4]) (fileio.cpp:584-589): glTF import -- every mesh primitive with its material, and the
// camera when a node carries one (gltf_import.hpp)
inline void load(Scene& scene, const std::string& filename, const mat4& m = mat4()) {
  pine_gltf::ImportedCamera cam;
  try {
    cam = pine_gltf::import_scene(scene.handle(), filename, m.m);
  } catch (const pine_gltf::Error& e) {
    throw Error(std::string("load: ") + e.what());
  }
  if (cam.present)
    scene.set(ThinLenCamera(Film(vec2i{cam.film_w, cam.film_h}), vec3(cam.from[0], cam.from[1], cam.from[2]),
                            vec3(cam.to[0], cam.to[1], cam.to[2]), cam.fov));
}

}  // namespace pine
