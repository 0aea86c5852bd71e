// examples/adapter/path_gpu.h -- the binding a pine maintainer adds to route `PathIntegrator(sampler, n).render(scene)`
// through libpine_gpu.so.  This file is compiled against the REAL reference headers
// (tests/test_adapter.py: g++ -std=c++20 -I/root/reference/src, linked with the reference objects of oracle/_ref and
// with libpine_gpu.so), so what it reads from a constructed pine::Scene is what is really there.
//
// pine's shape classes keep their state private (src/pine/core/geometry.h:23-25, :56-60, :92-96, :115-117, :134-140).
// In pine's tree the binding is a friend (one `friend struct GpuPathIntegrator;` line per class); built outside the
// tree, as here, the including file opens the access specifiers BEFORE including pine's headers
// (`#define private public`, see roundtrip.cpp) -- the object layout is the same either way.
//
// Shapes whose members are their constructor arguments (Sphere, AABB, OBB, Line, Cylinder, Mesh) go through the
// constructor-level calls; Rect / Disk / Plane / Cone / Triangle and the camera through the *_state calls, member for
// member: a stored normalised axis does not survive being normalised again bit for bit.
#pragma once
#include <pine/core/scene.h>
#include <pine/core/sampler.h>
#include <pine/core/material.h>
#include <pine/core/accel.h>
#include <pine/core/lightsampler.h>

#include <pine_gpu.h>

namespace pine {

struct GpuPathIntegrator {
  // PathIntegrator(Accel, Sampler, LightSampler, int) (program_context.cpp:76-78).  The accel decides the ORDER closest hits are
  // found in, which a few shapes can see (the transformed Box, bbox.cpp:149-171; Plane, Line, Cylinder): an Accel that holds an
  // EmbreeAccel -- the only one a .pine script can construct, and what the two-argument constructor of
  // program_context.cpp:79-81 passes -- renders in EmbreeAccel's order (PINE_GPU_FLAG_ORDER_EMBREE: the films of the reference's
  // EmbreeAccel, bit for bit); a BVH, in pine-BVH order.  The light sampler has one alternative, UniformLightSampler: nothing to choose.
  GpuPathIntegrator(const Accel& accel, Sampler sampler_, LightSampler, int max_path_length)
      : GpuPathIntegrator(MOVE(sampler_), max_path_length) {
    order_embree = accel.is<EmbreeAccel>();
  }
  GpuPathIntegrator(Sampler sampler_, int max_path_length) : sampler(MOVE(sampler_)), max_path_length(max_path_length) {
    if (max_path_length <= 0)
      SEVERE("`PathIntegrator` expect `max_path_length` to be positive, get", max_path_length);  // path.cpp:12-13
    if (!sampler.is<BlueSobolSampler>() && !sampler.is<SobolSampler>() && !sampler.is<HaltonSampler>())
      SEVERE("GpuPathIntegrator supports BlueSampler, SobolSampler and HaltonSampler");
  }

  // Replays an already built scene on the C ABI.  The caller owns the returned handle.
  static pine_gpu_scene* mirror(Scene& scene) {
    auto* s = pine_gpu_scene_create();
    psl::map<const Material*, int> ids;
    int counter = 0;
    auto material_id = [&](const psl::shared_ptr<Material>& m) -> int {
      if (auto it = ids.find(m.get()); it != ids.end()) return it->second;
      auto name = psl::string("m") + psl::to_string(counter++);
      // constant shading nodes: evaluated once at a dummy context (node graphs are replayed with pine_gpu_scene_node_*)
      auto c = NodeEvalCtx(vec3(0), vec3(0, 0, 1), vec2(0));
      int id = m->dispatch([&]<typename T>(const T& x) -> int {
        if constexpr (psl::same_as<T, EmissiveMaterial>) {
          vec3 v = x.color.eval(c);
          return pine_gpu_scene_add_material_emissive(s, name.c_str(), &v[0]);
        } else if constexpr (psl::same_as<T, DiffuseMaterial>) {
          vec3 v = x.albedo.eval(c);
          return pine_gpu_scene_add_material_diffuse(s, name.c_str(), &v[0]);
        } else if constexpr (psl::same_as<T, UberMaterial>) {
          vec3 v = x.albedo.eval(c);
          return pine_gpu_scene_add_material_uber(s, name.c_str(), &v[0], x.roughness.eval(c), x.metallic.eval(c), x.transmission.eval(c), x.ior);
        } else if constexpr (psl::same_as<T, SubsurfaceMaterial>) {
          vec3 v = x.albedo.eval(c);
          return pine_gpu_scene_add_material_subsurface(s, name.c_str(), &v[0], x.roughness.eval(c), &x.sigma_s[0]);
        } else {
          SEVERE("material not mirrored by this example (Metal / Glossy / Glass take node ids: pine_gpu_scene_add_material_metal ...)");
          return -1;
        }
      });
      return ids[m.get()] = id;
    };
    for (auto& g : scene.geometries) {  // geometry order = BVH primitive order: keep it
      const int m = material_id(g->material);
      g->shape.dispatch([&]<typename T>(const T& x) {
        if constexpr (psl::same_as<T, Rect>)
          pine_gpu_scene_add_rect_state(s, &x.position[0], &x.ex[0], &x.ey[0], &x.n[0], x.lx, x.ly, &x.rx[0], &x.ry[0], m);
        else if constexpr (psl::same_as<T, Disk>)
          pine_gpu_scene_add_disk_state(s, &x.position[0], &x.n[0], &x.u[0], &x.v[0], x.r, m);
        else if constexpr (psl::same_as<T, Plane>)
          pine_gpu_scene_add_plane_state(s, &x.position[0], &x.n[0], &x.u[0], &x.v[0], m);
        else if constexpr (psl::same_as<T, Cone>)
          pine_gpu_scene_add_cone_state(s, &x.p[0], &x.n[0], x.r, x.h, x.A, x.A2, x.S, &x.bottom.position[0], m);
        else if constexpr (psl::same_as<T, Triangle>)
          pine_gpu_scene_add_triangle_state(s, &x.v0[0], &x.v1[0], &x.v2[0], &x.n[0], m);
        else if constexpr (psl::same_as<T, Sphere>)
          pine_gpu_scene_add_sphere(s, &x.c[0], x.r, m);
        else if constexpr (psl::same_as<T, AABB>)
          pine_gpu_scene_add_aabb(s, &x.lower[0], &x.upper[0], m);
        else if constexpr (psl::same_as<T, OBB>)
          pine_gpu_scene_add_obb(s, &x.base.lower[0], &x.base.upper[0], &x.m[0][0], m);  // mat4: column vectors, m[c*4 + r]
        else if constexpr (psl::same_as<T, Line>)
          pine_gpu_scene_add_line(s, &x.p0[0], &x.p1[0], x.thickness, m);
        else if constexpr (psl::same_as<T, Cylinder>)
          pine_gpu_scene_add_cylinder(s, &x.p0[0], &x.p1[0], x.r, m);
        else if constexpr (psl::same_as<T, Mesh>)
          pine_gpu_scene_add_mesh(s, &x.vertices[0][0], int(x.vertices.size()), &x.indices[0][0], int(x.indices.size()), m);
        else
          SEVERE("shape not supported by the GPU path (SDF / CSG shapes carry script functions)");
      });
    }
    auto& cam = scene.camera.as<ThinLenCamera>();
    auto& film = cam.film();
    pine_gpu_scene_set_camera_thinlens_state(s, film.width(), film.height(), /*tonemapper*/ 0, &cam.position[0], &cam.c2w[0][0],
                                             &cam.fov2d[0], cam.len_radius, cam.focus_distance);
    return s;
  }

  void render(Scene& scene) {
    auto* s = mirror(scene);
    auto& film = scene.camera.film();
    pine_gpu_render_params prm{sampler.spp(), max_path_length, /*device*/ 0, /*rank*/ 0, /*world*/ 1, 0,
                               (specialize ? PINE_GPU_FLAG_SPECIALIZE : 0) | (order_embree ? PINE_GPU_FLAG_ORDER_EMBREE : 0),
                               sampler.is<SobolSampler>() ? PINE_GPU_SAMPLER_SOBOL : sampler.is<HaltonSampler>() ? PINE_GPU_SAMPLER_HALTON : PINE_GPU_SAMPLER_BLUE};
    // film.data() is Array2d<vec4>: W*H float4, row 0 first -- exactly the layout the ABI writes (array.h:51-55)
    const int rc = pine_gpu_path_render(s, &prm, &film.data()[0][0]);
    const psl::string err = rc < 0 ? pine_gpu_last_error() : "";
    pine_gpu_scene_destroy(s);
    if (rc < 0) SEVERE(err);  // the reference's error path: log + abort (log.h:45-51)
  }

  Sampler sampler;
  int max_path_length;
  // the scene's own path kernel (its exact feature set; small scenes baked in): same film, cbox 24 % faster.  By default the
  // library loads it from its on-disk cache, or compiles it in the background while the precompiled kernel renders this call;
  // true: wait for the compiler here (seconds of hipcc the first time a geometry is seen) and fail if it cannot be built
  bool specialize = false;
  // closest hits in EmbreeAccel's order (set by the four-argument constructor from the Accel it is handed)
  bool order_embree = false;
};

}  // namespace pine
