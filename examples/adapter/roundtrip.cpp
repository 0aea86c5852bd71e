// examples/adapter/roundtrip.cpp -- builds scenes with the REFERENCE's own API, mirrors them through the adapter
// (path_gpu.h) and prints the scene the C ABI received (pine_gpu_scene_describe).  tests/test_adapter.py compiles this
// against /root/reference/src, links it with oracle/_ref/libpine_ref.a and libpine_gpu.so (host code only: no GPU is
// touched), and checks that the oracle renders the mirrored scene to the same film, bit for bit, as the scene the
// product builds from the same constructor arguments.
//
//   roundtrip cbox|zoo          print the mirrored scene
//   roundtrip cbox|zoo records  print the device records (one line of 32 hex words per geometry, then the camera's 20)
//   roundtrip cbox|zoo render <W> <H> <spp> <depth> <out.film>   GpuPathIntegrator(BlueSampler(spp), depth).render(scene) on
//                               GPU 0 -- the call that replaces program_context.cpp:76-81 -- and the scene's own film
//                               (Array2d<vec4>, row 0 first) written raw: tests/test_adapter.py (-m gpu) compares it with the
//                               film the real reference rendered of the same scene
#define private public  // out-of-tree stand-in for `friend struct GpuPathIntegrator;` (see path_gpu.h)
#include <pine/core/scene.h>
#undef private
#include "path_gpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace pine;

static Scene cbox(vec2i size) {  // scenes/cbox.pine:4-21
  Scene scene;
  scene.add_material("floor", DiffuseMaterial(vec3(0.9f, 0.9f, 0.9f)));
  scene.add_material("blue", DiffuseMaterial(vec3(0.2f, 0.5f, 0.9f)));
  scene.add_material("red", DiffuseMaterial(vec3(0.9f, 0.1f, 0.05f)));
  scene.add_material("green", DiffuseMaterial(vec3(0.2f, 0.9f, 0.05f)));
  scene.add_geometry(Rect(vec3(0, 0, 1), vec3(2, 0, 0), vec3(0, 0, 2), true), "floor");
  scene.add_geometry(Rect(vec3(0, 2, 1), vec3(2, 0, 0), vec3(0, 0, 2)), "floor");
  scene.add_geometry(Rect(vec3(-1, 1, 1), vec3(0, 0, 2), vec3(0, 2, 0), true), "red");
  scene.add_geometry(Rect(vec3(1, 1, 1), vec3(0, 0, 2), vec3(0, 2, 0)), "green");
  scene.add_geometry(Rect(vec3(0, 1, 2), vec3(2, 0, 0), vec3(0, 2, 0), true), "blue");
  scene.add_geometry(OBB(AABB(vec3(0, 0, 0), vec3(1, 1, 1)), translate(vec3(0.0f, 0.0f, 0.6f)) * rotate_y(0.4f) * scale(vec3(0.6f, 0.6f, 0.6f))), "floor");
  scene.add_geometry(OBB(AABB(vec3(0, 0, 0), vec3(1, 1, 1)), translate(vec3(-0.6f, 0.0f, 1.0f)) * rotate_y(-0.4f) * scale(vec3(0.6f, 1.3f, 0.6f))), "floor");
  scene.add_geometry(Rect(vec3(0.0f, 1.9f, 1), vec3(0.1f, 0, 0), vec3(0, 0, 0.1f)), EmissiveMaterial(600.0f * vec3(1.0f, 0.64f, 0.185f)));
  scene.set_camera(ThinLenCamera(Film(size), vec3(0, 1, -4), vec3(0, 1, 0), 0.25f));
  return scene;
}

static Scene zoo(vec2i size) {  // one of every analytic shape kind, Uber and Subsurface materials, a thin lens
  Scene scene;
  scene.add_material("d", DiffuseMaterial(vec3(0.8f, 0.7f, 0.6f)));
  scene.add_material("u", UberMaterial(vec3(0.9f, 0.6f, 0.3f), 0.3f, 0.0f, 0.0f));
  scene.add_material("s", SubsurfaceMaterial(vec3(0.9f, 0.8f, 0.7f), 0.2f, vec3(20.0f, 30.0f, 40.0f)));
  scene.add_geometry(Plane(vec3(0, 0, 0), vec3(0.05f, 1, -0.02f)), "d");
  scene.add_geometry(Rect(vec3(0, 1, 2.2f), vec3(3, 0.1f, 0), vec3(0, 2, 0.3f), true), "d");
  scene.add_geometry(AABB(vec3(-0.9f, 0.0f, 0.2f), vec3(-0.5f, 0.5f, 0.6f)), "u");
  scene.add_geometry(Sphere(vec3(0.5f, 0.3f, 1.2f), 0.3f), "u");
  scene.add_geometry(Disk(vec3(0.0f, 1.5f, 1.0f), vec3(0.2f, -1.0f, 0.1f), 0.4f), "d");
  scene.add_geometry(Cone(vec3(-0.3f, 0.0f, 1.4f), vec3(0.1f, 1, 0.05f), 0.2f, 0.5f), "d");
  scene.add_geometry(Cylinder(vec3(-0.6f, 0.25f, 1.2f), vec3(-0.6f, 1.25f, 1.2f), 0.25f), "u");
  scene.add_geometry(Line(vec3(0.1f, 0.1f, 0.9f), vec3(0.8f, 0.9f, 1.5f), 0.06f), "u");
  scene.add_geometry(Triangle(vec3(-0.3f, 0.0f, 1.8f), vec3(0.5f, 0.0f, 1.9f), vec3(0.1f, 1.1f, 1.7f)), "d");
  {
    psl::vector<vec3> v;
    v.push_back(vec3(0.2f, 0.1f, 0.5f)), v.push_back(vec3(0.7f, 0.1f, 0.5f)), v.push_back(vec3(0.7f, 0.6f, 0.6f)), v.push_back(vec3(0.2f, 0.6f, 0.6f));
    v.push_back(vec3(0.45f, 0.35f, 0.2f));
    psl::vector<vec3u32> f;
    f.push_back(vec3u32(0, 1, 2)), f.push_back(vec3u32(0, 2, 3)), f.push_back(vec3u32(0, 4, 1)), f.push_back(vec3u32(1, 4, 2)),
        f.push_back(vec3u32(2, 4, 3)), f.push_back(vec3u32(3, 4, 0));
    scene.add_geometry(Mesh(v, f), "s");
  }
  scene.add_geometry(Rect(vec3(0.0f, 1.9f, 1), vec3(0.5f, 0, 0), vec3(0, 0, 0.5f)), EmissiveMaterial(vec3(20.0f, 18.0f, 15.0f)));
  scene.set_camera(ThinLenCamera(Film(size), vec3(0.1f, 1, -4), vec3(0, 1, 0), 0.25f, 0.03f, 4.5f));
  return scene;
}

int main(int argc, char** argv) {
  const std::string which = argc > 1 ? argv[1] : "cbox";
  if (argc == 8 && std::string(argv[2]) == "render") {
    const vec2i size(atoi(argv[3]), atoi(argv[4]));
    Scene scene = which == "zoo" ? zoo(size) : cbox(size);
    scene.camera.film().clear();
    // ROUNDTRIP_ACCEL=embree | bvh: the four-argument constructor with that Accel (never built: the GPU path needs none)
    const char* accel_name = getenv("ROUNDTRIP_ACCEL");
    GpuPathIntegrator integ = accel_name
        ? GpuPathIntegrator(std::string(accel_name) == "embree" ? Accel(EmbreeAccel()) : Accel(BVH()), Sampler(BlueSobolSampler(atoi(argv[5]))),
                            LightSampler(UniformLightSampler()), atoi(argv[6]))
        : GpuPathIntegrator(Sampler(BlueSobolSampler(atoi(argv[5]))), atoi(argv[6]));
    integ.specialize = getenv("ROUNDTRIP_SPECIALIZE") != nullptr;  // (the test renders both ways)
    integ.render(scene);  // (aborts through SEVERE on any error, as the reference's integrators do)
    auto& film = scene.camera.film();
    FILE* f = fopen(argv[7], "wb");
    if (!f || fwrite(film.data(), 16, size_t(size.x) * size.y, f) != size_t(size.x) * size.y) return 2;
    fclose(f);
    printf("rendered %dx%d\n", size.x, size.y);
    return 0;
  }
  Scene scene = which == "zoo" ? zoo(vec2i(40, 32)) : cbox(vec2i(48, 48));
  pine_gpu_scene* s = GpuPathIntegrator::mirror(scene);
  if (argc > 2 && std::string(argv[2]) == "records") {
    for (int g = 0; g < int(scene.geometries.size()); g++) {
      float rec[32];
      if (pine_gpu_scene_shape_record(s, g, rec) < 0) return 1;
      for (int i = 0; i < 32; i++) {
        unsigned u;
        memcpy(&u, &rec[i], 4);
        printf(i ? " %08x" : "%08x", u);
      }
      printf("\n");
    }
    float cam[20];
    pine_gpu_scene_camera_record(s, cam);
    for (int i = 0; i < 20; i++) {
      unsigned u;
      memcpy(&u, &cam[i], 4);
      printf(i ? " %08x" : "%08x", u);
    }
    printf("\n");
    pine_gpu_scene_destroy(s);
    return 0;
  }
  const long n = long(pine_gpu_scene_describe(s, nullptr, 0));
  std::vector<char> buf(size_t(n) + 1);
  pine_gpu_scene_describe(s, buf.data(), n + 1);
  fwrite(buf.data(), 1, size_t(n), stdout);
  pine_gpu_scene_destroy(s);
  // the integrator itself must at least construct and type-check against the real Sampler variant
  GpuPathIntegrator integ(Sampler(BlueSobolSampler(16)), 4);
  (void)integ;
  return 0;
}
