// examples/cbox.cpp -- scenes/cbox.pine written against the C++ facade (pine_amd/host/pine.hpp).
//   hipcc/g++ -std=c++17 examples/cbox.cpp -Lpine_amd/lib -lpine_gpu -Wl,-rpath,$PWD/pine_amd/lib -o cbox
//   ./cbox pine_amd/data/bluesobol_u8.bin 640 256 8 out.film
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../pine_amd/host/pine.hpp"

using namespace pine;

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s <bluesobol_u8.bin> <size> <spp> <depth> <out.film>\n", argv[0]);
    return 2;
  }
  try {
    check(pine_gpu_set_table_path(argv[1]), "tables");
    const int size = atoi(argv[2]), spp = atoi(argv[3]), depth = atoi(argv[4]);
    Scene scene;
    scene.add("floor", Diffuse{{0.9f, 0.9f, 0.9f}});
    scene.add("blue", Diffuse{{0.2f, 0.5f, 0.9f}});
    scene.add("red", Diffuse{{0.9f, 0.1f, 0.05f}});
    scene.add("green", Diffuse{{0.2f, 0.9f, 0.05f}});
    scene.add(Rect{{0, 0, 1}, {2, 0, 0}, {0, 0, 2}, true}, "floor");
    scene.add(Rect{{0, 2, 1}, {2, 0, 0}, {0, 0, 2}}, "floor");
    scene.add(Rect{{-1, 1, 1}, {0, 0, 2}, {0, 2, 0}, true}, "red");
    scene.add(Rect{{1, 1, 1}, {0, 0, 2}, {0, 2, 0}}, "green");
    scene.add(Rect{{0, 1, 2}, {2, 0, 0}, {0, 2, 0}, true}, "blue");
    scene.add(Box(AABB{{0, 0, 0}, {1, 1, 1}}, translate({0.0f, 0.0f, 0.6f}) * rotate_y(0.4f) * scale({0.6f, 0.6f, 0.6f})), "floor");
    scene.add(Box(AABB{{0, 0, 0}, {1, 1, 1}}, translate({-0.6f, 0.0f, 1.0f}) * rotate_y(-0.4f) * scale({0.6f, 1.3f, 0.6f})), "floor");
    scene.add(Rect{{0.0f, 1.9f, 1}, {0.1f, 0, 0}, {0, 0, 0.1f}}, Emissive{600.0f * vec3{1.0f, 0.64f, 0.185f}});
    scene.set(ThinLenCamera(Film({size, size}, Uncharted2()), {0, 0, 0}, {0, 0, 1}, 0.4f));

    auto t0 = std::chrono::steady_clock::now();
    // (the library uses the scene's own kernel when its cache has it and compiles it in the background otherwise;
    //  CBOX_SPECIALIZE=1 waits for the compiler here, =0 keeps to the precompiled kernels -- same film either way)
    PathIntegrator integrator(BlueSampler(spp), depth);
    if (const char* e = getenv("CBOX_SPECIALIZE")) integrator.specialize(atoi(e) != 0);
    integrator.render(scene);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto& film = scene.camera.film();
    FILE* f = fopen(argv[5], "wb");
    fwrite(film.pixels.data(), 4, film.pixels.size(), f);
    fclose(f);
    printf("rendered %dx%d spp %d depth %d in %.3f s (one-shot, incl. upload/download)\n", size, size, spp, depth, s);
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
