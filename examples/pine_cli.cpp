// examples/pine_cli.cpp -- `pine-mi355x script.pine`: what `pine script.pine` is in the reference
// (src/cli/pine.cpp:16-46: interpret the file on a worker, poll get_progress() on the main thread),
// with the PathIntegrator path running on the MI355X.  --dry-run prints the scene description a
// render would receive instead of touching the GPU.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <future>
#include <sstream>
#include <string>

#include "../include/pine_gpu.h"
#include "../include/pine_prl.h"

int main(int argc, char** argv) {
  int flags = PINE_PRL_ECHO, device = 0;
  const char* path = nullptr;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--dry-run")) flags |= PINE_PRL_DRY_RUN;
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--devices") && i + 1 < argc) setenv("PINE_GPU_DEVICES", argv[++i], 1);  // e.g. 0,1,2,3,4,5,6,7
    else if (!strcmp(argv[i], "--tables") && i + 1 < argc) pine_gpu_set_table_path(argv[++i]);
    // which accel the two-argument PathIntegrator(sampler, n) stands for: embree (what real pine's default EmbreeAccel renders: the
    // default for scenes without meshes) or bvh (pine-BVH order, the bits of Accel(BVH())); only order-dependent shapes -- a
    // transformed Box, Plane, Line, Cylinder -- can tell
    else if (!strcmp(argv[i], "--accel") && i + 1 < argc) setenv("PINE_PRL_ACCEL", argv[++i], 1);
    else path = argv[i];
  }
  if (!path) {
    fprintf(stderr, "Usage: pine-mi355x [--dry-run] [--device N | --devices 0,1,...] [--tables bluesobol_u8.bin] [--accel bvh|embree] [filename]\n");
    return 2;
  }
  std::ifstream f(path);
  if (!f) {
    fprintf(stderr, "Unable to open file `%s`\n", path);
    return 1;
  }
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string source = ss.str();
  // (pine_prl_last_error() is per thread: the worker reads it)
  std::string error;
  auto worker = std::async(std::launch::async, [&] {
    const int rc = pine_prl_interpret(source.c_str(), flags, device);
    if (rc < 0) error = pine_prl_last_error();
    return rc;
  });
  while (worker.wait_for(std::chrono::milliseconds(200)) != std::future_status::ready) {
    const float p = pine_gpu_progress();
    if (p > 0 && p < 1) fprintf(stderr, "\r[progress] %5.1f%%", 100.0 * p);
  }
  if (worker.get() < 0) {
    fprintf(stderr, "%s: %s\n", path, error.c_str());
    return 1;
  }
  return 0;
}
