"""GPU: scene-specialised vs precompiled path kernel on seeded random scenes (and the cbox family): bit equality of the
films, path-kernel time of each, time of the specialisation step.  usage: python tools/spec_scenes.py [first_seed [count]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pine_amd as pa
from pine_amd import scenes, _lib

first = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
stream = torch.cuda.current_stream().cuda_stream


def run(sc, spp, depth, sampler, spec):
    w, h = sc.camera.film().size
    t0 = time.perf_counter()
    plan = pa.Plan(sc, spp, depth, sampler=sampler, timing=True, specialize=spec)
    create = (time.perf_counter() - t0) * 1e3
    film = torch.zeros((h, w, 4), device="cuda")
    for _ in range(4):
        plan.launch(film.data_ptr(), stream)
    torch.cuda.synchronize()
    plan.check()
    st = plan.stats()
    out = film.cpu().numpy()
    plan.close()
    return out, st, create


cases = [("cbox", scenes.cbox((256, 256), "readme"), 64, 8, "blue"), ("cbox rects", scenes.cbox((256, 256), "readme", False), 64, 8, "blue"),
         ("mats_zoo", scenes.materials_zoo((128, 128)), 32, 6, "blue"), ("lights_zoo", scenes.lights_zoo((128, 128)), 32, 6, "blue"),
         ("xshapes", scenes.xshapes_zoo((128, 128)), 32, 5, "blue"), ("cones12", scenes.classic_cones((180, 90), 12), 32, 6, "blue")]
cases += [("C4 cones", scenes.classic_cones((720, 360), 100), 64, 6, "blue"), ("sss mesh", scenes.sss((256, 256), 2), 64, 8, "blue"),
          ("zoo", scenes.shapes_zoo((256, 256)), 64, 5, "blue"), ("cones12 checker", scenes.classic_cones((180, 90), 12, checker_floor=True), 32, 6, "blue")]
for seed in range(first, first + count):
    sc, spp, depth, sampler = scenes.random_scene(seed, variety=True if seed < 4000 else 2)
    cases.append((f"random {seed}", sc, max(spp, 64) if sampler == "blue" else spp, depth, sampler))
for name, sc, spp, depth, sampler in cases:
    a, st0, c0 = run(sc, spp, depth, sampler, False)
    b, st1, c1 = run(sc, spp, depth, sampler, True)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print(f"{name:14s} specialised {st1.specialized} equal {same}  path kernel {st0.trace_ms:8.3f} -> {st1.trace_ms:8.3f} ms  "
          f"features {st0.kernel_features:#x} -> {st1.kernel_features:#x}  plan creation {c0:7.1f} -> {c1:8.1f} ms (specialise {st1.specialize_ms:8.1f})", flush=True)
