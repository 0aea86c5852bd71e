"""Quick on-GPU check used during bring-up: render small cbox variants and diff against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pine_amd
from pine_amd import scenes, _lib
from oracle import oracle
import ctypes as C

def cmp(name, scene, spp, depth):
    w, h = scene.camera.film().size
    t0 = time.time()
    film = pine_amd.PathIntegrator(pine_amd.BlueSampler(spp), depth).render(scene).pixels
    t1 = time.time()
    ref, st = oracle.render(scene.describe(), (w, h), spp, depth)
    bad = (ref.view(np.uint32) != film.view(np.uint32)).any(axis=2)
    d = np.abs(ref[..., :3] - film[..., :3])
    print(f"{name}: {w}x{h} spp{spp} d{depth} gpu_oneshot {t1-t0:.3f}s mismatched_px={int(bad.sum())}/{w*h} "
          f"max_abs={d.max():.3e} mean ref={ref[...,:3].mean():.6f} gpu={film[...,:3].mean():.6f}", flush=True)
    if bad.any():
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:5]:
            print("   px", x, y, ref[y, x, :3], film[y, x, :3])
    return int(bad.sum())

# building blocks first
x = np.linspace(-7, 7, 200001, dtype=np.float32)
s = np.zeros_like(x); c = np.zeros_like(x)
_lib.check(_lib.lib.pine_gpu_test_sincos(0, x.ctypes.data_as(_lib.c_f_p), x.size, s.ctypes.data_as(_lib.c_f_p), c.ctypes.data_as(_lib.c_f_p)))
print("sincos mismatches vs host libm:", int((s.view(np.uint32) != np.sin(x, dtype=np.float32).view(np.uint32)).sum()),
      int((c.view(np.uint32) != np.cos(x, dtype=np.float32).view(np.uint32)).sum()), "(numpy may not be glibc)")
for spp in (1, 16, 256):
    ref = oracle.sampler_stream(spp)
    out = np.zeros_like(ref)
    _lib.check(_lib.lib.pine_gpu_test_sampler(0, spp, out.ctypes.data_as(_lib.c_f_p), out.size))
    print("sampler", spp, "bit-equal:", np.array_equal(ref.view(np.uint32), out.view(np.uint32)))
r = oracle.rng_stream(); o = np.zeros_like(r)
_lib.check(_lib.lib.pine_gpu_test_rng(0, o.ctypes.data_as(C.POINTER(C.c_uint64)), o.size))
print("rng bit-equal:", np.array_equal(r, o))

tot = 0
tot += cmp("cbox", scenes.cbox((64, 64)), 16, 4)
tot += cmp("cbox readme", scenes.cbox((64, 64), "readme"), 16, 4)
tot += cmp("cbox rect", scenes.cbox((64, 64), "readme", False), 64, 8)
tot += cmp("cbox 128 256spp", scenes.cbox((128, 128)), 256, 8)
tot += cmp("zoo", scenes.shapes_zoo((96, 96)), 16, 5)
tot += cmp("classic20", scenes.classic_cones((180, 90), 20), 64, 6)
tot += cmp("sss", scenes.sss((96, 96), 2), 64, 8)
print("TOTAL mismatched pixels:", tot)
