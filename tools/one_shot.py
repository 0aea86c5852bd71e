"""What the drop-in call costs end to end: PathIntegrator(BlueSampler(256), 8).render(scene) -- plan creation (BVH, device
buffers, tables, the scene's kernel from the cache), launch, film to the host, plan destruction -- called several times in a
row on BASELINE's C2; PINE_GPU_POOL_MB=0 shows the same without the device-memory pool.
usage (GPU box): python3 tools/one_shot.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pine_amd as pa
from pine_amd import scenes
sc = scenes.cbox((640, 640), "committed")
integ = pa.PathIntegrator(pa.BlueSampler(256), 8)
times = []
for i in range(8):
    t0 = time.perf_counter()
    integ.render(sc)
    times.append((time.perf_counter() - t0) * 1e3)
print("pool", os.environ.get("PINE_GPU_POOL_MB", "default"), "one-shot render ms:", " ".join(f"{t:.1f}" for t in times),
      f"-> {640 * 640 * 256 / (min(times[3:]) * 1e-3) * 1e-6:.0f} Msamples/s at best")
