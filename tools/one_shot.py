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
def run(n):
    times = []
    for i in range(n):
        t0 = time.perf_counter()
        integ.render(sc)
        times.append((time.perf_counter() - t0) * 1e3)
    return times
first = run(4)  # (an empty kernel cache: these run the precompiled kernel while the scene's kernel compiles in the background)
time.sleep(float(os.environ.get("ONE_SHOT_WAIT_S", "4")))
later = run(6)  # (the scene's kernel comes from the cache at plan creation)
print("pool", os.environ.get("PINE_GPU_POOL_MB", "default"), "| first calls ms:", " ".join(f"{t:.1f}" for t in first), "| after the background build:",
      " ".join(f"{t:.1f}" for t in later), f"-> {640 * 640 * 256 / (min(later) * 1e-3) * 1e-6:.0f} Msamples/s per one-shot call at best")
