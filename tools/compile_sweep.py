"""CPU (hipcc cross-compiles): does path_queue_kernel instantiate for feature sets no precompiled variant has?  Level 1 of
PINE_GPU_FLAG_SPECIALIZE compiles `need | layout` for whatever a scene contains: every combination must at least compile.
usage: python tools/compile_sweep.py [N random combinations per layout, default 6] [seed]"""
import sys, os, time, ctypes as C, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pine_amd import _lib

F = dict(AABB=1, OBB=2, SPHERE=4, DISK=8, CONE=16, MESH=32, UBER=64, SSS=128, LDS_SCENE=256, NODES=512, LIGHTS=1024, XSHAPES=2048, SOBOL=4096,
         LDS_TOP=8192, LDS_REST=16384, XSTAGE=32768)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
content = ["AABB", "OBB", "SPHERE", "DISK", "CONE", "UBER", "NODES", "LIGHTS", "XSHAPES", "SOBOL"]
layouts = [("scene in LDS", F["LDS_SCENE"], 1536, False, False), ("BVH top in LDS", F["LDS_TOP"], 1024, True, True),
           ("top + records in LDS", F["LDS_TOP"] | F["LDS_REST"], 1024, True, True), ("top + records, traversal stages", F["LDS_TOP"] | F["LDS_REST"] | F["XSTAGE"], 1024, True, True),
           ("global", 0, 1024, True, True)]
out = C.create_string_buffer(1024)
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    os.environ["PINE_GPU_CACHE_DIR"] = tmp
    for name, bits, ctx, mesh_ok, sss_ok in layouts:
        combos = [0, sum(F[c] for c in content)]                      # nothing but Rects + Diffuse; everything
        combos += [F[c] for c in content]                                # each alone
        for _ in range(n):
            combos.append(sum(F[c] for c in content if rng.random() < 0.4))
        for extra in ([0] + ([F["MESH"]] if mesh_ok else []) + ([F["SSS"], F["MESH"] | F["SSS"]] if sss_ok else [])):
            for c in combos if extra == 0 else combos[:2] + combos[-n:]:
                f = c | extra | bits
                if (f & F["XSTAGE"]) and not (f & F["MESH"]):
                    continue  # (traversal stages are chosen for mesh scenes only)
                if (f & F["SOBOL"]) and (f & F["SSS"]):
                    continue  # (rejected at plan creation)
                t = time.time()
                r = _lib.lib.pine_gpu_test_specialize_compile(None, f, ctx, b"gfx950", out, 1024)
                ok = r >= 0
                bad += not ok
                print(f"{name:34s} features {f:#07x} ctx {ctx}: {'ok' if ok else 'FAILED'} {time.time() - t:5.1f} s" + ("" if ok else "\n" + _lib.last_error()[-1500:]), flush=True)
print("failed:", bad)
sys.exit(1 if bad else 0)
