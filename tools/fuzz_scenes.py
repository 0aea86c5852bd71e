"""Randomised scene fuzzing: seeded random mixes of every shape, material and light kind.
   python tools/fuzz_scenes.py ref N   -- here (CPU): the oracle restatement against the real reference binary
   python tools/fuzz_scenes.py gpu N   -- on the GPU box: the HIP path against the oracle
   optional: VARIETY (1 | 2), the first SEED, and ORDER ("pine" | "embree": EmbreeAccel's order -- `ref`: the oracle's order mode
   "embree" against the real reference built WITH Embree (oracle/_ref/pine_ref_embree); `gpu`: PINE_GPU_FLAG_ORDER_EMBREE against
   the oracle's)
Every film must match bit for bit."""
import sys, os, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "gpu":
    import torch  # (before anything else touches the HIP runtime)
import pine_amd as pa
from oracle import oracle

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "pine_ref")


from pine_amd.scenes import random_scene as rnd_scene  # noqa: E402


def main():
    mode, n = sys.argv[1], int(sys.argv[2])
    variety = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # 2: also fractional Uber lobes and Subsurface meshes (in-path RNG)
    base = int(sys.argv[4]) if len(sys.argv) > 4 else 1000  # first seed
    order = sys.argv[5] if len(sys.argv) > 5 else "pine"
    bad = 0
    for seed in range(base, base + n):
        try:
            sc, spp, depth, sampler = rnd_scene(seed, variety=variety)
        except pa.PineError as e:  # e.g. a degenerate random Rect: a legitimate rejection
            print(seed, "scene rejected:", str(e)[:60])
            continue
        ps = sc.describe()
        w, h = sc.camera.film().size
        ref, _ = oracle.render(ps, (w, h), spp, depth, sampler=sampler, order=order)
        if mode == "ref":
            with tempfile.TemporaryDirectory() as tmp:
                sp, fp = os.path.join(tmp, "s.pscene"), os.path.join(tmp, "s.film")
                open(sp, "w").write(ps)
                r = subprocess.run([REF + "_embree" if order == "embree" else REF, "render", sp, str(spp), str(depth), fp] + (["sobol"] if sampler == "sobol" else []),
                                   capture_output=True, text=True, timeout=120, env=dict(os.environ, PINE_REF_ACCEL="embree" if order == "embree" else "bvh"))
                if r.returncode:
                    print(seed, "reference failed:", r.stderr[-200:])
                    bad += 1
                    continue
                other = np.fromfile(fp, dtype=np.float32).reshape(h, w, 4)
        else:
            plan = pa.Plan(sc, spp, depth, sampler=sampler, order=order)
            film = torch.zeros((h, w, 4), device="cuda")
            plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            other = film.cpu().numpy()
            plan.close()
        d = int((ref.view(np.uint32) != other.view(np.uint32)).any(axis=2).sum())
        nan = int(np.isnan(other).sum())
        print(seed, f"{w}x{h} {sampler} spp {spp} depth {depth}: mismatched pixels {d} nan {nan} mean {float(other[..., :3].mean()):.4f}", flush=True)
        bad += d > 0
    print("scenes with mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
