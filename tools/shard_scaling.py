"""Single-GPU rehearsal of strong scaling (no communication): for world = 1, 2, 4, 8 the time of EVERY rank's shard of a
BASELINE config (8x8 tiles dealt round-robin), one after the other on this GPU -> the slowest rank's prepass + path kernel
+ resolve is what a step would cost, and base / (world x slowest) the efficiency if communication were free.
usage: python tools/shard_scaling.py [c2 c3 c4 c5 ...]
Every plan WAITS for its scene's kernel (specialize=True: the steady state of the library's default mode -- with the default
itself the first plans of a geometry would run the precompiled kernel while the compiler works, and the rows would not be
comparable); PINE_GPU_SPECIALIZE=0 measures the precompiled kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pine_amd
import bench
cfgs = [a for a in sys.argv[1:]] or ["c2"]
stream = torch.cuda.current_stream().cuda_stream
for cfg in cfgs:
    build, spp, depth, text, _ = bench._configs()[cfg]
    scene = build()
    w, h = scene.camera.film().size
    film = torch.zeros((h, w, 4), device="cuda")
    base = None
    n = 3 if cfg == "c5" else 6
    for world in (1, 2, 4, 8):
        worst, worst_rank, mean = 0.0, 0, 0.0
        for rank in range(world):
            plan = pine_amd.Plan(scene, spp, depth, shard_rank=rank, shard_world=world, timing=True, specialize=True)
            plan.launch(film.data_ptr(), stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                plan.launch(film.data_ptr(), stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
            st = plan.stats()
            mean += dt / world
            if dt > worst:
                worst, worst_rank, wst = dt, rank, st
            plan.close()
        if world == 1:
            base = worst
        print(f"{cfg} world={world}: slowest shard (rank {worst_rank}) {worst:.3f} ms per step (trace {wst.trace_ms:.3f} prepass {wst.prepass_ms:.3f} resolve {wst.resolve_ms:.3f}), "
              f"mean over ranks {mean:.3f} ms; efficiency if comm were free: {base / world / worst:.3f}", flush=True)
    del film
    torch.cuda.empty_cache()
