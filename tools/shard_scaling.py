"""Single-GPU rehearsal of strong scaling: time rank 0's shard of the bench workload for world = 1, 2, 4, 8
(no communication) -> what the path kernel + prepass + resolve cost per rank, and the ideal-scaling ratio."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pine_amd
from pine_amd import scenes
scene = scenes.cbox((640, 640), "committed")
film = torch.zeros((640, 640, 4), device="cuda")
stream = torch.cuda.current_stream().cuda_stream
base = None
for spi in ([0] + [int(a) for a in sys.argv[1:]]):
    for world in (1, 2, 4, 8):
        plan = pine_amd.Plan(scene, 256, 8, shard_rank=0, shard_world=world, samples_per_item=spi, timing=True)
        for _ in range(2):
            plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        st = plan.stats()
        if world == 1:
            base = dt
        print(f"spi={st.samples_per_item} world={world}: step {dt:.3f} ms (trace {st.trace_ms:.3f} prepass {st.prepass_ms:.3f} resolve {st.resolve_ms:.3f}) "
              f"efficiency if comm were free: {base / world / dt:.3f}", flush=True)
        plan.close()
