"""One-off robustness check: a 40 000-cone scene (deep BVH, scene records in global memory) against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pine_amd as pa
from pine_amd import scenes
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sc = scenes.classic_cones((96, 48), n)
t0 = time.time()
plan = pa.Plan(sc, 8, 5, timing=True)
print(f"plan built in {time.time()-t0:.1f}s", flush=True)
film = torch.zeros((48, 96, 4), device="cuda")
plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
st = plan.stats()
print(f"trace {st.trace_ms:.2f} ms lds {st.lds_bytes} block {st.block_threads}", flush=True)
ref, _ = oracle.render(sc.describe(), (96, 48), 8, 5)
print("mismatched pixels", int((ref.view(np.uint32) != film.cpu().numpy().view(np.uint32)).any(axis=2).sum()))
