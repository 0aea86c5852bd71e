"""Diagnostic (GPU box): path-kernel time of the C5 scene (Subsurface icosphere, serial-RNG mode: one work item = one
pixel's whole sample sequence) against the number of pixels -- is the launch bound by throughput (time ~ pixels) or by the
sequential chain of a pixel's samples (time ~ rounds of pixels over the context slots)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pine_amd
from pine_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for side in [int(a) for a in sys.argv[2:]] or [256, 362, 512, 640, 724, 1024]:
    sc = scenes.sss((side, side), int(os.environ.get("SUBDIV", "3")))
    plan = pine_amd.Plan(sc, spp, 8, timing=True)
    film = torch.zeros((side, side, 4), device='cuda')
    for _ in range(2):
        plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    st = plan.stats()
    px = side * side
    slots = st.grid_blocks * 1024
    print(f'side {side} pixels {px} slots {slots} rounds {px / slots:.2f} trace_ms {st.trace_ms:.1f}  ns/sample {st.trace_ms * 1e6 / (px * spp):.2f}  Msamples/s {px * spp / st.trace_ms * 1e-3:.0f}', flush=True)
    plan.close()
