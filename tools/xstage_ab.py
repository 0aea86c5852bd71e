"""Diagnostic (GPU box): path-kernel time of F_LDS_TOP scenes with traversal stages (PINE_GPU_XSTAGE=1: XS / XC queues, lanes
refilled) against the flat traversal inside stages S / T (=0), and what plan_build picks by itself."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pine_amd as pa
from pine_amd import scenes
cases = {
    "cones 20x20 (nodes fit)": (lambda: scenes.classic_cones((720, 360), 20), 64, 6),
    "cones 40x40": (lambda: scenes.classic_cones((720, 360), 40), 64, 6),
    "cones 100x100 (C4)": (lambda: scenes.classic_cones((720, 360), 100), 64, 6),
    "glossy icosphere 1280": (lambda: scenes.sss((512, 512), 3, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15)), 64, 6),
    "glossy icosphere 20480": (lambda: scenes.sss((512, 512), 5, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15)), 64, 6),
    "sss icosphere 5120": (lambda: scenes.sss((512, 512), 4), 64, 8),
    "shapes zoo": (lambda: scenes.shapes_zoo((512, 512)), 64, 5),
}
for name, (build, spp, depth) in cases.items():
    row = []
    for mode in ("1", "0", None):
        if mode is None:
            os.environ.pop("PINE_GPU_XSTAGE", None)
        else:
            os.environ["PINE_GPU_XSTAGE"] = mode
        sc = build()
        w, h = sc.camera.film().size
        plan = pa.Plan(sc, spp, depth, timing=True)
        film = torch.zeros((h, w, 4), device="cuda")
        for _ in range(3):
            plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        st = plan.stats()
        row.append(f"{'auto' if mode is None else 'stages' if mode == '1' else 'inline'} {st.trace_ms:8.2f} ms (lds {st.lds_bytes})")
        plan.close()
    print(f"{name:28s}", " | ".join(row), flush=True)
