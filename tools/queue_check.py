"""Bring-up check of the stage-queued kernel (PINE_GPU_KERNEL=queue): parity vs oracle on small cases, then timing."""
import sys, os
os.environ["PINE_GPU_KERNEL"] = "queue"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pine_amd
from pine_amd import scenes
from oracle import oracle
def run(scene, spp, depth, reps=1):
    w, h = scene.camera.film().size
    plan = pine_amd.Plan(scene, spp, depth, timing=True)
    film = torch.full((h, w, 4), -1.0, device="cuda")
    for _ in range(reps):
        plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = plan.stats()
    import ctypes as C
    from pine_amd import _lib
    out = (C.c_uint64 * 16)()
    _lib.check(_lib.lib.pine_gpu_plan_debug_sections(plan._h, out))
    if out[15]:
        print("  !! queue kernel bailed out: count", out[15], "code", out[12], "a", out[13], "b", hex(out[14]), flush=True)
    return film.cpu().numpy(), st
def parity(name, scene, spp, depth):
    w, h = scene.camera.film().size
    f, st = run(scene, spp, depth)
    ref, ost = oracle.render(scene.describe(), (w, h), spp, depth)
    bad = int((ref.view(np.uint32) != f.view(np.uint32)).any(axis=2).sum())
    print(f"{name}: {w}x{h} spp{spp} d{depth} threads/block={st.block_threads} grid={st.grid_blocks} mismatched_px={bad}/{w*h} V={st.vertices/st.camera_samples:.4f}/{ost.vertices/ost.camera_samples:.4f} shadow={st.shadow_rays}/{ost.shadow_rays}", flush=True)
    return bad
bad = 0
bad += parity("tiny", scenes.cbox((16, 16)), 4, 3)
bad += parity("cbox64", scenes.cbox((64, 64)), 16, 4)
bad += parity("cbox readme", scenes.cbox((96, 80), "readme"), 64, 8)
bad += parity("ragged", scenes.cbox((45, 37)), 8, 3)
if bad == 0 and len(sys.argv) > 1:
    f, st = run(scenes.cbox((640, 640)), 256, 8, 3)
    print(f"C2: trace {st.trace_ms:.2f} ms -> {st.camera_samples/st.trace_ms*1e-3:.0f} Msamples/s (kernel only)  V={st.vertices/st.camera_samples:.4f}")
