"""On-GPU check of the other BASELINE configs (C3 large film, C4 10k cones, C5 SSS): timing + parity on a
cropped/small variant against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pine_amd
from pine_amd import scenes
from oracle import oracle

def timed(scene, spp, depth, reps=2):
    w, h = scene.camera.film().size
    plan = pine_amd.Plan(scene, spp, depth, timing=True)
    film = torch.zeros((h, w, 4), device="cuda")
    for _ in range(reps):
        plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = plan.stats()
    f = film.cpu().numpy()
    plan.close()
    return f, st

def parity(name, scene, spp, depth):
    w, h = scene.camera.film().size
    f, st = timed(scene, spp, depth, 1)
    ref, ost = oracle.render(scene.describe(), (w, h), spp, depth)
    bad = int((ref.view(np.uint32) != f.view(np.uint32)).any(axis=2).sum())
    rel = np.linalg.norm(ref[..., :3] - f[..., :3], axis=2) / (np.linalg.norm(ref[..., :3], axis=2) + 1e-3)
    print(f"{name}: parity {w}x{h} spp{spp} d{depth}: mismatched_px={bad}/{w*h} max_rel_l2={rel.max():.2e} V={st.vertices/st.camera_samples:.3f} (oracle {ost.vertices/ost.camera_samples:.3f})", flush=True)

print("C3: cbox 1920x1080, 256 effective spp (1024 requested), depth 8")
f, st = timed(scenes.cbox((1920, 1080)), 1024, 8)
print(f"   trace {st.trace_ms:.2f} ms resolve {st.resolve_ms:.2f} ms -> {st.camera_samples/(st.trace_ms+st.resolve_ms+st.prepass_ms)*1e-3:.1f} Msamples/s spp_eff={st.spp_effective}", flush=True)
parity("C3-small", scenes.cbox((192, 108)), 1024, 8)

print("C4: classic + 10k cones 720x360, BlueSampler(64), depth 6")
sc = scenes.classic_cones((720, 360), 100)
f, st = timed(sc, 64, 6)
print(f"   trace {st.trace_ms:.2f} ms -> {st.camera_samples/(st.trace_ms+st.resolve_ms+st.prepass_ms)*1e-3:.1f} Msamples/s V={st.vertices/st.camera_samples:.3f} lds={st.lds_bytes} grid={st.grid_blocks}", flush=True)
parity("C4-small", scenes.classic_cones((144, 72), 100), 16, 6)

print("C5: SSS icosphere(1280 tris) 640x640, 256 effective spp, depth 8 (serial RNG mode)")
sc = scenes.sss((640, 640), 3)
f, st = timed(sc, 512, 8, 1)
print(f"   trace {st.trace_ms:.2f} ms -> {st.camera_samples/(st.trace_ms+st.resolve_ms+st.prepass_ms)*1e-3:.1f} Msamples/s V={st.vertices/st.camera_samples:.3f} spi={st.samples_per_item}", flush=True)
parity("C5-small", scenes.sss((64, 64), 3), 64, 8)
