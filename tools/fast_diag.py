import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import pine_amd as pa
from pine_amd import scenes, _lib
def run(sc, spp, depth, flags):
    w, h = sc.camera.film().size
    plan = pa.Plan(sc, spp, depth, flags=flags)
    film = torch.zeros((h, w, 4), device="cuda")
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    s = plan.read_samples(); f = film.cpu().numpy(); plan.close(); return f, s
for name, sc, spp, depth in (("cbox_readme", scenes.cbox((64, 64), "readme"), 256, 8), ("cbox_rect", scenes.cbox((64, 64), "readme", False), 256, 8), ("cones", scenes.classic_cones((96, 48), 100), 64, 6)):
    fe, se = run(sc, spp, depth, 0); ff, sf = run(sc, spp, depth, _lib.FLAG_FAST)
    e = np.minimum(fe[..., :3].astype(np.float64), 8); f = np.minimum(ff[..., :3].astype(np.float64), 8)
    rel = np.linalg.norm(f - e, axis=-1) / (np.linalg.norm(e, axis=-1) + 1e-3)
    print(name, "pixel rel percentiles 50/90/99/max", [float(np.percentile(rel, q)) for q in (50, 90, 99, 100)], "rmse", float(np.sqrt(((f - e) ** 2).mean())))
    dv = se[..., 3] != sf[..., 3]
    rs = np.linalg.norm(sf[..., :3].astype(np.float64) - se[..., :3], axis=-1) / (np.linalg.norm(se[..., :3].astype(np.float64), axis=-1) + 1e-3)
    print("   samples: different vertex count %.5f ; rel diff > 1e-3: %.5f ; > 1e-5: %.5f ; median rel %.2e" % (dv.mean(), (rs > 1e-3).mean(), (rs > 1e-5).mean(), np.median(rs[rs > 0]) if (rs > 0).any() else 0))
