"""Soak test of the path kernels: many launches of several configurations, every film compared with the first
one of its configuration (run-to-run determinism) and the C2 film with the reference md5; reports bail-out
diagnostics of the queue kernel if any spin limit was ever hit.  usage: tools/soak.py [seconds]"""
import sys, os, time, json, hashlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pine_amd
from pine_amd import scenes, _lib
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
golden = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "stats_640.json")))["C2_cbox_640_s256_d8_committed"]["md5"]
configs = [
    ("C2", scenes.cbox((640, 640)), 256, 8, {}),
    ("C2 shard 3/8", scenes.cbox((640, 640)), 256, 8, dict(shard_rank=3, shard_world=8)),
    ("readme 200x120", scenes.cbox((200, 120), "readme"), 64, 8, {}),
    ("tiny 24x16", scenes.cbox((24, 16), "readme"), 16, 5, {}),
    ("classic12", scenes.classic_cones((90, 45), 12), 32, 6, {}),
    ("zoo", scenes.shapes_zoo((48, 48)), 16, 5, {}),
    ("xshapes", scenes.xshapes_zoo((48, 48)), 16, 5, {}),
    ("lights", scenes.lights_zoo((64, 64)), 32, 6, {}),
    ("sobol 64", scenes.cbox((96, 96), "readme"), pine_amd.SobolSampler(64), 6, {}),
    # traversal stages with refill, BSSRDF walk stage, sample tokens (waiting / woken contexts at the end of every launch)
    ("sss 96x96", scenes.sss((96, 96), 2), 64, 8, {}),
    ("sss 24x24 x256", scenes.sss((24, 24), 1), 256, 6, {}),
    ("sss + emissive mesh", scenes.sss((40, 40), 2, emissive_mesh=True), 32, 6, {}),
    ("classic20", scenes.classic_cones((180, 90), 20), 16, 6, {}),
    # tile classes: whole-pixel tiles first, independent items after; a shard; a film of one whole-pixel tile and little else
    ("sss 200x120 x64", scenes.sss((200, 120), 2), 64, 8, {}),
    ("sss 96x96 shard 1/3", scenes.sss((96, 96), 2), 64, 8, dict(shard_rank=1, shard_world=3)),
    ("sss 16x16 x256", scenes.sss((16, 16), 1), 256, 8, {}),
]
plans = []
for name, sc, spp, d, kw in configs:
    w, h = sc.camera.film().size
    plans.append((name, pine_amd.Plan(sc, spp, d, **kw), torch.zeros((h, w, 4), device="cuda"), None))
stream = torch.cuda.current_stream().cuda_stream
t0 = time.time(); n = 0; bad = 0; last = t0
while time.time() - t0 < budget:
    for k, (name, plan, film, first) in enumerate(plans):
        film.fill_(-1.0)
        plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        out = (C.c_uint64 * 16)()
        _lib.check(_lib.lib.pine_gpu_plan_debug_sections(plan._h, out))
        if out[15]:
            print(f"!! {name}: queue kernel bailed out {out[15]} times, code {out[12]} a {out[13]} b {hex(out[14])}", flush=True); bad += 1
        md5 = hashlib.md5(film.cpu().numpy().tobytes()).hexdigest()
        if first is None:
            plans[k] = (name, plan, film, md5)
            if name == "C2" and md5 != golden:
                print("!! C2 md5 differs from the reference", md5, flush=True); bad += 1
        elif md5 != first:
            print(f"!! {name}: film changed between launches", flush=True); bad += 1
        n += 1
    if time.time() - last > 20:
        last = time.time(); print(f"  {n} launches, {bad} problems, {time.time()-t0:.0f}s", flush=True)
print(f"soak: {n} launches in {time.time()-t0:.0f}s, problems: {bad}")
sys.exit(1 if bad else 0)
