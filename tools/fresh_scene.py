"""A geometry the library has never seen, through the DEFAULT call path (no flag): the precompiled kernel renders while the scene's
own kernel compiles in the background, a later launch adopts it.  Prints the path kernel's time and vertex rate before and after,
beside BASELINE's cbox through the same call -- the evidence that the headline's speed is not a property of a pre-built kernel.
usage (GPU box): python3 tools/fresh_scene.py [seed]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pine_amd as pa
from pine_amd import scenes
from oracle import oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
r = np.random.default_rng(seed)
f = lambda lo, hi: float(np.float32(r.uniform(lo, hi)))


def room():
    """A Cornell-like room with its own proportions, box poses and lamp (8 primitives, like cbox; never rendered before)."""
    s = pa.Scene()
    for n, c in (("w", [0.85, 0.85, 0.8]), ("a", [f(0.1, 0.9), f(0.1, 0.9), f(0.1, 0.9)]), ("b", [f(0.1, 0.9), f(0.1, 0.9), f(0.1, 0.9)])):
        s.add(n, pa.Diffuse(c))
    hw, hh, d = f(0.8, 1.3), f(0.8, 1.2), f(1.6, 2.4)
    s.add(pa.Rect([0, 0, d / 2], [2 * hw, 0, 0], [0, 0, d], True), "w")
    s.add(pa.Rect([0, 2 * hh, d / 2], [2 * hw, 0, 0], [0, 0, d]), "w")
    s.add(pa.Rect([-hw, hh, d / 2], [0, 0, d], [0, 2 * hh, 0], True), "a")
    s.add(pa.Rect([hw, hh, d / 2], [0, 0, d], [0, 2 * hh, 0]), "b")
    s.add(pa.Rect([0, hh, d], [2 * hw, 0, 0], [0, 2 * hh, 0], True), "w")
    for _ in range(2):
        m = pa.translate([f(-0.5, 0.3), 0.0, f(0.4, 1.2)]) * pa.rotate_y(f(-0.8, 0.8)) * pa.scale([f(0.3, 0.6), f(0.4, 1.2), f(0.3, 0.6)])
        s.add(pa.Box(pa.AABB([0, 0, 0], [1, 1, 1]), m), "w")
    s.add(pa.Rect([f(-0.3, 0.3), 2 * hh - 0.05, d / 2], [0.15, 0, 0], [0, 0, 0.15]), pa.Emissive([400.0, 300.0, 150.0]))
    s.set(pa.ThinLenCamera(pa.Film([640, 640]), [0, hh, -3.5], [0, hh, 0], 0.28))
    return s


def measure(scene, label, check=None):
    plan = pa.Plan(scene, 256, 8, timing=True)  # the default: no flag
    film = torch.zeros((640, 640, 4), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    seen = {}
    while time.perf_counter() - t0 < 120:
        for _ in range(3):
            plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        st = plan.stats()
        seen[st.specialized] = (st.trace_ms, st.vertices, st.specialize_source)
        if st.specialized != 0 or not st.specialize_pending:
            break
    for _ in range(5):
        plan.launch(film.data_ptr(), stream)
    torch.cuda.synchronize()
    st = plan.stats()
    seen[st.specialized] = (st.trace_ms, st.vertices, st.specialize_source)
    for k, (ms, v, src) in sorted(seen.items()):
        what = {0: "precompiled kernel", 1: "scene's kernel (feature set)", 2: "scene's kernel (baked)", -1: "precompiled (build failed)"}[k]
        print(f"{label}: {what:30s} {ms:7.3f} ms  {v / ms * 1e-6:7.2f} G vertices/s  source {src}  after {time.perf_counter() - t0:.1f} s")
    out = film.cpu().numpy()
    plan.close()
    if check is not None:
        w = 640
        ref, _ = oracle.render(scene.describe(), (w, w), 256, 8, rows=(320, 328))
        print(f"{label}: rows 320..327 equal the CPU restatement's:", bool((ref[320:328].view(np.uint32) == out[320:328].view(np.uint32)).all()))


with tempfile.TemporaryDirectory() as tmp:
    os.environ["PINE_GPU_CACHE_DIR"] = tmp  # (an empty cache: both scenes are first sights)
    measure(scenes.cbox((640, 640), "readme"), "cbox, README camera")
    measure(room(), f"fresh room (seed {seed})", check=True)
