"""GPU: fuzz the BAKED traversal of scene-specialised kernels against the CPU oracle: seeded random rooms of
axis-aligned Rects in every orientation (flipped or not, either edge order, negative edges, walls that overlap, touch and share
edges -- exact ties in t), plus a few other shapes, seen by cameras inside and outside, on the axes and off.
usage: python tools/fuzz_baked.py N [first_seed] [mesh]  -- every film must match bit for bit, every plan must be level 2.
With `mesh`: each room also holds ONE mesh (an icosphere or a few loose triangles; Subsurface, Diffuse, Glossy or Emissive) anywhere
-- inside, through a wall, outside -- so that it lands before, between and after the Rects in pine's order: the top level is
baked, the mesh stays with the flat traversal, and the replay rule of DESIGN.md 4.9 item 3 is exercised."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pine_amd as pa
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
base = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
with_mesh = len(sys.argv) > 3 and sys.argv[3] == "mesh"
from pine_amd.scenes import icosphere
stream = torch.cuda.current_stream().cuda_stream
bad = baked = 0
for seed in range(base, base + n):
    r = np.random.default_rng(seed)
    f = lambda lo, hi: float(np.float32(r.uniform(lo, hi)))
    q = lambda: float(r.choice(np.float32([-1, -0.5, 0, 0.25, 0.5, 1, 1.5, 2])))  # coordinates on a coarse grid: shared planes and edges
    s = pa.Scene()
    mats = []
    for i in range(int(r.integers(2, 5))):
        s.add(f"m{i}", pa.Diffuse([f(0.1, 0.95), f(0.1, 0.95), f(0.1, 0.95)]))
        mats.append(f"m{i}")
    pick = lambda: mats[int(r.integers(0, len(mats)))]
    nrect = int(r.integers(3, 8))
    for i in range(nrect):
        axis = int(r.integers(0, 3))
        ua, va = [(1, 2), (2, 0), (0, 1)][axis] if r.random() < 0.5 else [(2, 1), (0, 2), (1, 0)][axis]
        ex, ey = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
        ex[ua] = float(r.choice(np.float32([-2, -1, 0.5, 1, 2, 3]))) if r.random() < 0.7 else f(-2, 2) or 1.0
        ey[va] = float(r.choice(np.float32([-2, -1, 0.5, 1, 2, 3]))) if r.random() < 0.7 else f(-2, 2) or 1.0
        pos = [q(), q() + 1.0, q() + 1.0] if r.random() < 0.7 else [f(-1, 1), f(0, 2), f(0, 2)]
        s.add(pa.Rect(pos, ex, ey, bool(r.integers(0, 2))), pick())
    if with_mesh:
        nrect = min(nrect, 6)
        kind = int(r.integers(0, 4))
        mat = [pa.Subsurface([f(0.5, 1), f(0.5, 1), f(0.5, 1)], f(0.0, 0.4), [f(5, 40), f(5, 40), f(5, 40)]), pa.Diffuse([f(0.2, 0.9)] * 3),
               pa.Glossy([0.9, 0.5, 0.3], f(0.05, 0.4)), pa.Emissive([f(1, 5), f(1, 5), f(1, 5)])][kind]
        c = (f(-1.2, 1.2), f(-0.2, 2.2), f(-0.2, 2.2))
        if r.random() < 0.7:
            vs, fs = icosphere(int(r.integers(0, 3)), f(0.15, 0.6), c)
        else:
            vs = np.float32([[c[0] + f(-0.5, 0.5), c[1] + f(-0.5, 0.5), c[2] + f(-0.5, 0.5)] for _ in range(6)])
            fs = np.uint32([[0, 1, 2], [2, 3, 4], [3, 4, 5], [0, 2, 5]])
        s.add(pa.Mesh(vs, fs), mat)
    for i in range(int(r.integers(0, 10 - nrect - 1 + 1 - (1 if with_mesh else 0)))):
        k = int(r.integers(0, 4))
        c = [f(-0.8, 0.8), f(0.2, 1.6), f(0.4, 1.8)]
        if k == 0: s.add(pa.Sphere(c, f(0.1, 0.4)), pick())
        elif k == 1: s.add(pa.Box(pa.AABB([0, 0, 0], [1, 1, 1]), pa.translate(c) * pa.rotate_y(f(-1, 1)) * pa.scale([f(0.2, 0.6), f(0.2, 0.6), f(0.2, 0.6)])), pick())
        elif k == 2: s.add(pa.Rect(c, [f(-0.6, 0.6), f(-0.6, 0.6), f(-0.6, 0.6)], [f(-0.6, 0.6), f(-0.6, 0.6), f(-0.6, 0.6)]), pick())
        else: s.add(pa.Disk(c, [f(-1, 1), f(-1, 1), f(-1, 1)], f(0.1, 0.5)), pick())
    s.add(pa.Rect([q() * 0.5, 1.9, 1.0], [f(0.2, 0.8), 0, 0], [0, 0, f(0.2, 0.8)]), pa.Emissive([f(5, 40), f(5, 40), f(5, 40)]))
    w, h = int(r.integers(8, 40)), int(r.integers(8, 40))
    if r.random() < 0.4:  # on the axes: rays with zero components
        frm, to = [0.0, 1.0, float(r.choice(np.float32([-4, -2, 1])))], [0.0, 1.0, 2.0]
    else:
        frm, to = [f(-1.5, 1.5), f(0.2, 1.8), f(-4, 1.5)], [f(-0.5, 0.5), f(0.5, 1.5), f(0.5, 2)]
    try:
        s.set(pa.ThinLenCamera(pa.Film([w, h]), frm, to, f(0.15, 0.8)))
        spp, depth = int(r.choice([1, 2, 4, 16, 64])), int(r.integers(1, 9))
        plan = pa.Plan(s, spp, depth, specialize=True)
    except pa.PineError as e:
        print(seed, "rejected:", str(e)[:70])
        continue
    film = torch.zeros((h, w, 4), device="cuda")
    plan.launch(film.data_ptr(), stream)
    torch.cuda.synchronize()
    plan.check()
    level = plan.stats().specialized
    plan.close()
    ref, _ = oracle.render(s.describe(), (w, h), spp, depth)
    a = film.cpu().numpy()
    mism = int((a.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    st_feat = 0
    baked += level == 2
    bad += mism != 0
    print(f"{seed} {w}x{h} spp {spp} depth {depth} rects {nrect} level {level}: mismatched pixels {mism} mean {float(a[..., :3].mean()):.4f}", flush=True)
print(f"scenes with mismatches: {bad}; baked (level 2): {baked}")
