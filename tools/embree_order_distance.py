#!/usr/bin/env python3
"""usage (this container: needs oracle/_ref/pine_ref_embree, `make -C oracle embree`): tools/embree_order_distance.py > profiles/rNN_embree_order_distance.txt
How far the two traversal orders are from the REAL reference built with EmbreeAccel as the number of top-level primitives grows:
cbox (README camera, 48x48, 16 spp, depth 5) plus n random primitives -- rotated+scaled Boxes, Spheres, scaled Boxes, the kinds
whose image depends on the test order (bbox.cpp:149-171) -- rendered by oracle/_ref/pine_ref_embree and by the CPU restatement in
pine-BVH order and nearest-bounds-first order (PINE_GPU_FLAG_ORDER_NEAREST).  DESIGN.md 1 quotes the table."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["PINE_REF_ACCEL"] = "embree"
import make_golden as mg  # noqa: E402
from oracle import oracle  # noqa: E402
from pine_amd import scenes  # noqa: E402
from pine_amd.api import AABB, Box, Sphere, rotate_y, scale, translate  # noqa: E402

mg.REF = os.path.join(ROOT, "oracle", "_ref", "pine_ref_embree")


def scene_with(extra, seed):
    rng = np.random.default_rng(seed)
    s = scenes.cbox((48, 48), "readme")
    for i in range(extra):
        c = rng.uniform([-0.8, 0.1, 0.3], [0.8, 1.6, 1.8]).tolist()
        if i % 3 == 0:
            s.add(Box(AABB([0, 0, 0], [1, 1, 1]), translate(c) * rotate_y(float(rng.uniform(-1, 1))) * scale(rng.uniform(0.1, 0.35, 3).tolist())), "floor")
        elif i % 3 == 1:
            s.add(Sphere(c, float(rng.uniform(0.05, 0.2))), "red")
        else:
            s.add(Box(AABB([0, 0, 0], [1, 1, 1]), translate(c) * scale(rng.uniform(0.1, 0.3, 3).tolist())), "green")
    return s


print("top-level primitives | order | identical-pixel share vs real EmbreeAccel | RMSE")
with tempfile.TemporaryDirectory() as tmp:
    for extra in (0, 1, 2, 4, 8, 12, 24, 40, 55):
        ps, film, _ = mg.ref_film(scene_with(extra, 7 + extra), 16, 5, tmp)
        for order in ("embree", "nearest", "pine"):
            o, _ = oracle.render(ps, (48, 48), 16, 5, order=order)
            print(f"{extra + 8:3d} | {order:7s} | {np.all(o == film, axis=-1).mean():.4f} | {np.sqrt(((o - film) ** 2).mean()):.4g}", flush=True)
