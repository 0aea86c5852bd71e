#!/usr/bin/env python3
"""usage (this container: needs oracle/_ref/pine_ref_embree, `make -C oracle embree`): tools/embree_order_distance.py > profiles/rNN_embree_order_distance.txt
How far the two traversal orders are from the REAL reference built with EmbreeAccel as the number of top-level primitives grows:
cbox (README camera, 48x48, 16 spp, depth 5) plus n random primitives -- rotated+scaled Boxes, Spheres, scaled Boxes, the kinds
whose image depends on the test order (bbox.cpp:149-171) -- rendered by oracle/_ref/pine_ref_embree and by the CPU restatement in
pine-BVH order, in EmbreeAccel's restated order (PINE_GPU_FLAG_ORDER_EMBREE) and in the plain nearest-bounds-first order the
restatement replaced.  DESIGN.md 1 quotes the table."""
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

mg.REF = os.path.join(ROOT, "oracle", "_ref", "pine_ref_embree")


print("top-level primitives | order | identical-pixel share vs real EmbreeAccel | RMSE")
with tempfile.TemporaryDirectory() as tmp:
    for extra in (0, 1, 2, 4, 8, 12, 24, 40, 55):
        ps, film, _ = mg.ref_film(scenes.cbox_clutter((48, 48), extra, 7 + extra), 16, 5, tmp)
        for order in ("embree", "nearest", "pine"):
            o, _ = oracle.render(ps, (48, 48), 16, 5, order=order)
            print(f"{extra + 8:3d} | {order:7s} | {np.all(o == film, axis=-1).mean():.4f} | {np.sqrt(((o - film) ** 2).mean()):.4g}", flush=True)
