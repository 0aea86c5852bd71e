#!/usr/bin/env python3
"""usage (GPU box): tools/order_speed.py  -- path-kernel time of C2 / C3-sized cbox and the 10 000-cone scene in pine-BVH order and in
EmbreeAccel's order (PINE_GPU_FLAG_ORDER_EMBREE), precompiled kernels and the scene's own kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before anything else touches the HIP runtime)

import pine_amd as pa  # noqa: E402
from pine_amd import scenes  # noqa: E402


def run(name, sc, spp, depth, order, specialize):
    w, h = sc.camera.film().size
    plan = pa.Plan(sc, spp, depth, order=order, specialize=specialize, timing=True)
    film = torch.zeros((h, w, 4), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.launch(film.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        plan.launch(film.data_ptr(), s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    st = plan.stats()
    print(f"{name:8s} order {order:6s} specialize {str(specialize):5s}: {ms:8.2f} ms/launch  path kernel {st.trace_ms:8.2f} ms  features {st.kernel_features:#x} "
          f"specialized {st.specialized} threads {st.block_threads}", flush=True)
    plan.close()


def main():
    cases = [("c2", scenes.cbox((640, 640), "committed"), 256, 8), ("c2readme", scenes.cbox((640, 640), "readme"), 256, 8),
             ("c4", scenes.classic_cones((720, 360), 100), 64, 6)]
    for name, sc, spp, depth in cases:
        for order in ("pine", "embree"):
            for spec in (False, True):
                run(name, sc, spp, depth, order, spec)


if __name__ == "__main__":
    main()
