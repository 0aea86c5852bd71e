"""Host vs device BVH build time (plan creation, first plan of the scene): tools/bvh_build_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pine_amd as pa
from pine_amd import scenes, _lib
pa.Plan(scenes.cbox((16, 16)), 1, 2).close()  # load the module, warm the context
for name, build in (("c4 10k cones", lambda: scenes.classic_cones((720, 360), 100)), ("c5 icosphere", lambda: scenes.sss((640, 640), 3)),
                    ("icosphere 81920 tris", lambda: scenes.sss((64, 64), 6))):
    for flags, what in ((0, "host"), (_lib.FLAG_DEVICE_BVH, "device")):
        ts = []
        for _ in range(3):
            sc = build()
            p = pa.Plan(sc, 4, 4, flags=flags)
            st = p.stats(); ts.append(st.accel_build_ms); dev = st.accel_built_on_device; p.close()
        print(f"{name:24s} {what:6s} build_ms {min(ts):8.3f}  (on device: {dev})")
