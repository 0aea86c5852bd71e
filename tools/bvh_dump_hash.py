"""md5 of the flattened BVH (nodes + primitive order) of a set of scenes: used to check that a rewritten builder
(host, level-synchronous; device) produces the identical tree.  usage: python tools/bvh_dump_hash.py out.json"""
import ctypes as C, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pine_amd
from pine_amd import scenes, _lib, gltf


def dump(sc):
    n = _lib.check(_lib.lib.pine_gpu_scene_build_accel(sc._h))
    nodes = np.zeros((max(n, 1), 16), np.float32)
    prims = np.zeros(4_000_000, np.int32)
    npr = _lib.lib.pine_gpu_scene_accel_dump(sc._h, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes, prims.ctypes.data_as(C.POINTER(C.c_int32)), prims.size)
    return n, int(npr), hashlib.md5(nodes[:n].tobytes() + prims[:npr].tobytes()).hexdigest()


def all_scenes():
    yield "cbox", scenes.cbox((64, 64))
    yield "c4", scenes.classic_cones((720, 360), 100)
    yield "cones12", scenes.classic_cones((90, 45), 12)
    yield "c5", scenes.sss((64, 64), 3)
    yield "sss_emissive_mesh", scenes.sss((48, 48), 2, emissive_mesh=True)
    yield "zoo", scenes.shapes_zoo((48, 48))
    yield "mats", scenes.materials_zoo((48, 48))
    yield "xshapes", scenes.xshapes_zoo((48, 48))
    yield "glb", gltf.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "import_test.glb"))
    for seed in range(3000, 3012):
        yield f"random{seed}", scenes.random_scene(seed, variety=2)[0]


if __name__ == "__main__":
    out = {name: dump(sc) for name, sc in all_scenes()}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(len(out), "scenes;", "c4:", out["c4"])
