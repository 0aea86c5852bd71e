"""PINE_GPU_FLAG_ORDER_EMBREE: the order in which the reference's default accel hands the shapes to their tests.

The fixture tests/golden/embree_order.npz holds what the REAL Embree (the reference's vendored 4.3.1, built in the container by
`make -C oracle embree probe`) does with pine's user-primitive registration: 14 box sets of 1 .. 400 primitives x 48 rays, the
geometry ids in the order the intersect callback was called, the reported hit and the final tfar
(tools/embree_order_check.py --fixture; oracle/embree_probe.cpp).  CPU: the oracle's restatement must reproduce every call
sequence, and the PRODUCT's host-side hierarchy (pine_amd/csrc/pine_embree_order.h, through the C ABI, no GPU) must be the
oracle's.  GPU: the device traversal against the oracle ray by ray."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = 14


def _fixture():
    return np.load(os.path.join(GOLDEN, "embree_order.npz"))


@pytest.mark.parametrize("c", range(CASES))
def test_restated_order_equals_the_real_embree_call_sequences(oracle, c):
    z = _fixture()
    boxes, rays, hits, calls = z[f"boxes{c}"], z[f"rays{c}"], z[f"hits{c}"], z[f"calls{c}"]
    n = len(boxes)
    lib = oracle.lib()
    lib.oracle_embree_order.restype = C.c_int
    for k in range(len(rays)):
        ids = (C.c_int * (n + 1))()
        hid, tf = C.c_int(0), C.c_float(0)
        cnt = lib.oracle_embree_order(np.ascontiguousarray(boxes).ctypes.data_as(C.c_void_p), n, np.ascontiguousarray(rays[k]).ctypes.data_as(C.c_void_p),
                                      np.ascontiguousarray(hits[k]).ctypes.data_as(C.c_void_p), ids, n + 1, C.byref(hid), C.byref(tf))
        want = calls[k, 1:1 + calls[k, 0]].tolist()
        assert list(ids[:cnt]) == want, (c, k)
        assert hid.value == int(z[f"hit{c}"][k]) and np.float32(tf.value).tobytes() == z[f"tfar{c}"][k].tobytes(), (c, k)


def _tree_words(fn, boxes):
    n = len(boxes)
    cap = 1 + 8 * max(n, 1)
    words = (C.c_int * cap)()
    k = fn(np.ascontiguousarray(boxes, np.float32).ctypes.data_as(C.POINTER(C.c_float)), n, words, cap)
    assert k > 0
    return list(words[:k])


def test_product_hierarchy_is_the_oracles(oracle):
    """The library's own host code builds the hierarchy the oracle builds (root word + 8 child words per node in creation order):
    the fixture's box sets, degenerate ones (identical boxes: the fallback split; flat and point boxes; one invalid box, which
    Embree leaves out) and random ones."""
    from pine_amd import _lib
    lib = _lib.lib
    olib = oracle.lib()
    olib.oracle_embree_tree.restype = C.c_int
    olib.oracle_embree_tree.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int), C.c_int]
    z = _fixture()
    sets = [z[f"boxes{c}"] for c in range(CASES)]
    same = np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (11, 1))
    sets.append(same)
    flat = np.array([[i, 0, 0, i, 0, 0] for i in range(9)], np.float32)  # points on a line
    sets.append(flat)
    bad = np.array([[0, 0, 0, 1, 1, 1], [0, 0, 0, np.inf, 1, 1], [2, 0, 0, 3, 1, 1], [1, 2, 3, 0, 2, 3]], np.float32)  # infinite, inverted
    sets.append(bad)
    x = (2.0 ** (np.arange(2000) * 0.05)).astype(np.float32)  # a geometric progression: the deepest hierarchy float32 boxes give (116 stack entries per ray)
    sets.append(np.stack([x, x * 0, x * 0, x * np.float32(1.01), x * 0 + 1, x * 0 + 1], 1).astype(np.float32))
    rng = np.random.default_rng(77)
    for n in (7, 8, 9, 17, 65, 300, 2000):
        c = rng.uniform(-5, 5, (n, 3))
        e = rng.uniform(0, 1, (n, 3)) ** 3
        sets.append(np.concatenate([c - e, c + e], axis=1).astype(np.float32))
    for boxes in sets:
        assert _tree_words(lib.pine_gpu_test_embree_tree, boxes) == _tree_words(olib.oracle_embree_tree, boxes), len(boxes)
    words = _tree_words(lib.pine_gpu_test_embree_tree, bad)
    leaves = sorted(~w for w in words if w < 0 and w != -2**31)
    assert leaves == [0, 2], leaves  # the infinite and the inverted box are not in the hierarchy


@pytest.mark.parametrize("c", range(3))
def test_restated_triangle_test_equals_the_real_embree(oracle, c):
    """Meshes are Embree triangle geometry under EmbreeAccel: tests/golden/embree_triangles.npz holds the REAL Embree's closest hit --
    primitive, t, barycentrics, geometric normal -- for 200 rays on each of an icosphere, a soup of random triangles with slivers,
    and large far-away triangles (tools/embree_order_check.py --triangles: 60 000 rays, 0 differ; oracle/embree_tri_probe.cpp)."""
    z = np.load(os.path.join(GOLDEN, "embree_triangles.npz"))
    verts, idx, rays, want = z[f"verts{c}"], z[f"idx{c}"], z[f"rays{c}"], z[f"want{c}"]
    lib = oracle.lib()
    lib.oracle_embree_triangles.restype = C.c_int
    got = np.zeros((len(rays), 7), np.float32)
    lib.oracle_embree_triangles(np.ascontiguousarray(verts, np.float32).ctypes.data_as(C.c_void_p), np.ascontiguousarray(idx, np.uint32).ctypes.data_as(C.c_void_p),
                                len(idx), np.ascontiguousarray(rays, np.float32).ctypes.data_as(C.c_void_p), C.c_int64(len(rays)), got.ctypes.data_as(C.c_void_p))
    hit = want[:, 0] >= 0
    assert hit.sum() >= 100
    assert (got[:, 0] == want[:, 0]).all()
    assert (got[:, 1].view(np.uint32) == want[:, 1].view(np.uint32)).all()  # tfar
    assert (got[hit][:, 2:].view(np.uint32) == want[hit][:, 2:].view(np.uint32)).all()  # u, v, Ng


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["clutter63", "xshapes", "cones12", "cbox", "mesh"])
def test_device_traversal_hands_the_shapes_over_in_the_oracles_order(oracle, which):
    """The device's closest-hit query in EmbreeAccel's order, ray by ray, against the oracle's: the geometry indices handed to their
    tests, the winner and its distance -- 1500 camera-like, scene-crossing and short rays per scene."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from pine_amd import scenes, _lib
    from make_golden import bvh_rays
    sc = {"clutter63": lambda: scenes.cbox_clutter((48, 48), 55, 62), "xshapes": lambda: scenes.xshapes_zoo((48, 48)),
          "cones12": lambda: scenes.classic_cones((90, 45), 12), "cbox": lambda: scenes.cbox((64, 64), "readme"),
          "mesh": lambda: scenes.sss((48, 48), 2, emissive_mesh=True)}[which]()
    rays = np.ascontiguousarray(bvh_rays(sc, 1500, 5), np.float32)
    cap = 200
    n = len(rays)
    olib = oracle.lib()
    olib.oracle_embree_traverse.restype = C.c_int
    want = np.zeros((n, cap + 10), np.uint32)
    assert olib.oracle_embree_traverse(sc.describe().encode(), rays.ctypes.data_as(C.c_void_p), C.c_int64(n), cap, want.ctypes.data_as(C.c_void_p)) == 0
    got = np.zeros((n, 2 * cap + 5), np.uint32)
    _lib.check(_lib.lib.pine_gpu_test_traverse(sc._h, 0, rays.ctypes.data_as(_lib.c_f_p), n, 2, cap, got.ctypes.data_as(C.POINTER(C.c_uint32))))
    assert (got[:, 0] < cap).all() and (want[:, 0] < cap).all()
    for k in range(n):
        cnt = int(want[k, 0])
        words = got[k, 1:1 + int(got[k, 0])]
        words = words[(words & 0x40000000) == 0]  # (the device also logs the triangles it tests inside a mesh)
        assert len(words) == cnt and (words == want[k, 1:1 + cnt]).all(), (k, words, want[k, :cnt + 1])
    assert (got[:, cap] == want[:, cap]).all()  # hit
    hitm = want[:, cap] == 1
    assert (got[hitm, cap + 1] == want[hitm, cap + 1]).all()  # geometry
    assert (got[:, cap + 3] == want[:, cap + 2]).all()  # tmax bits
    assert (got[:, 2 * cap + 4] == want[:, cap + 3]).all()  # the any-hit query
