"""SURVEY.md 8(c) fixture 4: the BVH of the REAL reference (`pine_ref bvh`, bvh.cpp:30-147, 453-495) against this
repo's builders, node for node; and (GPU) the order in which the device traversals test primitives against the order in
which the reference's BVH::intersect / BVH::hit do (bvh.cpp:321-451, 497-548), ray by ray."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

SCENES = ["cbox", "cones10k", "sss_mesh", "zoo"]


def _scene(name):
    import pine_amd as pa  # noqa: F401
    from pine_amd import scenes
    return {"cbox": lambda: scenes.cbox((64, 64), "readme"), "cones10k": lambda: scenes.classic_cones((720, 360), 100),
            "sss_mesh": lambda: scenes.sss((64, 64), 2, emissive_mesh=True), "zoo": lambda: scenes.shapes_zoo((48, 48))}[name]()


def canonical_tree(sc, device=None):
    """This repo's flattened BVH (64-byte nodes numbered breadth-first, leaf ranges into one primitive list) as the
    canonical pre-order stream `pine_ref bvh` writes: independent of node numbering.  device: build it on that GPU."""
    from pine_amd import _lib
    n = _lib.check(_lib.lib.pine_gpu_scene_build_accel(sc._h) if device is None else _lib.lib.pine_gpu_scene_build_accel_device(sc._h, device))
    nodes = np.zeros((max(n, 1), 16), np.uint32)
    prims = np.zeros(4_000_000, np.int32)
    npr = _lib.lib.pine_gpu_scene_accel_dump(sc._h, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes, prims.ctypes.data_as(C.POINTER(C.c_int32)), prims.size)
    bv = np.zeros(5 * 4096, np.int32)
    nb = _lib.check(_lib.lib.pine_gpu_scene_accel_bvhs(sc._h, bv.ctypes.data_as(C.POINTER(C.c_int32)), bv.size))
    bv = bv[:5 * nb].reshape(nb, 5)
    prims = prims[:npr]
    nodes_i = nodes.view(np.int32)
    out = [np.uint32(nb)]

    def leaf(start, count, base):
        out.append(np.uint32(0x80000000 | count))
        out.extend(np.uint32(int(p) - base) for p in prims[start:start + count])

    def stream(root, root_start, root_count, base):
        if root_count > 0:
            return leaf(root_start, root_count, base)
        if root < 0:
            out.append(np.uint32(0x80000000))
            return
        stack = [("node", root)]
        while stack:
            kind, v = stack.pop()
            if kind == "leaf":
                leaf(v[0], v[1], base)
                continue
            nd, ni = nodes[v], nodes_i[v]
            out.append(np.uint32(0x40000000))
            out.extend(nd[0:3])    # child 0 lower
            out.extend(nd[6:9])    # child 1 lower
            out.extend(nd[3:6])    # child 0 upper
            out.extend(nd[9:12])   # child 1 upper
            kids = []
            for c in range(2):
                if ni[14 + c] > 0:
                    kids.append(("leaf", (int(ni[12 + c]), int(ni[14 + c]))))
                else:
                    kids.append(("node", int(ni[12 + c])))
            stack.append(kids[1])
            stack.append(kids[0])

    for b in range(nb):
        root, rs, rc, base, geom = (int(x) for x in bv[b])
        if b > 0:
            out.append(np.uint32(geom))
        stream(root, rs, rc, 0)  # (mesh entries of the primitive list are triangle indices within their mesh)
    return np.array(out, dtype=np.uint32)


@pytest.mark.parametrize("name", SCENES)
def test_host_bvh_build_equals_the_reference_tree(name):
    """Boxes, topology and the order of the primitives inside every leaf (which decides the order-dependent OBB results)
    of this repo's level-synchronous host build == the reference's recursive build_sah_binned, bit for bit."""
    z = np.load(os.path.join(GOLDEN, f"bvh_{name}.npz"))
    sc = _scene(name)
    assert sc.describe() == str(z["pscene"])
    mine = canonical_tree(sc)
    ref = z["tree"]
    assert mine.size == ref.size, (mine.size, ref.size)
    bad = np.nonzero(mine != ref)[0]
    assert bad.size == 0, f"first difference at word {bad[0]}: {mine[bad[0]]:#x} vs {ref[bad[0]]:#x}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_device_bvh_build_equals_the_reference_tree(name):
    """The level-synchronous build on the GPU (decide / scan / split kernels, pine_bvh_build_device.h) against the
    reference's own tree -- not against this repo's host build."""
    z = np.load(os.path.join(GOLDEN, f"bvh_{name}.npz"))
    sc = _scene(name)
    assert sc.describe() == str(z["pscene"])
    mine = canonical_tree(sc, device=0)
    assert np.array_equal(mine, z["tree"])


def _parse_trav(words, nrays):
    """-> per ray: (closest test words, (hit, geometry, triangle, tmax bits), any-hit test words, hit)"""
    out = []
    i = 0
    for _ in range(nrays):
        n = int(words[i])
        c = words[i + 1:i + 1 + n]
        res = tuple(int(x) for x in words[i + 1 + n:i + 5 + n])
        i += 5 + n
        m = int(words[i])
        a = words[i + 1:i + 1 + m]
        h = int(words[i + 1 + m])
        i += 2 + m
        out.append((c, res, a, h))
    assert i == len(words)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("flat", [0, 1])
@pytest.mark.parametrize("name", SCENES)
def test_device_traversal_order_equals_the_reference(name, flat):
    """Every primitive test of 1000 rays, in order, closest hit and any hit, and the results (hit, geometry, triangle,
    tmax bits): the nested traversal of the scene-in-LDS kernel variants (flat = 0) and the flat state machine of the
    others (flat = 1) against the reference's BVH::intersect / BVH::hit."""
    from pine_amd import _lib
    z = np.load(os.path.join(GOLDEN, f"bvh_{name}.npz"))
    sc = _scene(name)
    assert sc.describe() == str(z["pscene"])
    rays = np.ascontiguousarray(z["rays"])
    ref = _parse_trav(z["trav"], len(rays))
    cap = 2 + max(max(len(c), len(a)) for c, _, a, _ in ref)
    out = np.zeros((len(rays), 2 * cap + 5), np.uint32)
    _lib.check(_lib.lib.pine_gpu_test_traverse(sc._h, 0, rays.ctypes.data_as(_lib.c_f_p), len(rays), flat, cap,
                                               out.ctypes.data_as(C.POINTER(C.c_uint32))))
    bv = np.zeros(5 * 4096, np.int32)
    nb = _lib.check(_lib.lib.pine_gpu_scene_accel_bvhs(sc._h, bv.ctypes.data_as(C.POINTER(C.c_int32)), bv.size))
    meshes = {int(g) for g in bv[:5 * nb].reshape(nb, 5)[1:, 4]}
    for r, (c, res, a, h) in enumerate(ref):
        o = out[r]
        assert int(o[0]) == len(c) and np.array_equal(o[1:1 + len(c)], c), f"ray {r}: closest-hit test order"
        got = tuple(int(x) for x in o[cap:cap + 4])
        if res[0] and res[1] not in meshes:  # (the reference's triangle word is only meaningful when the winner is a mesh: bvh.cpp:519-523)
            got, res = got[:2] + got[3:], res[:2] + res[3:]
        assert got == res, f"ray {r}: closest-hit result"
        assert int(o[2 * cap + 4]) == h, f"ray {r}: any-hit result"
        # (an any-hit query may stop at the first hit: the reference's order up to there)
        assert int(o[cap + 4]) == len(a) and np.array_equal(o[cap + 5:cap + 5 + len(a)], a), f"ray {r}: any-hit test order"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cbox", "mats"])
def test_device_per_vertex_terms_equal_the_reference(name):
    """SURVEY.md 8(c) fixture 5: nee, f, cosine, pdf, is_delta, mis, light pdf and the clamped Lo of EVERY radiance()
    invocation of 256 paths (8x8 film, 4 spp) against `pine_ref vertices` -- the reference's own intersect / light
    sampler / bxdf objects driven by a restated radiance() whose film the driver checked against render()'s.  A
    difference is then a difference in one named term of one vertex, not in a pixel."""
    import torch
    import pine_amd as pa
    from pine_amd import _lib, scenes
    z = np.load(os.path.join(GOLDEN, f"vertices_{name}.npz"))
    sc = scenes.cbox((8, 8), "readme") if name == "cbox" else scenes.materials_zoo((8, 8))
    assert sc.describe() == str(z["pscene"])
    spp, depth, rec = int(z["spp"]), int(z["depth"]), z["records"]
    plan = pa.Plan(sc, spp, depth, flags=_lib.FLAG_VERTEX_LOG)
    n = _lib.lib.pine_gpu_plan_vertex_log(plan._h, None, 0)
    assert n == 8 * 8 * spp * depth * 16, _lib.last_error()
    film = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    plan.check()
    log = np.zeros(n, np.float32)
    assert _lib.lib.pine_gpu_plan_vertex_log(plan._h, log.ctypes.data_as(_lib.c_f_p), n) == n, _lib.last_error()
    plan.close()
    log = log.reshape(8, 8, spp, depth, 16)
    names = ["kind", "length", "nee.x", "nee.y", "nee.z", "f.x", "f.y", "f.z", "cosine", "pdf", "is_delta", "mis", "light_pdf", "Lo.x", "Lo.y", "Lo.z"]
    i = 0
    paths = 0
    for y in range(8):
        for x in range(8):
            for s in range(spp):
                k = int(rec[i])
                r = rec[i + 1:i + 1 + 16 * k].reshape(k, 16)
                i += 1 + 16 * k
                got = log[y, x, s, :k]
                for v in range(k):
                    for j in range(16):
                        assert got[v, j].view(np.uint32) == r[v, j].view(np.uint32), \
                            f"pixel ({x},{y}) sample {s} vertex {v}: {names[j]} {got[v, j]!r} vs the reference's {r[v, j]!r}"
                assert (log[y, x, s, k:] == 0).all(), f"pixel ({x},{y}) sample {s}: more vertices than the reference's {k}"
                paths += 1
    assert i == rec.size and paths == 256
