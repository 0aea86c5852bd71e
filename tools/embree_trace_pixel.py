#!/usr/bin/env python3
"""usage (this container: needs oracle/_ref/pine_ref_embree): tools/embree_trace_pixel.py <fuzz-seed | gltf | c5_320> <x> <y> [depth] [variety]
Where does a pixel of the restated EmbreeAccel mode part from the REAL reference built with Embree?  The oracle renders the pixel's
row with $PINE_ORACLE_TRACE_PIXEL=x,y -- every accel query of the pixel's paths goes to stderr -- and exactly those rays are put to
the reference's own EmbreeAccel::intersect / hit (`pine_ref_embree accelq`): the first query whose answers differ is the cause.
(How the negative-tfar any-hit semantics and the coplanar-triangle ties of DESIGN.md 1 were found.)"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "pine_ref_embree")


def scene_of(spec, variety):
    if spec == "gltf":
        from pine_amd import gltf
        sc = gltf.load(os.path.join(ROOT, "tests", "golden", "import_test.glb"))
        return sc, 4, 5, "blue"
    from pine_amd import scenes
    if spec == "c5_320":  # BASELINE's Subsurface icosphere at a quarter of the film
        return scenes.sss((320, 320), 3), 512, 8, "blue"
    return scenes.random_scene(int(spec), variety=variety)


def accelq(ps, rays):
    with tempfile.TemporaryDirectory() as tmp:
        sp, rp, op = [os.path.join(tmp, x) for x in ("s.pscene", "r.bin", "o.bin")]
        open(sp, "w").write(ps)
        rays.tofile(rp)
        subprocess.run([REF, "accelq", sp, rp, op], check=True, capture_output=True, env=dict(os.environ, PINE_REF_ACCEL="embree"))
        return np.fromfile(op, np.uint32).reshape(-1, 10)


def mine(ps, rays, cap=64):
    lib = oracle.lib()
    lib.oracle_embree_traverse.restype = C.c_int
    out = np.zeros((len(rays), cap + 10), np.uint32)
    assert lib.oracle_embree_traverse(ps.encode(), rays.ctypes.data_as(C.c_void_p), C.c_int64(len(rays)), cap, out.ctypes.data_as(C.c_void_p)) == 0
    return out[:, cap:cap + 10]


def main():
    spec, x, y = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    variety = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    sc, spp, depth, sampler = scene_of(spec, variety)
    if len(sys.argv) > 4:
        depth = int(sys.argv[4])
    ps = sc.describe()
    size = tuple(sc.camera.film().size)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tools')!r})\n"
            f"import embree_trace_pixel as t\nfrom oracle import oracle\n"
            f"sc, spp, depth, sampler = t.scene_of({spec!r}, {variety})\n"
            f"oracle.render(sc.describe(), {size}, spp, {depth}, threads=1, sampler=sampler, order='embree', rows=({y}, {y + 1}))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PINE_ORACLE_TRACE_PIXEL=f"{x},{y}"))
    qs = [l.split() for l in r.stderr.splitlines() if l.startswith("Q ")]
    if not qs:
        print("no queries traced:", r.stderr[-400:])
        return 1
    rays = np.array([[float.fromhex(v) for v in q[2:10]] for q in qs], np.float32)
    a, b = accelq(ps, rays), mine(ps, rays)
    bad = 0
    for k, q in enumerate(qs):
        same = (a[k, 0] == b[k, 0] and a[k, 3] == b[k, 3] and (a[k, 0] == 0 or ((a[k, 1:3] == b[k, 1:3]).all() and (a[k, 4:] == b[k, 4:]).all())))
        bad += not same
        if not same or len(qs) <= 64:
            print(q[1], " ".join(float(v).hex() for v in rays[k]), "| embree (hit, geometry, tmax bits, any-hit, p, n)", a[k].tolist(), "| restated", b[k].tolist(), "" if same else "  <<<< differs")
    print(f"{len(qs)} queries, {bad} differ")
    return 0


if __name__ == "__main__":
    sys.exit(main())
