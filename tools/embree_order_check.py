#!/usr/bin/env python3
"""usage (this container: needs oracle/_ref/embree_probe, see oracle/embree_probe.cpp; `make -C oracle embree probe`):
  tools/embree_order_check.py [scenes] [rays]     compare, print the count of differing rays
  tools/embree_order_check.py --fixture           (re)write tests/golden/embree_order.npz: 14 box sets x 48 rays with REAL Embree's answers
  tools/embree_order_check.py --triangles [n]     Embree's triangle test (oracle/embree_tri_probe.cpp) against the restated one on n rays
                                                  per mesh, and (re)write tests/golden/embree_triangles.npz (600 rays of it)
The order in which REAL Embree (the vendored 4.3.1, built here) calls the user callback against the order the restatement
(oracle order mode "embree": builder + traverser) produces, ray by ray: random box sets of 1 .. 400 primitives, rays with and
without reported hits.  Prints the number of rays whose call sequence, hit primitive or final tfar differ."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "embree_probe")


def hexf(a):
    return " ".join(float(x).hex() for x in a)


def case(rng, n, m):
    kind = rng.integers(0, 3)
    if kind == 0:    # scattered small boxes
        c = rng.uniform(-1, 1, (n, 3)); e = rng.uniform(0.01, 0.3, (n, 3))
    elif kind == 1:  # a grid (C4-like), flat boxes among them
        g = int(np.ceil(np.sqrt(n)))
        c = np.array([[-1 + 2 * (i % g) / g, 0, -1 + 2 * (i // g) / g] for i in range(n)]) + rng.uniform(-0.01, 0.01, (n, 3))
        e = np.full((n, 3), 0.4 / g); e[::5, 1] = 0.0
    else:            # cbox-like: big flat walls plus boxes
        c = rng.uniform(-1, 1, (n, 3)); e = rng.uniform(0.0, 1.0, (n, 3)) * (rng.uniform(size=(n, 3)) < 0.7)
    boxes = np.concatenate([c - e, c + e], axis=1).astype(np.float32)
    o = rng.uniform(-2, 2, (m, 3)); t = rng.uniform(-1, 1, (m, 3)); d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if m > 4:
        d[:m // 8, rng.integers(0, 3)] = 0.0  # axis-parallel rays (rcp_safe's clamp)
        d[:m // 8] /= np.maximum(np.linalg.norm(d[:m // 8], axis=1, keepdims=True), 1e-9)
    rays = np.concatenate([o, d, np.zeros((m, 1)), np.full((m, 1), 3.4028235e38)], axis=1).astype(np.float32)
    rays[::3, 7] = rng.uniform(0.5, 4.0, len(rays[::3]))
    hits = np.where(rng.uniform(size=(m, n)) < 0.5, rng.uniform(0.2, 4.0, (m, n)), -1.0).astype(np.float32)
    hits[::4] = -1.0
    return boxes, rays, hits


def probe(boxes, rays, hits):
    n, m = len(boxes), len(rays)
    text = f"{n}\n" + "\n".join(hexf(b) for b in boxes) + f"\n{m}\n" + "\n".join(hexf(r) + " " + hexf(h) for r, h in zip(rays, hits)) + "\n"
    out = subprocess.run([PROBE], input=text, capture_output=True, text=True, check=True).stdout.splitlines()
    seqs, res = [], []
    for k in range(m):
        seq, tail = out[k].split("|")
        wh, wt = tail.split()
        seqs.append([int(x) for x in seq.split()])
        res.append((int(wh), np.float32(float.fromhex(wt))))
    return seqs, res


def fixture():
    rng = np.random.default_rng(11)
    data = {}
    for c, n in enumerate([1, 2, 3, 5, 8, 9, 10, 12, 16, 20, 33, 64, 100, 400]):
        boxes, rays, hits = case(rng, n, 48)
        seqs, res = probe(boxes, rays, hits)
        flat = np.full((48, n + 1), -1, np.int32)
        for k, sq in enumerate(seqs):
            flat[k, 0] = len(sq)
            flat[k, 1:1 + len(sq)] = sq
        data[f"boxes{c}"], data[f"rays{c}"], data[f"hits{c}"], data[f"calls{c}"] = boxes, rays, hits, flat
        data[f"hit{c}"] = np.array([r[0] for r in res], np.int32)
        data[f"tfar{c}"] = np.array([r[1] for r in res], np.float32)
    path = os.path.join(ROOT, "tests", "golden", "embree_order.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes")


def tri_probe(verts, idx, rays):
    text = f"{len(verts)}\n" + "\n".join(hexf(v) for v in verts) + f"\n{len(idx)}\n" + "\n".join(" ".join(str(int(i)) for i in t) for t in idx) + \
           f"\n{len(rays)}\n" + "\n".join(hexf(r) for r in rays) + "\n"
    out = subprocess.run([PROBE.replace("embree_probe", "embree_tri_probe")], input=text, capture_output=True, text=True, check=True).stdout.splitlines()
    return np.array([[float(int(l.split()[0]))] + [float.fromhex(x) for x in l.split()[1:7]] for l in out], np.float32)


def tri_mine(verts, idx, rays):
    lib = oracle.lib()
    lib.oracle_embree_triangles.restype = C.c_int
    out = np.zeros((len(rays), 7), np.float32)
    idx32 = np.ascontiguousarray(idx, np.uint32)
    lib.oracle_embree_triangles(np.ascontiguousarray(verts, np.float32).ctypes.data_as(C.c_void_p), idx32.ctypes.data_as(C.c_void_p), len(idx),
                                np.ascontiguousarray(rays, np.float32).ctypes.data_as(C.c_void_p), C.c_int64(len(rays)), out.ctypes.data_as(C.c_void_p))
    return out


def triangles():
    from pine_amd import scenes
    n = int(sys.argv[sys.argv.index("--triangles") + 1]) if len(sys.argv) > sys.argv.index("--triangles") + 1 else 20000
    rng = np.random.default_rng(21)
    data, bad = {}, 0
    for c in range(3):
        if c == 0:    # an icosphere: shared edges and vertices
            v, f = scenes.icosphere(2, 0.8, (0.0, 0.0, 0.0))
            verts, idx = np.asarray(v, np.float32).reshape(-1, 3), np.asarray(f).reshape(-1, 3)
        elif c == 1:  # a soup of random triangles, slivers among them
            verts = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
            verts[::7] *= np.float32(1e-3)
            idx = rng.integers(0, 300, (200, 3))
            idx = idx[(idx[:, 0] != idx[:, 1]) & (idx[:, 1] != idx[:, 2]) & (idx[:, 0] != idx[:, 2])]
        else:         # large and far: big coordinates, tiny denominators
            verts = (rng.uniform(-1, 1, (120, 3)) * [100, 1e-2, 100] + [0, 5, 0]).astype(np.float32)
            idx = rng.integers(0, 120, (80, 3))
            idx = idx[(idx[:, 0] != idx[:, 1]) & (idx[:, 1] != idx[:, 2]) & (idx[:, 0] != idx[:, 2])]
        o = rng.uniform(-2, 2, (n, 3)) * ([1, 1, 1] if c < 2 else [50, 1, 50]) + ([0, 0, 0] if c < 2 else [0, 8, 0])
        t = rng.uniform(-1, 1, (n, 3)) * ([0.8, 0.8, 0.8] if c < 2 else [100, 0.01, 100]) + ([0, 0, 0] if c < 2 else [0, 5, 0])
        d = t - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, d, np.zeros((n, 1)), np.where(rng.uniform(size=(n, 1)) < 0.3, rng.uniform(0.5, 3.0, (n, 1)), 3.4028235e38)], 1).astype(np.float32)
        a, b = tri_probe(verts, idx, rays), tri_mine(verts, idx, rays)
        hit = a[:, 0] >= 0
        diff = (a[:, 0] != b[:, 0]) | (a[:, 1].view(np.uint32) != b[:, 1].view(np.uint32)) | (hit & (a[:, 2:].view(np.uint32) != b[:, 2:].view(np.uint32)).any(axis=1))
        print(f"mesh {c}: {len(idx)} triangles, {n} rays, {int(hit.sum())} hits, {int(diff.sum())} differ", flush=True)
        for k in np.argwhere(diff)[:3, 0]:
            print("  ", rays[k].tolist(), a[k].tolist(), b[k].tolist())
        bad += int(diff.sum())
        keep = np.concatenate([np.argwhere(hit)[:150, 0], np.argwhere(~hit)[:50, 0]])
        data[f"verts{c}"], data[f"idx{c}"], data[f"rays{c}"], data[f"want{c}"] = verts, idx.astype(np.uint32), rays[keep], a[keep]
    path = os.path.join(ROOT, "tests", "golden", "embree_triangles.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes;", bad, "rays differ")
    return 1 if bad else 0


def main():
    if "--fixture" in sys.argv:
        return fixture()
    if "--triangles" in sys.argv:
        return triangles()
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(5)
    lib = oracle.lib()
    lib.oracle_embree_order.restype = C.c_int
    bad = total = 0
    for s in range(scenes):
        n = int(rng.choice([1, 2, 3, 5, 8, 9, 10, 12, 16, 20, 33, 64, 100, 400]))
        boxes, rays, hits = case(rng, n, m)
        text = f"{n}\n" + "\n".join(hexf(b) for b in boxes) + f"\n{m}\n" + "\n".join(hexf(r) + " " + hexf(h) for r, h in zip(rays, hits)) + "\n"
        out = subprocess.run([PROBE], input=text, capture_output=True, text=True, check=True).stdout.splitlines()
        sbad = 0
        for k in range(m):
            seq, tail = out[k].split("|")
            want = [int(x) for x in seq.split()]
            wh, wt = tail.split()
            ids = (C.c_int * (n + 8))()
            hid = C.c_int(0)
            tf = C.c_float(0)
            cnt = lib.oracle_embree_order(boxes.ctypes.data_as(C.c_void_p), n, rays[k].ctypes.data_as(C.c_void_p), hits[k].ctypes.data_as(C.c_void_p), ids, n + 8, C.byref(hid), C.byref(tf))
            got = list(ids[:min(cnt, n + 8)])
            if got != want or hid.value != int(wh) or float(tf.value).hex() != float.fromhex(wt).hex():
                if sbad == 0 and bad < 5:
                    print(f"scene {s} n={n} ray {k}: embree {want} | {wh} {wt}   restated {got} | {hid.value} {float(tf.value).hex()}")
                sbad += 1
        bad += sbad
        total += m
        print(f"scene {s}: {n} primitives, {sbad} of {m} rays differ", flush=True)
    print(f"{bad} of {total} rays differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
