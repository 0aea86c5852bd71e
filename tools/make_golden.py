#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref/pine_ref, built from
/root/reference by oracle/Makefile).  Runs only in the build container; the outputs (data: inputs
and the reference's outputs, no reference source) are committed and travel to the GPU box.

    python tools/make_golden.py            # everything except the two 640x640 statistics
    python tools/make_golden.py --full     # also C1/C2 whole-image statistics (~1 min of CPU)
    python tools/make_golden.py --bvh --vertices   # the reference's BVH + traversal orders, per-vertex path terms
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pine_amd as pa  # noqa: E402
from pine_amd import scenes  # noqa: E402  (scene construction only: host code, no GPU)

REF = os.path.join(ROOT, "oracle", "_ref", "pine_ref")
OUT = os.path.join(ROOT, "tests", "golden")


def run_ref(*args):
    return subprocess.run([REF, *map(str, args)], capture_output=True, text=True, check=True).stdout


def ref_film(scene, spp, depth, tmp, sampler=None):
    w, h = scene.camera.film().size
    sp, fp = os.path.join(tmp, "s.pscene"), os.path.join(tmp, "s.film")
    ps = scene.describe()
    open(sp, "w").write(ps)
    extra = [sampler] if sampler else []  # "sobol": SobolSampler(spp) instead of BlueSampler(spp)
    info = json.loads(run_ref("render", sp, spp, depth, fp, *extra).strip().splitlines()[-1])
    return ps, np.fromfile(fp, dtype=np.float32).reshape(h, w, 4), info


FILMS = {
    # name: (builder, spp, depth)
    "cbox_committed_64_s16_d4": (lambda: scenes.cbox((64, 64), "committed"), 16, 4),
    "cbox_readme_64_s16_d4": (lambda: scenes.cbox((64, 64), "readme"), 16, 4),
    "cbox_readme_64_s256_d8": (lambda: scenes.cbox((64, 64), "readme"), 256, 8),
    "cbox_rect_readme_64_s64_d5": (lambda: scenes.cbox((64, 64), "readme", False), 64, 5),
    "cbox_committed_ragged_45x37_s8_d3": (lambda: scenes.cbox((45, 37), "committed"), 8, 3),
    "cbox_readme_64_s1_d1": (lambda: scenes.cbox((64, 64), "readme"), 1, 1),
    "zoo_48_s16_d5": (lambda: scenes.shapes_zoo((48, 48)), 16, 5),
    "classic_cones12_90x45_s32_d6": (lambda: scenes.classic_cones((90, 45), 12), 32, 6),
    "sss_48_s32_d8": (lambda: scenes.sss((48, 48), 1), 32, 8),
    # node-graph materials + Metal / Glossy / Glass (node.h, material.h:39-78)
    "mats_zoo_64_s32_d6": (lambda: scenes.materials_zoo((64, 64)), 32, 6),
    "classic_checker_cones8_90x45_s32_d6": (lambda: scenes.classic_cones((90, 45), 8, checker_floor=True), 32, 6),
    # Point / Spot / Directional lights + Sky environment light (light.h, path.cpp:75-81,104-106)
    "lights_zoo_64_s32_d6": (lambda: scenes.lights_zoo((64, 64)), 32, 6),
    "lights_nosky_48_s16_d4": (lambda: scenes.lights_zoo((48, 48), with_sky=False), 16, 4),
    # triangle meshes without Subsurface (two-level BVH, mesh area light): the stage-queued kernel's mesh path
    # Plane / Line / Cylinder / stand-alone Triangle, incl. emissive Triangle / Line / Plane (geometry.cpp:31-70,171-244,466-595)
    "xshapes_48_s16_d5": (lambda: scenes.xshapes_zoo((48, 48)), 16, 5),
    "xshapes_nolights_40_s8_d3": (lambda: scenes.xshapes_zoo((40, 40), extra_lights=False), 8, 3),
    "mesh_glossy_48_s32_d6": (lambda: scenes.sss((48, 48), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True), 32, 6),
}


SOBOL_FILMS = {
    # SobolSampler(spp) (sampler.h:83-164): odd and even log2(spp), a non-power-of-two spp, more than 256 spp
    "sobol_cbox_readme_48_s8_d4": (lambda: scenes.cbox((48, 48), "readme"), 8, 4),
    "sobol_cbox_ragged_45x37_s12_d3": (lambda: scenes.cbox((45, 37), "committed"), 12, 3),
    "sobol_mats_zoo_32_s16_d6": (lambda: scenes.materials_zoo((32, 32)), 16, 6),
    "sobol_cbox_readme_24_s512_d5": (lambda: scenes.cbox((24, 24), "readme"), 512, 5),
    # with Subsurface: a BSSRDF walk's draws push the dimension counter far beyond what any other path reaches
    "sobol_sss_32_s8_d6": (lambda: scenes.sss((32, 32), 2), 8, 6),
}


HALTON_FILMS = {
    # HaltonSampler(spp) (sampler.h:40-81): restated in the oracle only (the device refuses it)
    "halton_cbox_readme_40_s8_d4": (lambda: scenes.cbox((40, 40), "readme"), 8, 4),
    "halton_mats_zoo_32_s12_d6": (lambda: scenes.materials_zoo((32, 32)), 12, 6),
    # with Subsurface (the dimension wraps to 2 at the 1000-prime table's end), a count that is not a power of two, a ragged film
    "halton_sss_24x20_s12_d5": (lambda: scenes.sss((24, 20), 1, camera="committed"), 12, 5),
}


OTHER_CONFIGS = (
    # the BASELINE configs other than C1 / C2 at full size (whole-image statistics; too big to commit as films)
    # C2 with the README camera [0,1,-4] -> [0,1,0], fov 0.25 (README.md:32), which sees the whole room
    # (the as-committed camera leaves half the film empty): V = 5.06 radiance() invocations per sample instead of 2.64
    ("C2_cbox_640_s256_d8_readme", lambda: scenes.cbox((640, 640), "readme"), 256, 8),
    ("C3_cbox_1920x1080_s1024_d8", lambda: scenes.cbox((1920, 1080), "committed"), 1024, 8),
    ("C4_classic_10k_cones_720x360_s64_d6", lambda: scenes.classic_cones((720, 360), 100), 64, 6),
    ("C5_sss_320_s512_d8", lambda: scenes.sss((320, 320), 3), 512, 8),
    ("C5_sss_640_s512_d8", lambda: scenes.sss((640, 640), 3), 512, 8),  # BASELINE's size: minutes of the reference on 8 threads
)


def other_config_stats(tmp, stats, only=None):
    for name, build, spp, depth in OTHER_CONFIGS:
        if only and name not in only:
            continue
        ps, film, info = ref_film(build(), spp, depth, tmp)
        stats[name] = {"spp": spp, "depth": depth,
                       "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                       "md5": hashlib.md5(film.tobytes()).hexdigest(),
                       "ref_seconds": info["seconds"], "ref_threads": info["threads"],
                       "ref_msamples_per_s": info["msamples_per_s"]}
        print(name, stats[name], flush=True)


def finalize_fixture(tmp):
    """(f)2 film finalize + save: the reference's own Film::finalize / invert_y / to_uint8_array / stb PNG on a
    synthetic HDR film (values above 1, exact zeros, tiny values, a ragged size) and on a rendered golden film, for
    both tone mappers -> tests/golden/finalize.npz (inputs, expected RGBA8, and the reference's PNG decoded)."""
    from PIL import Image
    rng = np.random.default_rng(2025)
    w, h = 37, 23
    hdr = np.zeros((h, w, 4), dtype=np.float32)
    hdr[..., :3] = (10.0 ** rng.uniform(-4, 2.5, (h, w, 3))).astype(np.float32)
    hdr[..., 3] = 1.0
    hdr[0, :5, :3] = 0.0          # exact zeros
    hdr[1, :5, :3] = 600.0        # a visible light source
    hdr[2, :5, :3] = np.float32(1e-30)
    hdr[3, :5, 3] = 7.0           # finalize overwrites w
    rendered = np.load(os.path.join(OUT, "film_cbox_readme_64_s16_d4.npz"))
    rendered = rendered["film"].astype(np.float32).reshape(64, 64, 4)
    out = {}
    for tag, film in (("hdr", hdr), ("cbox", rendered)):
        fh, fw = film.shape[:2]
        fin = os.path.join(tmp, "f.bin")
        film.tofile(fin)
        out[tag + "_film"] = film
        for tm, tmname in ((0, "uncharted2"), (1, "aces")):
            u8p, pngp = os.path.join(tmp, "o.u8"), os.path.join(tmp, "o.png")
            run_ref("finalize", fin, fw, fh, tm, u8p, pngp)
            u8 = np.fromfile(u8p, dtype=np.uint8).reshape(fh, fw, 4)
            png = np.asarray(Image.open(pngp).convert("RGBA"))
            assert np.array_equal(png, u8), "the reference's PNG must decode to its own uint8 array"
            out[f"{tag}_{tmname}_u8"] = u8
    np.savez_compressed(os.path.join(OUT, "finalize.npz"), **out)
    print("finalize.npz", {k: v.shape for k, v in out.items()})


EMBREE_FILMS = {
    # SURVEY.md oracle variant O-gcc-embree (what a `.pine` script gets: program_context.cpp:79-81).  Rendered by
    # oracle/_ref/pine_ref_embree (make -C oracle embree) with PINE_REF_ACCEL=embree.
    "embree_cbox_committed_64_s16_d4": (lambda: scenes.cbox((64, 64), "committed"), 16, 4),
    "embree_cbox_readme_64_s16_d4": (lambda: scenes.cbox((64, 64), "readme"), 16, 4),
    "embree_cbox_readme_64_s256_d8": (lambda: scenes.cbox((64, 64), "readme"), 256, 8),
    "embree_cbox_rect_readme_64_s64_d5": (lambda: scenes.cbox((64, 64), "readme", False), 64, 5),
    # more primitives than one BVH8 node; the other order-dependent shapes (Plane's finite bounds, Line, Cylinder); many primitives
    "embree_clutter20_48_s16_d5": (lambda: scenes.cbox_clutter((48, 48), 12, 19), 16, 5),
    "embree_clutter63_48_s16_d5": (lambda: scenes.cbox_clutter((48, 48), 55, 62), 16, 5),
    "embree_xshapes_48_s16_d5": (lambda: scenes.xshapes_zoo((48, 48)), 16, 5),
    "embree_lights_zoo_48_s16_d6": (lambda: scenes.lights_zoo((48, 48)), 16, 6),
    "embree_classic_cones12_90x45_s16_d6": (lambda: scenes.classic_cones((90, 45), 12), 16, 6),
    # meshes: Embree triangle geometry, Embree's own Moeller-Trumbore test and barycentrics (a Subsurface mesh; a glossy mesh and a mesh area light)
    "embree_sss_48_s32_d8": (lambda: scenes.sss((48, 48), 1), 32, 8),
    "embree_mesh_glossy_48_s32_d6": (lambda: scenes.sss((48, 48), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True), 32, 6),
}


def embree_fixtures(tmp):
    global REF
    exe = os.path.join(ROOT, "oracle", "_ref", "pine_ref_embree")
    if not os.access(exe, os.X_OK):
        raise SystemExit("oracle/_ref/pine_ref_embree missing: make -C oracle embree (about 8 minutes)")
    keep, REF = REF, exe
    os.environ["PINE_REF_ACCEL"] = "embree"
    try:
        for name, (build, spp, depth) in EMBREE_FILMS.items():
            ps, film, info = ref_film(build(), spp, depth, tmp)
            np.savez_compressed(os.path.join(OUT, f"film_{name}.npz"), film=film, pscene=ps, spp=spp, depth=depth)
            print(name, film[..., :3].mean(axis=(0, 1)))
    finally:
        REF = keep
        del os.environ["PINE_REF_ACCEL"]


def bvh_scenes():
    """Scenes of the BVH fixtures (SURVEY.md 8(c) fixture 4): cbox, the 10 000-cone scene of C4, a mesh scene."""
    return {"cbox": scenes.cbox((64, 64), "readme"), "cones10k": scenes.classic_cones((720, 360), 100),
            "sss_mesh": scenes.sss((64, 64), 2, emissive_mesh=True), "zoo": scenes.shapes_zoo((48, 48))}


def bvh_rays(scene, n, seed):
    """n rays for the traversal fixture: camera-like rays (from the camera position into the view cone), rays between
    random points of the scene's extent, and short ones (finite tmax, as shadow rays have)."""
    rng = np.random.default_rng(seed)
    cam = scene.camera
    frm = np.array(cam.frm, dtype=np.float64)
    to = np.array(cam.to, dtype=np.float64)
    rays = np.zeros((n, 8), np.float32)
    fwd = (to - frm) / np.linalg.norm(to - frm)
    k = n // 2
    d = fwd[None, :] + rng.uniform(-0.35, 0.35, (k, 3))
    rays[:k, 0:3] = frm
    rays[:k, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:k, 7] = np.finfo(np.float32).max
    ext = 2.0 if np.linalg.norm(frm) < 6 else 9.0
    a = rng.uniform(-ext, ext, (n - k, 3)) * [1, 0.5, 1] + [0, 1.0, 1.0 if ext < 3 else 0.0]
    b = rng.uniform(-ext, ext, (n - k, 3)) * [1, 0.5, 1] + [0, 0.6, 1.0 if ext < 3 else 0.0]
    dd = b - a
    ln = np.linalg.norm(dd, axis=1)
    rays[k:, 0:3] = a
    rays[k:, 3:6] = dd / ln[:, None]
    rays[k:, 7] = np.where(rng.uniform(size=n - k) < 0.5, ln * 0.999, np.finfo(np.float32).max).astype(np.float32)
    return rays


def bvh_fixtures(tmp):
    """tests/golden/bvh_<scene>.npz from `pine_ref bvh`: the reference's own BVH as a canonical pre-order stream and,
    for 1000 rays, the primitives BVH::intersect / BVH::hit test in order with their results."""
    for i, (name, sc) in enumerate(bvh_scenes().items()):
        sp, rp, tp, vp = (os.path.join(tmp, f) for f in ("s.pscene", "r.bin", "tree.bin", "trav.bin"))
        ps = sc.describe()
        open(sp, "w").write(ps)
        rays = bvh_rays(sc, 1000, 77 + i)
        rays.tofile(rp)
        info = json.loads(run_ref("bvh", sp, rp, tp, vp).strip().splitlines()[-1])
        np.savez_compressed(os.path.join(OUT, f"bvh_{name}.npz"), pscene=ps, rays=rays, tree=np.fromfile(tp, dtype=np.uint32),
                            trav=np.fromfile(vp, dtype=np.uint32))
        print(name, info, os.path.getsize(os.path.join(OUT, f"bvh_{name}.npz")), "bytes")


def vertex_fixture(tmp):
    """tests/golden/vertices_cbox.npz from `pine_ref vertices` (SURVEY.md 8(c) fixture 5): the per-vertex terms of all
    256 paths of an 8x8 film at 4 spp, depth 8, README camera (16 floats per radiance() invocation, ref_driver.cpp)."""
    for name, sc, spp, depth in (("cbox", scenes.cbox((8, 8), "readme"), 4, 8), ("mats", scenes.materials_zoo((8, 8)), 4, 6)):
        sp, vp = os.path.join(tmp, "s.pscene"), os.path.join(tmp, "v.bin")
        ps = sc.describe()
        open(sp, "w").write(ps)
        info = json.loads(run_ref("vertices", sp, spp, depth, vp).strip().splitlines()[-1])
        assert info["restated_loop_equals_render"] is True
        np.savez_compressed(os.path.join(OUT, f"vertices_{name}.npz"), pscene=ps, spp=spp, depth=depth, records=np.fromfile(vp, dtype=np.float32))
        print(name, info)


def gltf_fixture(tmp):
    """SURVEY.md 8(f)4 mesh import: tests/golden/import_test.glb (tools/make_test_glb.py) through the reference's OWN importer
    (`pine_ref gltf`: load_scene -> scene_from_gltf, fileio.cpp:146-330) and PathIntegrator(BVH) -> whole-film statistics."""
    glb = os.path.join(OUT, "import_test.glb")
    out = {}
    for spp, depth in ((4, 5), (16, 6)):
        fp = os.path.join(tmp, "g.film")
        info = json.loads(run_ref("gltf", glb, spp, depth, fp).strip().splitlines()[-1])
        film = np.fromfile(fp, dtype=np.float32).reshape(info["h"], info["w"], 4)
        out[f"s{spp}_d{depth}"] = {"spp": spp, "depth": depth, "size": [info["w"], info["h"]], "geometries": info["geometries"],
                                    "md5": hashlib.md5(film.tobytes()).hexdigest(),
                                    "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                                    "row_320": [float(x) for x in film[320, ::64, :3].reshape(-1)]}
        print("gltf", out[f"s{spp}_d{depth}"]["md5"], out[f"s{spp}_d{depth}"]["mean_rgb"])
    # ... and by the build with EmbreeAccel (oracle/_ref/pine_ref_embree, PINE_REF_ACCEL=embree): what the script's own
    # PathIntegrator(sampler, n) renders on real pine
    exe = os.path.join(ROOT, "oracle", "_ref", "pine_ref_embree")
    if os.access(exe, os.X_OK):
        global REF
        keep, REF = REF, exe
        os.environ["PINE_REF_ACCEL"] = "embree"
        try:
            fp = os.path.join(tmp, "g.film")
            info = json.loads(run_ref("gltf", glb, 4, 5, fp).strip().splitlines()[-1])
            film = np.fromfile(fp, dtype=np.float32).reshape(info["h"], info["w"], 4)
            # The import scene has COPLANAR triangles of different meshes (a box standing on the floor): where a ray meets both at
            # the same t, Embree's own hierarchy -- not restated -- decides which it reports.  The fixture therefore carries, beside
            # the film's md5, the reference's values at the pixels where the restated order (oracle, order "embree") differs: a
            # film equals the reference's everywhere else exactly when it has this md5 after those pixels are overwritten.
            from pine_amd import gltf
            from oracle import oracle
            sc = gltf.load(glb)
            mine, _ = oracle.render(sc.describe(), (info["w"], info["h"]), 4, 5, order="embree")
            ties = np.argwhere((mine.view(np.uint32) != film.view(np.uint32)).any(axis=2))
            if len(ties) > 1000:
                raise SystemExit(f"{len(ties)} pixels differ from the Embree build: more than coplanar ties explain")
            out["embree_s4_d5"] = {"spp": 4, "depth": 5, "size": [info["w"], info["h"]], "md5": hashlib.md5(film.tobytes()).hexdigest(),
                                   "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                                   "tie_pixels": [[int(y), int(x)] + [float(v).hex() for v in film[y, x]] for y, x in ties]}
            print("gltf embree", out["embree_s4_d5"]["md5"], "tie pixels", len(ties))
        finally:
            REF = keep
            del os.environ["PINE_REF_ACCEL"]
    else:
        out["embree_s4_d5"] = json.load(open(os.path.join(OUT, "gltf_import.json"))).get("embree_s4_d5")
    json.dump(out, open(os.path.join(OUT, "gltf_import.json"), "w"), indent=1)


def main():
    full = "--full" in sys.argv
    if "--gltf" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            gltf_fixture(tmp)
        return
    if "--bvh" in sys.argv or "--vertices" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            if "--bvh" in sys.argv:
                bvh_fixtures(tmp)
            if "--vertices" in sys.argv:
                vertex_fixture(tmp)
        return
    if "--embree" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            embree_fixtures(tmp)
        return
    if "--embree-full" in sys.argv:  # tests/golden/stats_640_embree.json: the BASELINE configs at full size by the build with EmbreeAccel
        global REF
        exe = os.path.join(ROOT, "oracle", "_ref", "pine_ref_embree")
        if not os.access(exe, os.X_OK):
            raise SystemExit("oracle/_ref/pine_ref_embree missing: make -C oracle embree")
        REF = exe
        os.environ["PINE_REF_ACCEL"] = "embree"
        out = {}
        with tempfile.TemporaryDirectory() as tmp:
            for name, build, spp, depth in (("C1_cbox_640_s16_d4_committed", lambda: scenes.cbox((640, 640), "committed"), 16, 4),
                                            ("C2_cbox_640_s256_d8_committed", lambda: scenes.cbox((640, 640), "committed"), 256, 8),
                                            ("C2_cbox_640_s256_d8_readme", lambda: scenes.cbox((640, 640), "readme"), 256, 8),
                                            ("C3_cbox_1920x1080_s1024_d8", lambda: scenes.cbox((1920, 1080), "committed"), 1024, 8),
                                            ("C4_classic_10k_cones_720x360_s64_d6", lambda: scenes.classic_cones((720, 360), 100), 64, 6),
                                            ("C5_sss_320_s512_d8", lambda: scenes.sss((320, 320), 3), 512, 8),
                                            ("C5_sss_640_s512_d8", lambda: scenes.sss((640, 640), 3), 512, 8)):
                ps, film, info = ref_film(build(), spp, depth, tmp)
                out[name] = {"spp": spp, "depth": depth, "md5": hashlib.md5(film.tobytes()).hexdigest(),
                             "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                             "ref_seconds": info["seconds"], "ref_threads": info["threads"], "tie_pixels": []}
                if name.startswith("C5"):
                    # A closed mesh has rays through SHARED EDGES: where two triangles report the very same t, Embree's own triangle
                    # hierarchy -- not restated -- decides which one it is (another normal, the same distance).  One query in 10^8; the
                    # fixture carries the reference's values at the pixels where the restated mode (oracle, order "embree") differs,
                    # as tests/golden/gltf_import.json does for its coplanar faces.
                    from oracle import oracle
                    mine, _ = oracle.render(ps, (film.shape[1], film.shape[0]), spp, depth, order="embree")
                    ties = np.argwhere((mine.view(np.uint32) != film.view(np.uint32)).any(axis=2))
                    if len(ties) > 8:
                        raise SystemExit(f"{name}: {len(ties)} pixels differ from the Embree build: more than shared-edge ties explain")
                    out[name]["tie_pixels"] = [[int(y), int(x)] + [float(v).hex() for v in film[y, x]] for y, x in ties]
                print(name, {k: v for k, v in out[name].items() if k != "tie_pixels"}, "tie pixels:", len(out[name]["tie_pixels"]), flush=True)
        json.dump(out, open(os.path.join(OUT, "stats_640_embree.json"), "w"), indent=1)
        return
    if "--finalize-only" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            finalize_fixture(tmp)
        return
    os.makedirs(OUT, exist_ok=True)
    only = next((a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--stats-only=")), None)
    if only:  # just (re)render the named whole-image statistics entries with the real reference
        stats_path = os.path.join(OUT, "stats_640.json")
        stats = json.load(open(stats_path))
        with tempfile.TemporaryDirectory() as tmp:
            other_config_stats(tmp, stats, only)
        json.dump(stats, open(stats_path, "w"), indent=1)
        return
    with tempfile.TemporaryDirectory() as tmp:
        # 1. sampler / rng / host-math known answers
        for spp in (1, 16, 256):
            fp = os.path.join(tmp, "samp.bin")
            run_ref("sampler", spp, fp)
            v = np.fromfile(fp, dtype=np.float32)
            q = np.floor(v * 256).astype(np.uint8)  # values are (k + 0.5) / 256, k < 256: lossless
            assert np.array_equal(((q.astype(np.float32) + np.float32(0.5)) / np.float32(256)).view(np.uint32), v.view(np.uint32))
            np.savez_compressed(os.path.join(OUT, f"sampler_spp{spp}.npz"), k=q)
        fp = os.path.join(tmp, "rng.bin")
        run_ref("rng", fp)
        np.save(os.path.join(OUT, "rng.npy"), np.fromfile(fp, dtype=np.uint64))
        fp = os.path.join(tmp, "host.bin")
        run_ref("host", fp)
        np.save(os.path.join(OUT, "host_math.npy"), np.fromfile(fp, dtype=np.float32))

        # 1b. PRL literal / constant-expression semantics from the reference's own psl::stof / stoi /
        #     to_string and vecmath (pins pine_amd/host/prl.cpp; the JIT itself needs LLVM-18)
        literals = ["0.64", "0.9", "0.185", "1.0", "0.05", "0.1", "0.2", "0.5", "600", "256", "1.3", "0.4", "1.", ".5",
                    "3.14159", "123456.789", "0.001", "100", "2147483647", "0.333333333", "16777217.0", "0.6", "1.9",
                    "0.25", "2.5", "0.3", "0.7", "7", "0", "0.0", "12.375", "99.99999", "0.98", "0.55", "0.02", "160"]
        open(os.path.join(OUT, "prl_semantics.txt"), "w").write(run_ref("prl", *literals))

        # 2. per-shape records on a fixed ray set (grazing / inside / behind / tmax-clipped included)
        zoo = scenes.shapes_zoo((48, 48))
        rng = np.random.default_rng(12345)
        n = 1500
        o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32) + np.array([0, 1, 1], np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        d[:50] = np.array([0, -1, 0], np.float32)           # axis-parallel
        d[50:100] = np.array([1, 0, 0], np.float32)
        o[100:150] = np.array([0.3, 0.3, 0.9], np.float32)  # origins inside the OBB / near the sphere
        tmin = np.zeros((n, 1), np.float32)
        tmax = np.full((n, 1), np.finfo(np.float32).max, np.float32)
        tmax[200:400, 0] = rng.uniform(0.1, 2.5, 200).astype(np.float32)  # clipped rays
        rays = np.concatenate([o, d, tmin, tmax], axis=1).astype(np.float32)
        sp, rp, op = (os.path.join(tmp, x) for x in ("z.pscene", "rays.bin", "shapes.bin"))
        rays.tofile(rp)
        # shapes_zoo: Rect / AABB / OBB / Sphere / Disk / Cone;  shapes_xzoo: Plane / Line / Cylinder / Triangle
        for fname, zscene in (("shapes_zoo.npz", zoo), ("shapes_xzoo.npz", scenes.xshapes_zoo((48, 48)))):
            zps = zscene.describe()
            open(sp, "w").write(zps)
            run_ref("shapes", sp, rp, op)
            rec = np.fromfile(op, dtype=np.float32).reshape(-1, n, 11)
            np.savez_compressed(os.path.join(OUT, fname), rays=rays, records=rec, pscene=np.array(zps))

        # 3. films
        meta = {}
        for name, (build, spp, depth) in FILMS.items():
            ps, film, info = ref_film(build(), spp, depth, tmp)
            np.savez_compressed(os.path.join(OUT, f"film_{name}.npz"), film=film, pscene=np.array(ps),
                                spp=spp, depth=depth)
            meta[name] = {"spp": spp, "depth": depth, "size": [info["w"], info["h"]],
                          "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                          "md5": hashlib.md5(film.tobytes()).hexdigest()}
            print(name, meta[name]["mean_rgb"])

        # 3b. SobolSampler films (oracle-only so far: pins oracle.render(..., sampler="sobol"))
        for name, (build, spp, depth) in SOBOL_FILMS.items():
            ps, film, info = ref_film(build(), spp, depth, tmp, "sobol")
            np.savez_compressed(os.path.join(OUT, f"film_{name}.npz"), film=film, pscene=np.array(ps), spp=spp, depth=depth)
            meta[name] = {"spp": spp, "depth": depth, "size": [info["w"], info["h"]], "sampler": "sobol",
                          "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                          "md5": hashlib.md5(film.tobytes()).hexdigest()}
            print(name, meta[name]["mean_rgb"])

        for name, (build, spp, depth) in HALTON_FILMS.items():
            ps, film, info = ref_film(build(), spp, depth, tmp, "halton")
            np.savez_compressed(os.path.join(OUT, f"film_{name}.npz"), film=film, pscene=np.array(ps), spp=spp, depth=depth)
            meta[name] = {"spp": spp, "depth": depth, "size": [info["w"], info["h"]], "sampler": "halton",
                          "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                          "md5": hashlib.md5(film.tobytes()).hexdigest()}
            print(name, meta[name]["mean_rgb"])

        # 4. whole-image statistics of the BASELINE configs (too big to commit as films)
        stats_path = os.path.join(OUT, "stats_640.json")
        stats = json.load(open(stats_path)) if os.path.exists(stats_path) else {}
        if full:
            for name, spp, depth, cam in (("C1_cbox_640_s16_d4_committed", 16, 4, "committed"),
                                           ("C2_cbox_640_s256_d8_committed", 256, 8, "committed"),
                                           ("cbox_640_s16_d4_readme", 16, 4, "readme")):
                ps, film, info = ref_film(scenes.cbox((640, 640), cam), spp, depth, tmp)
                stats[name] = {"spp": spp, "depth": depth, "camera": cam,
                               "mean_rgb": [float(x) for x in film[..., :3].mean(axis=(0, 1), dtype=np.float64)],
                               "md5": hashlib.md5(film.tobytes()).hexdigest(),
                               "center_pixel": [float(x) for x in film[320, 320, :3]],
                               "black_pixels": int((film[..., :3] == 0).all(axis=2).sum()),
                               "ref_seconds": info["seconds"], "ref_threads": info["threads"],
                               "ref_msamples_per_s": info["msamples_per_s"]}
                print(name, stats[name])
            other_config_stats(tmp, stats)
            json.dump(stats, open(stats_path, "w"), indent=1)
        json.dump(meta, open(os.path.join(OUT, "films.json"), "w"), indent=1)
        finalize_fixture(tmp)


if __name__ == "__main__":
    main()
