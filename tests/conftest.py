import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = bits(a) != bits(b)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values differ; first at {np.argwhere(bad)[0].tolist()}: {a[tuple(np.argwhere(bad)[0])]} vs {b[tuple(np.argwhere(bad)[0])]}"


def film_distance(a, b):
    """SURVEY.md 8(d)'s film metrics of `a` against the oracle film `b` (radiance clamped to 8, as the per-level
    clamp bounds every non-emissive contribution): share of bit-identical pixels, share of pixels with per-pixel
    relative L2 <= 1e-4, whole-image RMSE."""
    a3 = np.minimum(np.asarray(a, dtype=np.float64)[..., :3], 8.0)
    b3 = np.minimum(np.asarray(b, dtype=np.float64)[..., :3], 8.0)
    identical = float((bits(a)[..., :3] == bits(b)[..., :3]).all(axis=-1).mean())
    rel = np.linalg.norm(a3 - b3, axis=-1) / (np.linalg.norm(b3, axis=-1) + 1e-3)
    return {"identical": identical, "within_1e-4": float((rel <= 1e-4).mean()), "rmse": float(np.sqrt(((a3 - b3) ** 2).mean()))}


EMBREE_FILM_NAMES = ["embree_cbox_committed_64_s16_d4", "embree_cbox_readme_64_s16_d4", "embree_cbox_readme_64_s256_d8",
                     "embree_cbox_rect_readme_64_s64_d5"]
# ... and films of the same build on scenes with more primitives than one BVH8 node, with the other order-dependent shapes
# (Plane's finite bounds, Line, Cylinder), with many primitives: what PINE_GPU_FLAG_ORDER_EMBREE must render bit for bit
EMBREE_MORE_FILM_NAMES = ["embree_clutter20_48_s16_d5", "embree_clutter63_48_s16_d5", "embree_xshapes_48_s16_d5",
                          "embree_lights_zoo_48_s16_d6", "embree_classic_cones12_90x45_s16_d6",
                          "embree_sss_48_s32_d8", "embree_mesh_glossy_48_s32_d6"]  # (meshes: Embree's own triangle test)


def embree_scene(name):
    """The scene a tests/golden/film_embree_* fixture was rendered from (tools/make_golden.py EMBREE_FILMS)."""
    import pine_amd as pa
    from pine_amd import scenes
    return {"embree_sss_48_s32_d8": lambda: scenes.sss((48, 48), 1),
            "embree_mesh_glossy_48_s32_d6": lambda: scenes.sss((48, 48), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True),
            "embree_cbox_readme_64_s16_d4": lambda: scenes.cbox((64, 64), "readme"),
            "embree_cbox_committed_64_s16_d4": lambda: scenes.cbox((64, 64), "committed"),
            "embree_cbox_readme_64_s256_d8": lambda: scenes.cbox((64, 64), "readme"),
            "embree_cbox_rect_readme_64_s64_d5": lambda: scenes.cbox((64, 64), "readme", False),
            "embree_clutter20_48_s16_d5": lambda: scenes.cbox_clutter((48, 48), 12, 19),
            "embree_clutter63_48_s16_d5": lambda: scenes.cbox_clutter((48, 48), 55, 62),
            "embree_xshapes_48_s16_d5": lambda: scenes.xshapes_zoo((48, 48)),
            "embree_lights_zoo_48_s16_d6": lambda: scenes.lights_zoo((48, 48)),
            "embree_classic_cones12_90x45_s16_d6": lambda: scenes.classic_cones((90, 45), 12)}[name]()
# What separates pine's own BVH order (reproduced here bit for bit) from EmbreeAccel (the `.pine` default) on each of
# those films: measured once with the two real reference builds (tools/make_golden.py --embree).  The
# Rect-only scene has no order-dependent shape: the two accels agree to the last bit.
EMBREE_EXPECTED = {
    "embree_cbox_committed_64_s16_d4": {"identical": (0.80, 0.86), "rmse": (0.02, 0.2)},
    "embree_cbox_readme_64_s16_d4": {"identical": (0.35, 0.42), "rmse": (0.05, 0.3)},
    "embree_cbox_readme_64_s256_d8": {"identical": (0.0, 0.01), "rmse": (0.05, 0.3)},  # (at 256 spp every pixel has a path that meets a box)
    "embree_cbox_rect_readme_64_s64_d5": {"identical": (1.0, 1.0), "rmse": (0.0, 0.0)},
}


def load_film(name):
    z = np.load(os.path.join(GOLDEN, f"film_{name}.npz"))
    return z["film"], str(z["pscene"]), int(z["spp"]), int(z["depth"])


FILM_NAMES = [
    "cbox_committed_64_s16_d4", "cbox_readme_64_s16_d4", "cbox_readme_64_s256_d8",
    "cbox_rect_readme_64_s64_d5", "cbox_committed_ragged_45x37_s8_d3", "cbox_readme_64_s1_d1",
    "zoo_48_s16_d5", "classic_cones12_90x45_s32_d6", "sss_48_s32_d8",
    "mats_zoo_64_s32_d6", "classic_checker_cones8_90x45_s32_d6",   # node-graph materials, Metal / Glossy / Glass
    "lights_zoo_64_s32_d6", "lights_nosky_48_s16_d4",               # delta lights, Sky environment light
    "mesh_glossy_48_s32_d6",                                        # meshes (incl. a mesh area light) without Subsurface
    "xshapes_48_s16_d5", "xshapes_nolights_40_s8_d3",               # Plane / Line / Cylinder / Triangle
]


SOBOL_FILM_NAMES = ["sobol_cbox_readme_48_s8_d4", "sobol_cbox_ragged_45x37_s12_d3", "sobol_mats_zoo_32_s16_d6",
                    "sobol_cbox_readme_24_s512_d5", "sobol_sss_32_s8_d6"]   # rendered by the reference with SobolSampler(spp)


HALTON_FILM_NAMES = ["halton_cbox_readme_40_s8_d4", "halton_mats_zoo_32_s12_d6", "halton_sss_24x20_s12_d5"]  # rendered by the reference with HaltonSampler(spp)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
