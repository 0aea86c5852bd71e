"""CPU: the oracle (oracle/pine_oracle.cpp) against the golden vectors produced by the REAL reference
(tests/golden, tools/make_golden.py).  Everything is bit-exact: integer streams, host math, per-shape
records and whole films."""
import json
import os

import numpy as np
import pytest

from conftest import (GOLDEN, FILM_NAMES, SOBOL_FILM_NAMES, HALTON_FILM_NAMES, EMBREE_FILM_NAMES, EMBREE_MORE_FILM_NAMES, EMBREE_EXPECTED, assert_bit_equal,
                      film_distance, load_film)


@pytest.mark.parametrize("spp", [1, 16, 256])
def test_sampler_stream_matches_reference(oracle, spp):
    k = np.load(os.path.join(GOLDEN, f"sampler_spp{spp}.npz"))["k"]
    ref = (k.astype(np.float32) + np.float32(0.5)) / np.float32(256)
    assert_bit_equal(oracle.sampler_stream(spp), ref, f"BlueSobolSampler({spp}) stream")


def test_sampler_known_answers(oracle):
    # BlueSampler(16), pixel (3,5), sample 0, dims 0..5 and sample 1, dims 0..2
    s = oracle.sampler_stream(16).reshape(6, -1)[2]  # pixel (3,5) is the third test pixel
    np.testing.assert_array_equal(s[:6], np.float32([0.564453125, 0.935546875, 0.431640625, 0.025390625, 0.880859375, 0.021484375]))
    np.testing.assert_array_equal(s[260:263], np.float32([0.060546875, 0.162109375, 0.845703125]))
    s = oracle.sampler_stream(256).reshape(6, -1)[5]  # pixel (639,639)
    np.testing.assert_array_equal(s[:3], np.float32([0.568359375, 0.693359375, 0.728515625]))


def test_sampler_spp_rounding():
    from oracle import oracle as o
    # BlueSobolSampler ctor: round up to a power of two, clamp to 256 (sampler.cpp:115-121)
    assert [o.effective_spp(n) for n in (1, 2, 3, 5, 16, 17, 100, 256, 512, 1024)] == [1, 2, 4, 8, 16, 32, 128, 256, 256, 256]


def test_rng_stream_matches_reference(oracle):
    ref = np.load(os.path.join(GOLDEN, "rng.npy"))
    got = oracle.rng_stream()
    assert np.array_equal(got, ref)
    # SURVEY.md Appendix D known answers
    r = got.reshape(6, 19)
    assert int(r[0, 0]) == 0x6C9A7A0560404F9B            # hash((0,0), 0)
    assert int(r[2, 0]) == 0x56B273D97D8B5B00            # hash((3,5), 0)
    assert int(r[2, 1]) == 0x387FD842DB87B11A and int(r[2, 2]) == 0xDC2BCC5E60C5924B
    f = r[2, 3:7].astype(np.uint32).view(np.float32)
    np.testing.assert_array_equal(f, np.float32([0.159773335, 0.0878059268, 0.203855723, 0.123972036]))


def test_host_math_matches_reference(oracle):
    assert_bit_equal(oracle.host_math(), np.load(os.path.join(GOLDEN, "host_math.npy")), "host math")


@pytest.mark.parametrize("which", ["shapes_zoo", "shapes_xzoo"])
def test_shape_records_match_reference(oracle, which):
    z = np.load(os.path.join(GOLDEN, which + ".npz"))
    rec = z["records"]
    got = oracle.shapes(str(z["pscene"]), z["rays"], rec.shape[0])
    # compute_surface_info output is only defined on a hit (the reference leaves `it` untouched otherwise)
    assert_bit_equal(got[..., :3], rec[..., :3], "hit / intersect / tmax")
    hit = rec[..., 1] == 1
    assert hit.sum() > 500
    assert_bit_equal(got[hit][:, 3:], rec[hit][:, 3:], "surface info on hits")


@pytest.mark.parametrize("name", FILM_NAMES)
def test_film_bit_identical_to_reference(oracle, name):
    ref, ps, spp, depth = load_film(name)
    h, w, _ = ref.shape
    film, st = oracle.render(ps, (w, h), spp, depth)
    assert_bit_equal(film, ref, name)
    assert st.camera_samples == w * h * oracle.effective_spp(spp)


@pytest.mark.parametrize("name", SOBOL_FILM_NAMES)
def test_sobol_sampler_film_bit_identical_to_reference(oracle, name):
    """SobolSampler (sampler.h:83-164, sampler.cpp:81-113): Morton-indexed, base-4 digit permutations, fast Owen
    scrambling -- the oracle's restatement against films the real reference rendered with SobolSampler(spp)."""
    ref, ps, spp, depth = load_film(name)
    h, w, _ = ref.shape
    film, st = oracle.render(ps, (w, h), spp, depth, sampler="sobol")
    assert_bit_equal(film, ref, name)
    assert st.camera_samples == w * h * spp  # no rounding, no clamp


@pytest.mark.parametrize("name", HALTON_FILM_NAMES)
def test_halton_sampler_film_bit_identical_to_reference(oracle, name):
    """HaltonSampler (sampler.h:40-81, sampler.cpp:16-79): scrambled radical inverses over the first 1000 primes, the
    digit permutations shuffled by a default-seeded RNG, pixel offsets through the 128 x 243 grid -- the oracle's
    restatement (tables derived, not stored) against films the real reference rendered with HaltonSampler(spp).
    (The device path is held to the same two films and to this restatement: tests/test_gpu_parity.py.)"""
    ref, ps, spp, depth = load_film(name)
    h, w, _ = ref.shape
    film, st = oracle.render(ps, (w, h), spp, depth, sampler="halton")
    assert_bit_equal(film, ref, name)
    assert st.camera_samples == w * h * spp


def test_row_range_and_shards_compose(oracle):
    ref, ps, spp, depth = load_film("cbox_committed_ragged_45x37_s8_d3")
    h, w, _ = ref.shape
    a, _ = oracle.render(ps, (w, h), spp, depth, rows=(0, 20))
    b, _ = oracle.render(ps, (w, h), spp, depth, rows=(20, h))
    assert_bit_equal(a + b, ref, "row ranges")
    tot = sum(oracle.render_shard(ps, (w, h), spp, depth, r, 3) for r in range(3))
    assert_bit_equal(tot, ref, "3 tile shards")


def test_depth_must_be_positive(oracle):
    _, ps, spp, _ = load_film("cbox_readme_64_s1_d1")
    with pytest.raises(RuntimeError, match="max_path_length"):
        oracle.render(ps, (64, 64), spp, 0)


@pytest.mark.skipif(not os.path.exists("/root/reference/src/pine"), reason="reference sources absent")
def test_oracle_against_live_reference(oracle):
    """In the build container the real reference binary is available: one fresh config, bit-exact."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/pine_ref not built")
    from pine_amd import scenes
    sc = scenes.cbox((40, 24), "readme")
    ps = sc.describe()
    ref, _ = oracle.ref_render(ps, (40, 24), 32, 6)
    got, _ = oracle.render(ps, (40, 24), 32, 6)
    assert_bit_equal(got, ref, "live reference")


def test_stats_file_present():
    st = json.load(open(os.path.join(GOLDEN, "stats_640.json")))
    c1 = st["C1_cbox_640_s16_d4_committed"]
    # SURVEY.md Appendix C sanity values of the g++ oracle
    np.testing.assert_allclose(c1["mean_rgb"], [0.093782, 0.058198, 0.017012], atol=1e-6)
    np.testing.assert_allclose(c1["center_pixel"], [0.087721169, 0.0352332816, 0.00938287377], rtol=1e-7)


def test_random_scenes_oracle_equals_the_reference_binary(oracle):
    """Fuzzing: seeded random mixes of every shape / material / light kind (pine_amd.scenes.random_scene); the
    restatement against oracle/_ref/pine_ref, bit for bit.  (tools/fuzz_scenes.py ran 300 seeds both ways.)"""
    import subprocess
    from pine_amd import scenes
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "pine_ref")
    if not os.access(exe, os.X_OK):
        pytest.skip("oracle/_ref/pine_ref not built (no /root/reference here)")
    import tempfile
    for seed, variety in [(k, 1) for k in range(2000, 2012)] + [(k, 2) for k in range(2100, 2106)]:
        # variety 1: + odd film sizes, thin lens, SobolSampler; 2: + fractional Uber lobes, Subsurface meshes (in-path RNG)
        sc, spp, depth, sampler = scenes.random_scene(seed, variety=variety)
        ps = sc.describe()
        w, h = sc.camera.film().size
        mine, _ = oracle.render(ps, (w, h), spp, depth, sampler=sampler)
        with tempfile.TemporaryDirectory() as tmp:
            sp, fp = os.path.join(tmp, "s.pscene"), os.path.join(tmp, "s.film")
            open(sp, "w").write(ps)
            subprocess.run([exe, "render", sp, str(spp), str(depth), fp] + (["sobol"] if sampler == "sobol" else []),
                           check=True, capture_output=True, timeout=120)
            ref = np.fromfile(fp, dtype=np.float32).reshape(h, w, 4)
        assert_bit_equal(mine, ref, f"random scene {seed}")


@pytest.mark.parametrize("name", EMBREE_FILM_NAMES)
def test_distance_to_the_embree_oracle(oracle, name):
    """the second oracle, O-gcc-embree -- pine built with g++ and its EmbreeAccel, which is what a
    `.pine` script gets (program_context.cpp:79-81).  This build reproduces pine's own BVH order (O-gcc-bvh) bit for
    bit; against Embree's order the result is bit-identical on scenes without order-dependent shapes (Rect-only
    cbox) and measurably different where a scaled Box(AABB, mat4) is clipped by the world tmax (bbox.cpp:144-172,
    SURVEY.md Appendix A3).  The numbers pinned here are the reference's own BVH-vs-Embree distance."""
    emb, ps, spp, depth = load_film(name)
    h, w = emb.shape[:2]
    film, _ = oracle.render(ps, (w, h), spp, depth)
    bvh_name = name.replace("embree_", "")
    ref_bvh, ps2, _, _ = load_film(bvh_name)
    assert ps2 == ps  # the same scene description went to both reference builds
    assert_bit_equal(film, ref_bvh, f"{bvh_name}: restatement vs the reference with pine's BVH")
    d = film_distance(film, emb)
    lo, hi = EMBREE_EXPECTED[name]["identical"]
    assert lo <= d["identical"] <= hi, d
    lo, hi = EMBREE_EXPECTED[name]["rmse"]
    assert lo <= d["rmse"] <= hi, d
    if "rect" in name:
        assert_bit_equal(film, emb, "Rect-only cbox: both accels, both oracles, one film")


@pytest.mark.parametrize("name", EMBREE_FILM_NAMES + EMBREE_MORE_FILM_NAMES)
def test_embree_order_reproduces_the_embree_oracle(oracle, name):
    """SURVEY.md Appendix A3's second traversal order, taken to its end: the order in which the reference's default accel hands the
    shapes to their tests, restated from the vendored Embree's BVH8 builder and single-ray traverser (oracle order mode "embree").
    With it the CPU restatement renders the films of the REAL reference built with EmbreeAccel bit for bit -- scaled boxes, planes,
    lines and cylinders, 8 to 155 primitives; this is the checker of PINE_GPU_FLAG_ORDER_EMBREE."""
    emb, ps, spp, depth = load_film(name)
    film, _ = oracle.render(ps, (emb.shape[1], emb.shape[0]), spp, depth, order="embree")
    assert_bit_equal(film, emb, f"EmbreeAccel's order vs O-gcc-embree, {name}")


@pytest.mark.parametrize("name", EMBREE_FILM_NAMES)
def test_nearest_bounds_first_is_that_order_while_one_node_holds_the_scene(oracle, name):
    """The plain nearest-bounds-first order (oracle order mode "nearest": what the product's order mode was before the restatement)
    coincides with Embree's while the scene fits ONE BVH8 node -- cbox -- and no longer once it does not
    (profiles/r04_embree_order_distance.txt): kept as the measured reason for restating the hierarchy."""
    emb, ps, spp, depth = load_film(name)
    film, _ = oracle.render(ps, (emb.shape[1], emb.shape[0]), spp, depth, order="nearest")
    assert_bit_equal(film, emb, f"nearest-bounds-first order vs O-gcc-embree, {name}")
    big, ps, spp, depth = load_film("embree_clutter63_48_s16_d5")
    film, _ = oracle.render(ps, (48, 48), spp, depth, order="nearest")
    assert (film.view(np.uint32) != big.view(np.uint32)).any()
