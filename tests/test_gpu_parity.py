"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the reference's golden
vectors.  Integer streams AND films are compared bit for bit: the device arithmetic is IEEE binary32
in the reference's operand order with a libm-exact sin/cos (pine_amd/csrc/pine_libm.h), so the stated
tolerance is ZERO for every scene: powf / logf (Uber Schlick term, BSSRDF free flight) and atan2f / acosf (a
sphere's uv) are glibc's algorithms restated too, so no device-libm call is left on the path."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import SOBOL_FILM_NAMES, HALTON_FILM_NAMES, GOLDEN, FILM_NAMES, assert_bit_equal, load_film

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["queue", "mega"])
def path_kernel(request, monkeypatch):
    """Every test runs against both path kernels: the stage-queued one (the default wherever it
    applies) and the lane-owns-a-path megakernel (PINE_GPU_KERNEL=mega; the only one for meshes and
    Subsurface).  Both must reproduce the reference bit for bit."""
    if request.param == "mega":
        monkeypatch.setenv("PINE_GPU_KERNEL", "mega")
    else:
        monkeypatch.delenv("PINE_GPU_KERNEL", raising=False)
    return request.param


def _loaded_native():
    """Fail loudly if the HIP extension is not the thing that runs."""
    from pine_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libpine_gpu" in maps, "libpine_gpu.so is not mapped into this process"


def test_native_library_loaded():
    _loaded_native()


def test_device_sincos_equals_host_libm():
    from pine_amd import _lib
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-7, 7, 200000), rng.uniform(-0.8, 0.8, 100000), rng.uniform(-119, 119, 50000),
                        np.float32([0.0, -0.0, 1e-5, 0.785398163, 1.57079637, 3.14159274, 6.28318548])]).astype(np.float32)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    _lib.check(_lib.lib.pine_gpu_test_sincos(0, x.ctypes.data_as(_lib.c_f_p), x.size, s.ctypes.data_as(_lib.c_f_p),
                                             c.ctypes.data_as(_lib.c_f_p)))
    hs = np.float32([libm.sinf(float(v)) for v in x])
    hc = np.float32([libm.cosf(float(v)) for v in x])
    assert_bit_equal(s, hs, "device sinf vs libm")
    assert_bit_equal(c, hc, "device cosf vs libm")


def test_device_powf_logf_equal_host_libm():
    """FrSchlick's pow, the node '^' and the BSSRDF free flight's log on the device == glibc's, bit for bit."""
    from pine_amd import _lib
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = libm.logf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    libm.logf.argtypes = [C.c_float]
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0, 1, 150000), rng.uniform(0, 50, 50000), 10.0 ** rng.uniform(-30, 30, 50000),
                        np.float32([0.0, 1.0, 0.5, 2.0, 1e-40, 3.0])]).astype(np.float32)
    y = np.concatenate([np.full(100000, 5.0), rng.uniform(-8, 8, 150000), np.float32([0.0, 5.0, 5.0, 0.5, 2.0, -1.5])]).astype(np.float32)
    p = np.zeros_like(x)
    lg = np.zeros_like(x)
    _lib.check(_lib.lib.pine_gpu_test_powlog(0, x.ctypes.data_as(_lib.c_f_p), y.ctypes.data_as(_lib.c_f_p), x.size,
                                             p.ctypes.data_as(_lib.c_f_p), lg.ctypes.data_as(_lib.c_f_p)))
    hp = np.float32([libm.powf(float(a), float(b)) for a, b in zip(x, y)])
    hl = np.float32([libm.logf(float(a)) for a in x])
    assert_bit_equal(p, hp, "device powf vs libm")
    assert_bit_equal(lg, hl, "device logf vs libm")


def test_device_atan2f_acosf_equal_host_libm():
    """A Sphere's uv (cartesian_to_spherical, read by node graphs through UV()) on the device == glibc's atan2f / acosf,
    bit for bit: components of unit vectors, the quadrant / axis special cases, huge and tiny ratios."""
    from pine_amd import _lib
    libm = C.CDLL("libm.so.6")
    libm.atan2f.restype = libm.acosf.restype = C.c_float
    libm.atan2f.argtypes = [C.c_float, C.c_float]
    libm.acosf.argtypes = [C.c_float]
    rng = np.random.default_rng(13)
    v = rng.normal(size=(200000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    sp = np.float32([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-30, -1e-30, 1e30, 3e-39, np.inf, -np.inf, 0.4375, 0.6875, 1.1875, 2.4375])
    gy, gx = np.meshgrid(sp, sp)
    y = np.concatenate([v[:, 1], rng.uniform(-1, 1, 100000) * 10.0 ** rng.uniform(-20, 20, 100000), gy.ravel()]).astype(np.float32)
    x = np.concatenate([v[:, 0], rng.uniform(-1, 1, 100000) * 10.0 ** rng.uniform(-20, 20, 100000), gx.ravel()]).astype(np.float32)
    x[200000:250000] = rng.uniform(-1, 1, 50000).astype(np.float32)  # (acosf's domain)
    a = np.zeros_like(x)
    c = np.zeros_like(x)
    _lib.check(_lib.lib.pine_gpu_test_atan(0, y.ctypes.data_as(_lib.c_f_p), x.ctypes.data_as(_lib.c_f_p), x.size,
                                           a.ctypes.data_as(_lib.c_f_p), c.ctypes.data_as(_lib.c_f_p)))
    with np.errstate(invalid="ignore"):
        ha = np.float32([libm.atan2f(float(p), float(q)) for p, q in zip(y, x)])
        hc = np.float32([libm.acosf(float(q)) for q in x])
    ok = ~np.isnan(hc)
    assert np.isnan(c[~ok]).all()
    assert_bit_equal(a, ha, "device atan2f vs libm")
    assert_bit_equal(c[ok], hc[ok], "device acosf vs libm")


@pytest.mark.parametrize("spp", [1, 16, 256])
def test_device_sampler_stream(spp):
    from pine_amd import _lib
    k = np.load(os.path.join(GOLDEN, f"sampler_spp{spp}.npz"))["k"]
    ref = (k.astype(np.float32) + np.float32(0.5)) / np.float32(256)
    out = np.zeros_like(ref)
    _lib.check(_lib.lib.pine_gpu_test_sampler(0, spp, out.ctypes.data_as(_lib.c_f_p), out.size))
    assert_bit_equal(out, ref, f"device BlueSobolSampler({spp})")


def test_device_rng_stream():
    from pine_amd import _lib
    ref = np.load(os.path.join(GOLDEN, "rng.npy"))
    out = np.zeros_like(ref)
    _lib.check(_lib.lib.pine_gpu_test_rng(0, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size))
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("which", ["shapes_zoo", "shapes_xzoo"])
def test_device_shape_records(which):
    from pine_amd import _lib, scenes
    z = np.load(os.path.join(GOLDEN, which + ".npz"))
    rec, rays = z["records"], np.ascontiguousarray(z["rays"])
    sc = scenes.shapes_zoo((48, 48)) if which == "shapes_zoo" else scenes.xshapes_zoo((48, 48))
    assert sc.describe() == str(z["pscene"])
    out = np.zeros_like(rec)
    _lib.check(_lib.lib.pine_gpu_test_shapes(sc._h, 0, rays.ctypes.data_as(_lib.c_f_p), len(rays),
                                             out.ctypes.data_as(_lib.c_f_p), out.size))
    assert_bit_equal(out[..., :3], rec[..., :3], "hit / intersect / tmax")
    hit = rec[..., 1] == 1
    # p, n and uv bit-exact (a sphere's uv goes through glibc's atan2f / acosf, restated in pine_libm.h)
    assert_bit_equal(out[hit][:, 3:9], rec[hit][:, 3:9], "surface p, n")
    assert_bit_equal(out[hit][:, 9:], rec[hit][:, 9:], "surface uv")


def _render(scene, spp, depth, **kw):
    import torch
    import pine_amd as pa
    w, h = scene.camera.film().size
    plan = pa.Plan(scene, spp, depth, **kw)
    film = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = plan.stats()
    out = film.cpu().numpy()
    plan.close()
    return out, st


LIBM_TOLERANCE_FILMS = set()  # (was {"mats_zoo_64_s32_d6"} until powf/logf became glibc-exact on the device)


def _scene_for(name):
    import pine_amd as pa
    from pine_amd import scenes
    return {
        "cbox_committed_64_s16_d4": lambda: scenes.cbox((64, 64), "committed"),
        "cbox_readme_64_s16_d4": lambda: scenes.cbox((64, 64), "readme"),
        "cbox_readme_64_s256_d8": lambda: scenes.cbox((64, 64), "readme"),
        "cbox_rect_readme_64_s64_d5": lambda: scenes.cbox((64, 64), "readme", False),
        "cbox_committed_ragged_45x37_s8_d3": lambda: scenes.cbox((45, 37), "committed"),
        "cbox_readme_64_s1_d1": lambda: scenes.cbox((64, 64), "readme"),
        "zoo_48_s16_d5": lambda: scenes.shapes_zoo((48, 48)),
        "classic_cones12_90x45_s32_d6": lambda: scenes.classic_cones((90, 45), 12),
        "sss_48_s32_d8": lambda: scenes.sss((48, 48), 1),
        "mats_zoo_64_s32_d6": lambda: scenes.materials_zoo((64, 64)),
        "classic_checker_cones8_90x45_s32_d6": lambda: scenes.classic_cones((90, 45), 8, checker_floor=True),
        "lights_zoo_64_s32_d6": lambda: scenes.lights_zoo((64, 64)),
        "lights_nosky_48_s16_d4": lambda: scenes.lights_zoo((48, 48), with_sky=False),
        "xshapes_48_s16_d5": lambda: scenes.xshapes_zoo((48, 48)),
        "xshapes_nolights_40_s8_d3": lambda: scenes.xshapes_zoo((40, 40), extra_lights=False),
        "mesh_glossy_48_s32_d6": lambda: scenes.sss((48, 48), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True),
    }[name]()


@pytest.mark.parametrize("name", FILM_NAMES)
def test_film_matches_reference_golden(name):
    """HIP film vs the REAL reference's film (tests/golden): bit for bit."""
    ref, ps, spp, depth = load_film(name)
    sc = _scene_for(name)
    assert sc.describe() == ps
    film, st = _render(sc, spp, depth)
    if name in LIBM_TOLERANCE_FILMS:
        # microfacet lobes with Schlick's powf under node-driven parameters: the device powf is within
        # 1 ulp of glibc's, not identical -- the declared tolerance of SURVEY.md 8(d) applies
        rel = np.linalg.norm(film[..., :3] - ref[..., :3], axis=2) / (np.linalg.norm(ref[..., :3], axis=2) + 1e-3)
        assert (rel <= 1e-4).mean() >= 0.999 and rel.max() <= 1e-2, (float((rel <= 1e-4).mean()), float(rel.max()))
        assert (film.view(np.uint32) != ref.view(np.uint32)).any(axis=2).mean() < 0.01  # and all but a few pixels are identical
        return
    assert_bit_equal(film, ref, name)


def _sobol_scene_for(name):
    from pine_amd import scenes
    return {
        "sobol_cbox_readme_48_s8_d4": lambda: scenes.cbox((48, 48), "readme"),
        "sobol_cbox_ragged_45x37_s12_d3": lambda: scenes.cbox((45, 37), "committed"),
        "sobol_mats_zoo_32_s16_d6": lambda: scenes.materials_zoo((32, 32)),
        "sobol_cbox_readme_24_s512_d5": lambda: scenes.cbox((24, 24), "readme"),
        "sobol_sss_32_s8_d6": lambda: scenes.sss((32, 32), 2),
    }[name]()


@pytest.mark.parametrize("name", SOBOL_FILM_NAMES)
def test_sobol_sampler_film_matches_reference_golden(name):
    """PathIntegrator(SobolSampler(spp), depth): HIP film vs the film the REAL reference rendered with SobolSampler,
    bit for bit (incl. 512 spp, above BlueSampler's cap, and 12 spp: a count that is not a power of two is one work item per
    pixel on the device, sampler.cpp:81-113 takes any)."""
    import pine_amd as pa
    ref, ps, spp, depth = load_film(name)
    sc = _sobol_scene_for(name)
    assert sc.describe() == ps
    film, st = _render(sc, pa.SobolSampler(spp), depth)
    assert (st.samples_per_item == spp) if spp & (spp - 1) else True
    assert st.spp_effective == spp
    assert_bit_equal(film, ref, name)


@pytest.mark.parametrize("name", HALTON_FILM_NAMES)
def test_halton_sampler_film_matches_reference_golden(name):
    """PathIntegrator(HaltonSampler(spp), depth) on the device (sampler.h:40-81: scrambled radical inverses, digit
    permutations derived on the host from a default-seeded RNG, pixel offsets through the 128 x 243 grid) against the film
    the REAL reference rendered with HaltonSampler, 8 and 12 samples per pixel (any count: sampler.h:44-46)."""
    import pine_amd as pa
    from pine_amd import scenes
    ref, ps, spp, depth = load_film(name)
    sc = {"halton_cbox_readme_40_s8_d4": lambda: scenes.cbox((40, 40), "readme"),
          "halton_mats_zoo_32_s12_d6": lambda: scenes.materials_zoo((32, 32)),
          "halton_sss_24x20_s12_d5": lambda: scenes.sss((24, 20), 1, camera="committed")}[name]()
    assert sc.describe() == ps
    film, st = _render(sc, pa.HaltonSampler(spp), depth)
    assert st.spp_effective == spp
    assert_bit_equal(film, ref, name)


def test_halton_sampler_matches_the_oracle(oracle, path_kernel):
    """... and against the CPU restatement (itself pinned by both reference films) on scenes with every material, node
    graphs, every light kind, meshes and 10 000-cone-class BVHs, up to 64 spp and depth 8 (42 sampler dimensions)."""
    import pine_amd as pa
    from pine_amd import scenes
    cases = [(scenes.materials_zoo((32, 32)), 16, 6), (scenes.lights_zoo((40, 40)), 8, 5), (scenes.classic_cones((64, 32), 12), 4, 6),
             (scenes.sss((32, 32), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True), 8, 6), (scenes.cbox((37, 29)), 64, 8)]
    for i, (sc, spp, depth) in enumerate(cases):
        w, h = sc.camera.film().size
        f, st = _render(sc, pa.HaltonSampler(spp), depth)
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth, sampler="halton")
        assert_bit_equal(f, ref, f"HaltonSampler case {i} ({path_kernel})")


@pytest.mark.parametrize("sampler", ["sobol", "halton"])
def test_sobol_and_halton_samplers_with_subsurface(oracle, path_kernel, sampler):
    """A BSSRDF walk draws three sampler dimensions per step and has no bound on its steps: SobolSampler's dimension counter never
    wraps (sampler.h:143-155) and HaltonSampler's wraps to 2 at the 1000-prime table's end (:52-63) -- both beyond the nine bits the
    packed path state keeps for BlueSampler (which wraps at 256).  The Subsurface + Sobol variants keep the counter in a word
    of its own; films equal the CPU restatement's, walks of hundreds of steps included (sigma_s 40 in a sphere of radius 0.4)."""
    import pine_amd as pa
    from pine_amd import scenes
    make = pa.SobolSampler if sampler == "sobol" else pa.HaltonSampler
    for sc, size, spp, depth in ((scenes.sss((32, 32), 2), (32, 32), 8, 6), (scenes.sss((24, 20), 1, camera="committed"), (24, 20), 12, 5)):
        film, st = _render(sc, make(spp), depth)
        ref, ost = oracle.render(sc.describe(), size, spp, depth, sampler=sampler)
        assert st.walk_steps > 0 or path_kernel == "mega"
        assert_bit_equal(film, ref, f"{sampler} sampler with Subsurface ({path_kernel})")


def test_sobol_sampler_limits():
    import pine_amd as pa
    from pine_amd import scenes
    with pytest.raises(pa.PineError, match="positive"):
        pa.Plan(scenes.cbox((16, 16)), pa.SobolSampler(0), 4)
    with pytest.raises(pa.PineError, match="at most 4096"):
        pa.Plan(scenes.cbox((16, 16)), pa.SobolSampler(5000), 4)


def test_one_shot_host_film_entry_point():
    """The drop-in form (host film out, as PathIntegrator(...).render(scene) in a .pine script)."""
    import pine_amd as pa
    ref, ps, spp, depth = load_film("cbox_committed_64_s16_d4")
    sc = _scene_for("cbox_committed_64_s16_d4")
    film = pa.PathIntegrator(pa.BlueSampler(spp), depth).render(sc).pixels
    assert_bit_equal(film, ref, "one-shot")
    assert sc.camera.film().finalize_u8().shape == (64, 64, 4)


@pytest.mark.parametrize("cfg", [
    ("cbox", (96, 80), "committed", True, 32, 8),
    ("cbox", (33, 17), "readme", True, 4, 2),
    ("cbox", (128, 128), "readme", False, 128, 8),
    ("cbox", (8, 8), "readme", True, 2, 12),
])
def test_film_matches_oracle(oracle, cfg, path_kernel):
    from pine_amd import scenes
    _, size, cam, boxes, spp, depth = cfg
    sc = scenes.cbox(size, cam, boxes)
    film, st = _render(sc, spp, depth)
    assert st.block_threads == (1024 if path_kernel == "queue" else 256)  # the kernel under test is the one that ran
    ref, ost = oracle.render(sc.describe(), size, spp, depth)
    assert_bit_equal(film, ref, str(cfg))
    assert st.vertices == ost.vertices and st.shadow_rays == ost.shadow_rays  # same work, vertex for vertex


def test_per_sample_radiance_matches_oracle(oracle):
    import torch
    import pine_amd as pa
    from pine_amd import scenes
    sc = scenes.cbox((24, 16), "readme")
    plan = pa.Plan(sc, 16, 6)
    film = torch.zeros((16, 24, 4), device="cuda")
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    got = plan.read_samples()
    ref = oracle.render_samples(sc.describe(), (24, 16), 16, 16, 6)
    assert_bit_equal(got, ref, "per-sample radiance and vertex counts")


def test_scheduling_invariance():
    """Work-item size, shard count and launch repetition must not change a single bit."""
    from pine_amd import scenes
    sc = scenes.cbox((72, 40), "readme")
    base, _ = _render(sc, 64, 8)
    for spi in (1, 2, 16, 64):
        f, st = _render(sc, 64, 8, samples_per_item=spi)
        assert st.samples_per_item == spi
        assert_bit_equal(f, base, f"samples_per_item={spi}")
    for world in (2, 3, 8):
        tot = np.zeros_like(base)
        for r in range(world):
            f, _ = _render(sc, 64, 8, shard_rank=r, shard_world=world)
            tot += f  # disjoint shards, zeros elsewhere: exact
        assert_bit_equal(tot, base, f"{world} shards")
    again, _ = _render(sc, 64, 8)
    assert_bit_equal(again, base, "run-to-run determinism")


def test_packed_slabs_unpack_to_the_same_film():
    """The multi-GPU form bench.py uses: per-rank tile-major slabs, concatenated as a gather would, then
    scattered on the device -- must equal the one-rank film bit for bit (ragged film: border tiles)."""
    import torch
    import pine_amd as pa
    from pine_amd import scenes
    sc = scenes.cbox((77, 45), "readme")
    base, _ = _render(sc, 16, 5)
    for world in (1, 2, 5):
        slabs = []
        for r in range(world):
            plan = pa.Plan(sc, 16, 5, shard_rank=r, shard_world=world)
            slab = torch.full((plan.slab_floats(),), -7.0, dtype=torch.float32, device="cuda")
            plan.launch_packed(slab.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            slabs.append(slab)
            plan.close()
        allslabs = torch.stack(slabs)
        film = torch.full((45, 77, 4), -1.0, dtype=torch.float32, device="cuda")
        pa.film_unpack((77, 45), world, allslabs.data_ptr(), film.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert_bit_equal(film.cpu().numpy(), base, f"packed, {world} ranks")


def test_tile_classes_under_sharding_and_packed_slabs(path_kernel):
    """Subsurface scenes with tile classes (the shard's tiles reordered: whole-pixel tiles first): shards still sum to the
    one-rank film, the packed slab of a rank stays in the shard's NATURAL tile order (what the gather + unpack of the
    multi-GPU path rely on), per-sample radiance comes back in film order, and a ragged film works."""
    import torch
    import pine_amd as pa
    from pine_amd import scenes
    sc = scenes.sss((77, 45), 2)
    base, st = _render(sc, 32, 6)
    if path_kernel == "queue":
        assert 0 < st.serial_tiles < 10 * 6
    for world in (2, 3):
        tot = np.zeros_like(base)
        slabs = []
        for r in range(world):
            f, st_r = _render(sc, 32, 6, shard_rank=r, shard_world=world)
            tot += f
            plan = pa.Plan(sc, 32, 6, shard_rank=r, shard_world=world)
            slab = torch.full((plan.slab_floats(),), -7.0, dtype=torch.float32, device="cuda")
            plan.launch_packed(slab.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            slabs.append(slab)
            plan.close()
        assert_bit_equal(tot, base, f"subsurface scene, {world} shards")
        film = torch.full((45, 77, 4), -1.0, dtype=torch.float32, device="cuda")
        pa.film_unpack((77, 45), world, torch.stack(slabs).data_ptr(), film.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert_bit_equal(film.cpu().numpy(), base, f"subsurface scene packed, {world} ranks")


def test_edge_cases():
    import pine_amd as pa
    # empty scene: every ray misses -> black film with w = 1 (path.cpp:38,75-81)
    s = pa.Scene()
    s.set(pa.ThinLenCamera(pa.Film([16, 8]), [0, 0, 0], [0, 0, 1], 0.4))
    f, st = _render(s, 4, 4)
    assert (f[..., :3] == 0).all() and (f[..., 3] == 1).all() and st.vertices == 16 * 8 * 4
    # geometry but no light: NEE draws its three dimensions and finds no light (lightsampler.cpp:13-14)
    s.add("d", pa.Diffuse([0.5, 0.5, 0.5]))
    s.add(pa.Rect([0, 0, 2], [4, 0, 0], [0, 4, 0]), "d")
    f, _ = _render(s, 4, 4)
    assert (f[..., :3] == 0).all()
    # thin-lens branch consumes the lens sample (camera.cpp:26-32)
    from pine_amd import scenes
    sc = scenes.cbox((32, 32), "readme")
    sc.set(pa.ThinLenCamera(pa.Film([32, 32]), [0, 1, -4], [0, 1, 0], 0.25, 0.05, 4.0))
    f, _ = _render(sc, 16, 4)
    from oracle import oracle as o
    ref, _ = o.render(sc.describe(), (32, 32), 16, 4)
    assert_bit_equal(f, ref, "thin lens")
    # spp request above 256 clamps (sampler.cpp:116-119); non-power-of-two rounds up
    f512, st = _render(scenes.cbox((16, 16)), 512, 3)
    assert st.spp_effective == 256
    f5, st = _render(scenes.cbox((16, 16)), 5, 3)
    assert st.spp_effective == 8


def test_shards_without_tiles_and_tiny_films(oracle):
    """More ranks than tiles: a rank that owns no tile launches, finishes and writes zeros; the shards still sum
    to the full film.  And a 1x1 film (one partial tile)."""
    from pine_amd import scenes
    sc = scenes.cbox((24, 16), "readme")  # 3 x 2 = 6 tiles
    ref, _ = oracle.render(sc.describe(), (24, 16), 16, 4)
    tot = np.zeros_like(ref)
    for r in range(8):
        f, st = _render(sc, 16, 4, shard_rank=r, shard_world=8)
        if r >= 6:
            assert st.camera_samples == 0 and (f == 0).all()
        tot += f
    assert_bit_equal(tot, ref, "8 shards of a 6-tile film")
    # a scene whose only geometry is a mesh without triangles: no top-level primitive at all (bvh.cpp:458 skips it)
    import pine_amd as pa
    em = pa.Scene()
    em.add("d", pa.Diffuse([0.5, 0.5, 0.5]))
    em.add(pa.Mesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32)), "d")
    em.set(pa.ThinLenCamera(pa.Film([8, 8]), [0, 0, 0], [0, 0, 1], 0.4))
    f, st = _render(em, 4, 3)
    assert (f[..., :3] == 0).all() and (f[..., 3] == 1).all() and st.vertices == 8 * 8 * 4
    one = scenes.cbox((1, 1), "readme")
    f, st = _render(one, 8, 3)
    ref1, _ = oracle.render(one.describe(), (1, 1), 8, 3)
    assert_bit_equal(f, ref1, "1x1 film")


def test_small_meshes_and_two_area_lights(oracle):
    """A one-triangle and a two-triangle mesh (leaf-root mesh BVHs, one of them emissive), and two emissive Rects
    in the lean Rect + Box variant (the light sampler's N != 1 branch, lightsampler.cpp:13-29)."""
    import pine_amd as pa
    from pine_amd import scenes
    sc = scenes.cbox((40, 40), "readme", boxes=False)
    sc.add("m", pa.Diffuse([0.7, 0.3, 0.2]))
    sc.add(pa.Mesh(np.float32([[-0.5, 0.2, 1.2], [0.3, 0.2, 1.4], [-0.1, 1.0, 1.3]]), np.uint32([[0, 1, 2]])), "m")
    sc.add(pa.Mesh(np.float32([[0.2, 0.1, 0.8], [0.8, 0.1, 0.9], [0.8, 0.7, 0.9], [0.2, 0.7, 0.8]]), np.uint32([[0, 1, 2], [0, 2, 3]])),
           pa.Emissive([6.0, 5.0, 3.0]))
    f, _ = _render(sc, 16, 5)
    ref, _ = oracle.render(sc.describe(), (40, 40), 16, 5)
    assert_bit_equal(f, ref, "small meshes")
    two = scenes.cbox((40, 40), "readme")
    two.add(pa.Rect([0.5, 1.0, 1.9], [0.3, 0, 0], [0, 0.3, 0], True), pa.Emissive([30.0, 10.0, 5.0]))
    f, _ = _render(two, 16, 5)
    ref, _ = oracle.render(two.describe(), (40, 40), 16, 5)
    assert_bit_equal(f, ref, "two area lights")


def test_random_scenes_match_the_oracle(oracle):
    """Fuzzing on the device: 40 seeded random scenes (every shape / material / light kind, random spp and depth)."""
    from pine_amd import scenes
    for seed in range(3000, 3040):
        sc, spp, depth, sampler = scenes.random_scene(seed, variety=True)  # + odd film sizes, thin lens, SobolSampler
        w, h = sc.camera.film().size
        f, _ = _render(sc, spp, depth, sampler=sampler)
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth, sampler=sampler)
        assert_bit_equal(f, ref, f"random scene {seed}")
    for seed in range(4000, 4015):  # fractional Uber lobes (in-path RNG: serial samples) and Subsurface meshes (megakernel)
        sc, spp, depth, sampler = scenes.random_scene(seed, variety=2)
        w, h = sc.camera.film().size
        f, _ = _render(sc, spp, depth, sampler=sampler)
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth, sampler=sampler)
        assert_bit_equal(f, ref, f"random scene {seed} (in-path RNG)")


def test_random_scenes_shard_sums_and_packed_slabs():
    """Fuzzing the multi-GPU decomposition: for random scenes and world sizes, the shards' films sum to the
    one-GPU film, and the packed slabs unpack to it (what bench.py gathers)."""
    import torch
    import pine_amd as pa
    from pine_amd import scenes
    rng = np.random.default_rng(5)
    for seed in range(5000, 5012):
        sc, spp, depth, sampler = scenes.random_scene(seed, variety=1)
        w, h = sc.camera.film().size
        full, _ = _render(sc, spp, depth, sampler=sampler)
        world = int(rng.choice([2, 3, 5, 8]))
        tot = np.zeros_like(full)
        n = int(pa._lib.lib.pine_gpu_packed_slab_floats(w, h, world))
        slabs = torch.zeros((world, n), dtype=torch.float32, device="cuda")
        for r in range(world):
            f, _ = _render(sc, spp, depth, sampler=sampler, shard_rank=r, shard_world=world)
            tot += f
            plan = pa.Plan(sc, spp, depth, sampler=sampler, shard_rank=r, shard_world=world)
            plan.launch_packed(slabs[r].data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            plan.close()
        assert_bit_equal(tot, full, f"scene {seed}: {world} shards")
        film = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        pa.film_unpack((w, h), world, slabs.data_ptr(), film.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert_bit_equal(film.cpu().numpy(), full, f"scene {seed}: packed slabs of {world} ranks")


def test_errors_are_reported_not_fatal():
    import pine_amd as pa
    from pine_amd import scenes
    with pytest.raises(pa.PineError, match="max_path_length"):
        pa.Plan(scenes.cbox((16, 16)), 4, 0)
    with pytest.raises(pa.PineError, match="fold-stack depth"):
        pa.Plan(scenes.cbox((16, 16)), 4, 1000)
    with pytest.raises(pa.PineError, match="camera"):
        pa.Plan(pa.Scene(), 4, 4)


def test_c1_full_size_whole_image_md5():
    """BASELINE config C1 (640x640, 16 spp, depth 4) at full size: the whole film's md5 equals the
    REAL reference's (tests/golden/stats_640.json)."""
    from pine_amd import scenes
    st = json.load(open(os.path.join(GOLDEN, "stats_640.json")))["C1_cbox_640_s16_d4_committed"]
    film, _ = _render(scenes.cbox((640, 640), "committed"), 16, 4)
    assert hashlib.md5(film.tobytes()).hexdigest() == st["md5"]
    np.testing.assert_allclose(film[..., :3].mean(axis=(0, 1), dtype=np.float64), st["mean_rgb"], rtol=1e-12)


@pytest.mark.parametrize("name", ["C2_cbox_640_s256_d8_readme", "C3_cbox_1920x1080_s1024_d8", "C4_classic_10k_cones_720x360_s64_d6",
                                  "C5_sss_320_s512_d8", "C5_sss_640_s512_d8"])
def test_other_baseline_configs_whole_image_md5(name):
    """C2 with the README camera, BASELINE configs C3 (1920x1080, BlueSampler(1024) -> 256 spp), C4 (classic + 10 000 cones) at full size and
    C5 (Subsurface icosphere, a quarter of the film and BASELINE's 640 x 640): the whole film's md5 equals the REAL
    reference's (both path kernels; the stage-queued one runs the BSSRDF walk as its stage W)."""
    from pine_amd import scenes
    st = json.load(open(os.path.join(GOLDEN, "stats_640.json")))[name]
    sc = {"C2_": lambda: scenes.cbox((640, 640), "readme"),  # the README camera sees the whole room (V = 5.06)
          "C3_": lambda: scenes.cbox((1920, 1080), "committed"),
          "C4_": lambda: scenes.classic_cones((720, 360), 100),
          "C5_sss_320": lambda: scenes.sss((320, 320), 3),
          "C5_sss_640": lambda: scenes.sss((640, 640), 3)}[name[:3] if name[1] != "5" else name[:10]]()
    film, _ = _render(sc, st["spp"], st["depth"])
    assert hashlib.md5(film.tobytes()).hexdigest() == st["md5"]
    np.testing.assert_allclose(film[..., :3].mean(axis=(0, 1), dtype=np.float64), st["mean_rgb"], rtol=1e-12)


def test_c2_full_size_whole_image_md5_and_properties():
    """BASELINE config C2 (640x640, 256 spp, depth 8): md5 vs the reference, plus size-independent
    properties: shard sum == whole, black lower half (camera on the floor plane), vertex count."""
    from pine_amd import scenes
    st = json.load(open(os.path.join(GOLDEN, "stats_640.json")))["C2_cbox_640_s256_d8_committed"]
    sc = scenes.cbox((640, 640), "committed")
    film, ps = _render(sc, 256, 8)
    assert hashlib.md5(film.tobytes()).hexdigest() == st["md5"]
    assert int((film[..., :3] == 0).all(axis=2).sum()) == st["black_pixels"]
    assert (film[:319, :, :3] == 0).all()  # rows below the horizon escape on the first ray
    assert abs(ps.vertices / ps.camera_samples - 2.6425) < 1e-3
    tot = np.zeros_like(film)
    for r in range(2):
        f, _ = _render(sc, 256, 8, shard_rank=r, shard_world=2)
        tot += f
    assert_bit_equal(tot, film, "2 shards at full size")


def test_kernel_bail_out_is_reported_by_every_entry_point(path_kernel):
    """A protocol failure of the stage-queued kernel (bounded wait ran out) leaves an incomplete film: the
    one-shot render, stats_get and plan_check must all FAIL with the bail-out code, never return that film.
    PINE_GPU_FLAG_DEBUG_FORCE_BAIL raises the bail-out deterministically; the next ordinary launch is clean."""
    import torch
    import pine_amd as pa
    from pine_amd import scenes, _lib
    if path_kernel == "mega":
        pytest.skip("the lane-owns-a-path kernel has no waits to bail out of")
    sc = scenes.cbox((64, 64), "readme")
    with pytest.raises(pa.PineError, match="bailed out.*code 7"):
        pa.PathIntegrator(pa.BlueSampler(8), 4, flags=_lib.FLAG_DEBUG_FORCE_BAIL).render(sc)
    plan = pa.Plan(sc, 8, 4, flags=_lib.FLAG_DEBUG_FORCE_BAIL)
    film = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda")
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    with pytest.raises(pa.PineError, match="bailed out"):
        plan.check()
    with pytest.raises(pa.PineError, match="bailed out"):
        plan.stats()
    plan.close()
    good = pa.Plan(sc, 8, 4)
    good.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    good.check()
    ref = pa.PathIntegrator(pa.BlueSampler(8), 4).render(sc).pixels
    assert_bit_equal(film.cpu().numpy(), ref, "launch after a bailed-out one")
    good.close()


def test_subsurface_runs_on_the_stage_queued_kernel(path_kernel):
    """Subsurface scenes take the stage-queued kernel (walk stage W) by default: 1024-thread workgroups; the
    film equals the megakernel's and the oracle's bit for bit (mesh walk; sphere walk, whose exit point is the
    reference's zero vector -- SURVEY.md Appendix A5; a textured skin evaluated at the ENTRY point)."""
    import pine_amd as pa
    from pine_amd import scenes
    from oracle import oracle
    cases = [scenes.sss((40, 40), 1), scenes.sss((40, 40), 2, skin=pa.Subsurface([0.9, 0.6, 0.4], 0.3, [12.0, 25.0, 40.0]))]
    sph = scenes.cbox((36, 36), "readme", boxes=False)
    sph.add("skin", pa.Subsurface([0.8, 0.8, 0.7], 0.2, [30.0, 30.0, 30.0]))
    sph.add(pa.Sphere([0.1, 0.5, 1.0], 0.4), "skin")
    cases.append(sph)
    for i, sc in enumerate(cases):
        w, h = sc.camera.film().size
        f, st = _render(sc, 16, 8)
        assert st.block_threads == (256 if path_kernel == "mega" else 1024)
        # in-path RNG draws: a pixel's samples stay sequential -- in every pixel, or (tile classes) in the tiles from which
        # a camera ray can reach the Subsurface shape
        assert st.samples_per_item == st.spp_effective or (path_kernel == "queue" and st.serial_tiles > 0)
        ref, _ = oracle.render(sc.describe(), (w, h), 16, 8)
        assert_bit_equal(f, ref, f"subsurface case {i} ({path_kernel})")


def test_plan_limits_are_errors_not_wraparound():
    """Pixel coordinates travel as 16 + 16 bits and a context's sample-buffer index as 32 bits: larger renders are refused."""
    import pine_amd as pa
    from pine_amd import scenes
    with pytest.raises(pa.PineError, match="65535"):
        pa.Plan(scenes.cbox((70000, 8)), 1, 2)
    with pytest.raises(pa.PineError, match="2\\^32"):
        pa.Plan(scenes.cbox((4096, 4096)), pa.SobolSampler(512), 2)


@pytest.mark.parametrize("min_lanes,min_trips", [(0, 8), (24, 8), (64, 1)])
def test_traversal_stage_refill_settings_change_nothing(oracle, monkeypatch, path_kernel, min_lanes, min_trips):
    """pine_queue_kernel.h, stages XS / XC: a wave keeps its lanes' traversals in registers, retires the finished ones
    and hands the free lanes new rays from the queue whenever fewer than `min_lanes` still travel (checked every
    `min_trips` trips).  The setting decides which rays share a wave and when, never what a ray computes: 0 never
    refills (a wave runs its rays to the end), (64, 1) retires and refills after every single trip."""
    import pine_amd as pa
    from pine_amd import scenes
    if path_kernel == "mega":
        pytest.skip("the stage-queued kernel's machinery")
    monkeypatch.setenv("PINE_GPU_TRAV_MIN_LANES", str(min_lanes))
    monkeypatch.setenv("PINE_GPU_TRAV_MIN_TRIPS", str(min_trips))
    cases = [(scenes.classic_cones((96, 48), 40), 16, 6),
             (scenes.sss((48, 48), 2), 16, 8),
             (scenes.sss((40, 40), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True), 16, 6),
             (scenes.lights_zoo((40, 40)), 8, 5)]
    # lights_zoo is a cbox-class scene (whole scene in LDS: traversal inside the stages) unless LDS staging is refused
    for i, (sc, spp, depth) in enumerate(cases):
        if i == 3:
            monkeypatch.setenv("PINE_GPU_NO_LDS_SCENE", "1")
        w, h = sc.camera.film().size
        f, st = _render(sc, spp, depth)
        assert st.block_threads == 1024
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth)
        assert_bit_equal(f, ref, f"traversal stages, case {i}")


@pytest.mark.parametrize("xstage", ["0", "1"])
def test_traversal_as_stages_or_inside_stages_changes_nothing(oracle, monkeypatch, path_kernel, xstage):
    """Every F_LDS_TOP feature set that can hold meshes is compiled twice: traversal as stages XS / XC (F_XSTAGE; what
    plan_build picks for mesh scenes whose BVH fits the LDS node cache) and the same flat traversal inside stages S / T
    (analytic scenes; BVHs that stay in L2).  PINE_GPU_XSTAGE forces one or the other on scenes of both classes."""
    import pine_amd as pa
    from pine_amd import scenes
    if path_kernel == "mega":
        pytest.skip("the stage-queued kernel's machinery")
    monkeypatch.setenv("PINE_GPU_XSTAGE", xstage)
    cases = [(scenes.classic_cones((96, 48), 40), 16, 6),
             (scenes.sss((48, 48), 2), 16, 8),
             (scenes.sss((40, 40), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True), 16, 6),
             (scenes.xshapes_zoo((40, 40)), 8, 5)]
    for i, (sc, spp, depth) in enumerate(cases):
        if i == 3:
            monkeypatch.setenv("PINE_GPU_NO_LDS_SCENE", "1")  # (a cbox-class scene otherwise: whole scene in LDS)
        w, h = sc.camera.film().size
        f, st = _render(sc, spp, depth)
        assert st.block_threads == 1024
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth)
        assert_bit_equal(f, ref, f"PINE_GPU_XSTAGE={xstage}, case {i}")


@pytest.mark.parametrize("env", [{}, {"PINE_GPU_NO_FORK": "1"}, {"PINE_GPU_POOL_ITEMS": "16"}, {"PINE_GPU_POOL_ITEMS": "4096"},
                                 {"PINE_GPU_MAX_PIXELS": "1"}, {"PINE_GPU_MAX_PIXELS": "100000"},
                                 # ... and without tile classes (every pixel a whole-pixel item, as before round 3)
                                 {"PINE_GPU_NO_TILE_CLASSES": "1"}, {"PINE_GPU_NO_TILE_CLASSES": "1", "PINE_GPU_POOL_ITEMS": "16"},
                                 {"PINE_GPU_NO_TILE_CLASSES": "1", "PINE_GPU_POOL_ITEMS": "4096"},
                                 {"PINE_GPU_NO_TILE_CLASSES": "1", "PINE_GPU_MAX_PIXELS": "1"},
                                 # the stage picker's starvation guard: never / the walk queue at every pick / the shortest queue at every pick
                                 {"PINE_GPU_FAIR_PERIOD": "0"}, {"PINE_GPU_FAIR_PERIOD": "1"}, {"PINE_GPU_FAIR_PERIOD": "-1"},
                                 {"PINE_GPU_NO_TILE_CLASSES": "1", "PINE_GPU_FAIR_PERIOD": "0"}])
def test_sample_tokens_change_nothing(oracle, monkeypatch, path_kernel, env):
    """Subsurface scenes: a pixel's samples are sequentially dependent through the pixel's RNG (the BSSRDF channel pick,
    bxdf.cpp:335), but only until a path's first non-delta bounce -- after it the path draws no RNG value any more, so it
    releases a token (pixel, next sample index, RNG state) and ANOTHER context starts the pixel's next sample while this
    path is still being traced.  Same films with the mechanism off, with it on, whatever the size of a workgroup's
    work-item claims and whatever the number of pixels it may have in flight (default 320; 1: a pixel at a time); a film larger than one round of contexts and 64 samples per pixel, so that tokens, waiting
    contexts and wake-ups all occur."""
    import pine_amd as pa
    from pine_amd import scenes
    if path_kernel == "mega":
        pytest.skip("the stage-queued kernel's machinery")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    mixed = scenes.sss((40, 40), 2, emissive_mesh=True)  # the all-features variant (a second mesh that is a light)
    mixed.add("gold", pa.Metal([0.9, 0.7, 0.3], 0.2))
    mixed.add(pa.Sphere([0.55, 0.3, 0.5], 0.25), "gold")
    cases = [(scenes.sss((72, 56), 2), 64, 8), (scenes.sss((24, 24), 1), 256, 6), (mixed, 32, 6)]
    for i, (sc, spp, depth) in enumerate(cases):
        w, h = sc.camera.film().size
        f, st = _render(sc, spp, depth)
        assert st.block_threads == 1024 and (st.samples_per_item == st.spp_effective or st.serial_tiles > 0)
        if "PINE_GPU_NO_TILE_CLASSES" in env or "PINE_GPU_NO_FORK" in env:
            assert st.serial_tiles == 0
        elif i < 2:  # (Diffuse + Subsurface + Emissive only, pinhole camera: the tiles away from the mesh are independent items)
            assert 0 < st.serial_tiles < ((w + 7) // 8) * ((h + 7) // 8)
        else:  # a Metal sphere: a delta-capable material, every pixel stays one item
            assert st.serial_tiles == 0
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth)
        assert_bit_equal(f, ref, f"sample tokens {env}, case {i}")


@pytest.mark.parametrize("lds_tris", ["0", "1"])
def test_triangle_packets_in_lds_change_nothing(oracle, monkeypatch, path_kernel, lds_tris):
    """Mesh triangles as LDS packets (8-byte entries of 16-bit vertex numbers + the scene's distinct vertices as float4,
    DeviceScene::tri_packets): the traversal stages and the BSSRDF walk read the same floats from LDS instead of the 48-byte
    leaf records in memory.  Forced off and forced on (whenever they fit), against the oracle; a mesh that is a light, a
    mesh with per-vertex normals (the imported glTF scene: shading reads tri_verts / tri_attrs as before)."""
    import pine_amd as pa
    from pine_amd import scenes, gltf
    if path_kernel == "mega":
        pytest.skip("the stage-queued kernel's machinery")
    monkeypatch.setenv("PINE_GPU_LDS_TRIS", lds_tris)
    cases = [(scenes.sss((48, 48), 3), 16, 8),
             (scenes.sss((40, 40), 2, skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True), 16, 6),
             (gltf.load(os.path.join(GOLDEN, "import_test.glb")), 4, 5)]
    for i, (sc, spp, depth) in enumerate(cases):
        w, h = sc.camera.film().size
        if i == 2:
            w, h = 96, 96
            sc.set(pa.ThinLenCamera(pa.Film([w, h]), sc.camera.frm, sc.camera.to, sc.camera.fov))
        f, st = _render(sc, spp, depth)
        assert st.block_threads == 1024
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth)
        assert_bit_equal(f, ref, f"triangle packets {lds_tris}, case {i}")


def test_one_process_several_devices_entry_point(oracle):
    """pine_gpu_path_render_devices / _multi (SURVEY.md 8(b)'s device_mask form): shards on a list of devices, slabs
    gathered with peer copies, one film -- bit-identical to the one-device render.  On the 1-GPU box the list names
    device 0 three times (three plans, three streams, three slabs); ragged film: border tiles and an empty shard."""
    import ctypes as C
    import pine_amd as pa
    from pine_amd import scenes, _lib
    for sc, spp, depth in ((scenes.cbox((77, 45), "readme"), 16, 5), (scenes.sss((24, 24), 1), 8, 6), (scenes.cbox((9, 7)), 4, 3)):
        one = pa.PathIntegrator(pa.BlueSampler(spp), depth).render(sc).pixels.copy()
        for devs in ([0, 0], [0, 0, 0], [0] * 5):
            many = pa.PathIntegrator(pa.BlueSampler(spp), depth, devices=devs).render(sc).pixels
            assert_bit_equal(many, one, f"{len(devs)} shards in one process")
        prm = _lib.RenderParams(spp, depth, 0, 0, 1, 0, 0, 0)
        out = np.zeros_like(one)
        _lib.check(_lib.lib.pine_gpu_path_render_multi(sc._h, C.byref(prm), 1, out.ctypes.data_as(_lib.c_f_p)))
        assert_bit_equal(out, one, "device_mask = 1")
        assert _lib.lib.pine_gpu_path_render_multi(sc._h, C.byref(prm), 0, out.ctypes.data_as(_lib.c_f_p)) < 0
        assert _lib.lib.pine_gpu_path_render_multi(sc._h, C.byref(prm), 1 << 40, out.ctypes.data_as(_lib.c_f_p)) < 0


@pytest.mark.parametrize("name", ["embree_cbox_readme_64_s16_d4", "embree_cbox_committed_64_s16_d4", "embree_cbox_rect_readme_64_s64_d5"])
def test_gpu_film_against_both_oracles(name):
    """the GPU film against BOTH oracles -- bit-equal to the reference with pine's own BVH (the order
    this build reproduces), at the reference's own BVH-vs-Embree distance from the EmbreeAccel oracle,
    and bit-equal to both on the Rect-only scene."""
    from conftest import EMBREE_EXPECTED, film_distance
    from pine_amd import scenes
    emb, _, spp, depth = load_film(name)
    bvh, _, _, _ = load_film(name.replace("embree_", ""))
    sc = {"embree_cbox_readme_64_s16_d4": lambda: scenes.cbox((64, 64), "readme"),
          "embree_cbox_committed_64_s16_d4": lambda: scenes.cbox((64, 64), "committed"),
          "embree_cbox_rect_readme_64_s64_d5": lambda: scenes.cbox((64, 64), "readme", False)}[name]()
    film, _ = _render(sc, spp, depth)
    assert_bit_equal(film, bvh, "GPU vs O-gcc-bvh")
    d = film_distance(film, emb)
    lo, hi = EMBREE_EXPECTED[name]["identical"]
    assert lo <= d["identical"] <= hi, d
    if "rect" in name:
        assert_bit_equal(film, emb, "GPU vs O-gcc-embree on the Rect-only scene")


# PINE_GPU_FLAG_FAST's declared tolerance (include/pine_gpu.h).  Sampler / RNG / hash streams stay exact; every float
# operation may differ in its last bits.  Where the path is a continuous function of those bits the film agrees to ~1e-7
# (Rect-only cbox below: 99.96 % of the pixels within 1e-4, RMSE 4e-6).  Where the reference's own algorithm makes discontinuous decisions on
# nearly equal numbers -- the scaled-OBB world-tmax clip (bbox.cpp:150-172), grazing hits on 10 000 tiny cones, near-delta
# microfacet lobes -- a last-bit change sends a sample down another path (0.8 - 2 % of the samples), exactly as a different
# compiler does to the reference itself (clang vs g++ builds of pine).  Those films are compared as
# Monte-Carlo estimates: whole-image RMSE at 256 spp and the bias of the image mean.
FAST_TOLERANCE = {  # case: (min share of pixels within rel-L2 1e-4, max RMSE at 256 spp, max |relative bias of the image mean|)
    "cbox_rect_256": (0.999, 1e-4, 1e-5),  # SURVEY.md 8(d)'s criterion for this very scene: >= 99.9 % of the pixels within 1e-4
    "cbox_readme_256": (0.10, 4e-3, 3e-3),
    "cbox_committed_256": (0.60, 2e-3, 3e-3),
    "classic_cones": (0.40, 2e-2, 5e-3),
    "sss": (0.85, 1e-3, 3e-3),
}


@pytest.mark.parametrize("case", list(FAST_TOLERANCE))
def test_fast_mode_within_declared_tolerance(case, path_kernel):
    """PINE_GPU_FLAG_FAST (pine_kernels_fast.hip: contracted multiply-adds, 1-ulp rcp / sqrt / division, native
    sin / cos / pow / log) against the exact film of the same scene.  Exact mode stays the parity gate."""
    import pine_amd as pa
    from conftest import film_distance
    from pine_amd import scenes, _lib
    if path_kernel == "mega":
        pytest.skip("fast variants exist for the stage-queued kernel only")
    sc, spp, depth = {"cbox_rect_256": (scenes.cbox((96, 96), "readme", False), 256, 8),
                      "cbox_readme_256": (scenes.cbox((96, 96), "readme"), 256, 8),
                      "cbox_committed_256": (scenes.cbox((96, 96), "committed"), 256, 8),
                      "classic_cones": (scenes.classic_cones((144, 72), 100), 256, 6),
                      "sss": (scenes.sss((64, 64), 3), 256, 8)}[case]
    exact, st0 = _render(sc, spp, depth)
    fast, st1 = _render(sc, spp, depth, flags=_lib.FLAG_FAST)
    assert st1.block_threads == 1024 and np.isfinite(fast).all()
    d = film_distance(fast, exact)
    me, mf = exact[..., :3].mean(dtype=np.float64), fast[..., :3].mean(dtype=np.float64)
    bias = abs(mf - me) / me
    print(case, d, "bias of the image mean %.2e" % bias, "vertices/sample exact %.4f fast %.4f" % (st0.vertices / st0.camera_samples, st1.vertices / st1.camera_samples))
    share, rmse, max_bias = FAST_TOLERANCE[case]
    assert d["within_1e-4"] >= share and d["rmse"] <= rmse and bias <= max_bias, (d, bias)


def test_fast_mode_refuses_scenes_it_has_no_variant_for():
    import pine_amd as pa
    from pine_amd import scenes, _lib
    with pytest.raises(pa.PineError, match="PINE_GPU_FLAG_FAST"):
        pa.Plan(scenes.lights_zoo((16, 16)), 4, 4, flags=_lib.FLAG_FAST)


def _accel_dump(sc, device=None):
    from pine_amd import _lib
    n = _lib.check(_lib.lib.pine_gpu_scene_build_accel(sc._h) if device is None else _lib.lib.pine_gpu_scene_build_accel_device(sc._h, device), "build_accel")
    nodes = np.zeros((max(n, 1), 16), np.float32)
    prims = np.zeros(200_000, np.int32)
    npr = _lib.lib.pine_gpu_scene_accel_dump(sc._h, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes, prims.ctypes.data_as(C.POINTER(C.c_int32)), prims.size)
    return nodes[:n].view(np.uint32).copy(), prims[:npr].copy()


def test_device_bvh_build_produces_the_host_tree(oracle):
    """the BVH built on the GPU (pine_bvh_build_device.h: decide / scan / split kernels per level) against the host
    build of the same schedule -- node array and primitive order bit for bit, on the 10 000-cone scene (5 692 nodes, depth 15),
    meshes (two-level), cbox (order-dependent OBBs), one-primitive and empty scenes, random scenes; then a render on that tree."""
    import pine_amd as pa
    from pine_amd import scenes, gltf, _lib
    cases = [scenes.classic_cones((64, 32), 100), scenes.sss((32, 32), 3), scenes.sss((32, 32), 2, emissive_mesh=True), scenes.cbox((32, 32)),
             scenes.xshapes_zoo((32, 32)), gltf.load(os.path.join(GOLDEN, "import_test.glb"))]
    cases += [scenes.random_scene(seed, variety=2)[0] for seed in range(5000, 5010)]
    one = pa.Scene()
    one.add(pa.Sphere([0, 0, 2], 0.5), pa.Diffuse([0.5, 0.5, 0.5]))
    one.set(pa.ThinLenCamera(pa.Film([8, 8]), [0, 0, 0], [0, 0, 1], 0.4))
    cases.append(one)
    for i, sc in enumerate(cases):
        hn, hp = _accel_dump(sc)
        dn, dp = _accel_dump(sc, device=0)
        assert hn.shape == dn.shape and np.array_equal(hn, dn), f"case {i}: node arrays differ"
        assert np.array_equal(hp, dp), f"case {i}: primitive order differs"
    sc = scenes.classic_cones((96, 48), 40)
    f, st = _render(sc, 16, 6, flags=_lib.FLAG_DEVICE_BVH)
    assert st.accel_built_on_device == 1
    ref, _ = oracle.render(sc.describe(), (96, 48), 16, 6)
    assert_bit_equal(f, ref, "render on the device-built BVH")


# ---- PINE_GPU_FLAG_ORDER_EMBREE (SURVEY.md Appendix A3's second traversal order: the reference's default accel's) ---------------
from conftest import EMBREE_FILM_NAMES, EMBREE_MORE_FILM_NAMES, embree_scene  # noqa: E402


@pytest.mark.parametrize("name", EMBREE_FILM_NAMES + EMBREE_MORE_FILM_NAMES)
def test_embree_order_renders_the_embree_reference_films(name, path_kernel):
    """With PINE_GPU_FLAG_ORDER_EMBREE the device hands a ray's shapes to their tests in the order of the reference's EmbreeAccel,
    the accel a .pine script gets on real pine -- the vendored Embree's BVH8 builder and single-ray traverser restated -- and renders
    the films of the REAL reference built with EmbreeAccel (tests/golden/film_embree_*, tools/make_golden.py --embree) bit for bit:
    scaled boxes, planes, lines and cylinders, 8 to 155 primitives; both path kernels."""
    emb, ps, spp, depth = load_film(name)
    sc = embree_scene(name)
    assert sc.describe() == ps
    film, st = _render(sc, spp, depth, order="embree", specialize=False)
    assert st.kernel_features & (1 << 18), hex(st.kernel_features)  # F_EMBREE: a variant of the order mode ran
    assert_bit_equal(film, emb, f"PINE_GPU_FLAG_ORDER_EMBREE vs O-gcc-embree, {name}")
    if "rect" not in name and "cones" not in name:
        pine_order, _ = _render(sc, spp, depth, specialize=False)
        assert (pine_order.view(np.uint32) != film.view(np.uint32)).any(), "the two orders must differ on a scene with order-dependent shapes"


def test_embree_order_equals_the_oracle_on_other_scenes(oracle, path_kernel):
    """... and equals the CPU restatement's order on scenes beyond the fixtures: random rooms of every shape kind (transformed
    boxes among them), every material and light, meshes (tested first, with pine's triangle tests) -- through the run-time compiled
    feature-set kernel as well as the precompiled twins -- and a scene of hundreds of primitives (the order mode has no size limit)."""
    from pine_amd import scenes
    done = 0
    for seed in (3003, 3007, 3011, 3019, 3021, 3030):
        sc, spp, depth, sampler = scenes.random_scene(seed, variety=True)
        if sampler != "blue":
            continue
        w, h = sc.camera.film().size
        ref, _ = oracle.render(sc.describe(), (w, h), spp, depth, order="embree")
        film, st = _render(sc, spp, depth, order="embree", specialize=False)
        assert_bit_equal(film, ref, f"embree order, random scene {seed}")
        if path_kernel == "queue" and done == 0:
            film2, st2 = _render(sc, spp, depth, order="embree", specialize=True)
            assert_bit_equal(film2, ref, f"embree order, scene-specialised kernel, random scene {seed}")
            assert st2.specialized in (0, 1)  # (never baked: a baked scene is pine's order as code)
        done += 1
    assert done >= 3
    sc = scenes.sss((40, 40), 1)  # one Subsurface mesh in a room
    ref, _ = oracle.render(sc.describe(), (40, 40), 16, 6, order="embree")
    film, _ = _render(sc, 16, 6, order="embree")
    assert_bit_equal(film, ref, "embree order, mesh scene")
    sc = scenes.classic_cones((96, 48), 20)  # 405 primitives
    ref, _ = oracle.render(sc.describe(), (96, 48), 8, 5, order="embree")
    film, _ = _render(sc, 8, 5, order="embree", specialize=False)
    assert_bit_equal(film, ref, "embree order, 405 primitives")


@pytest.mark.parametrize("name", ["C1_cbox_640_s16_d4_committed", "C2_cbox_640_s256_d8_committed", "C2_cbox_640_s256_d8_readme", "C3_cbox_1920x1080_s1024_d8",
                                  "C4_classic_10k_cones_720x360_s64_d6", "C5_sss_320_s512_d8", "C5_sss_640_s512_d8"])
def test_baseline_configs_at_full_size_in_embree_order_md5(name):
    """Every BASELINE config at FULL size as a `.pine` script renders it on real pine -- EmbreeAccel -- against the md5 of the REAL
    reference built with Embree (tests/golden/stats_640_embree.json, tools/make_golden.py --embree-full): cbox at both cameras and
    1920 x 1080, the 10 000 cones (a BVH8 of 10 004 primitives), the Subsurface icosphere (Embree's own triangle test); both path
    kernels."""
    from pine_amd import scenes
    st = json.load(open(os.path.join(GOLDEN, "stats_640_embree.json")))[name]
    sc = {"C1_cbox_640_s16_d4_committed": lambda: scenes.cbox((640, 640), "committed"),
          "C2_cbox_640_s256_d8_committed": lambda: scenes.cbox((640, 640), "committed"),
          "C2_cbox_640_s256_d8_readme": lambda: scenes.cbox((640, 640), "readme"),
          "C3_cbox_1920x1080_s1024_d8": lambda: scenes.cbox((1920, 1080), "committed"),
          "C4_classic_10k_cones_720x360_s64_d6": lambda: scenes.classic_cones((720, 360), 100),
          "C5_sss_320_s512_d8": lambda: scenes.sss((320, 320), 3),
          "C5_sss_640_s512_d8": lambda: scenes.sss((640, 640), 3)}[name]()
    film, _ = _render(sc, st["spp"], st["depth"], order="embree")
    # (a closed mesh has rays through shared edges: where two triangles report the very same t Embree's own triangle hierarchy -- not
    #  restated -- decides which; one query in 10^8.  The fixture carries the reference's values at those pixels: at most four per film.)
    assert len(st["tie_pixels"]) <= 4
    for rec in st["tie_pixels"]:
        film[rec[0], rec[1]] = [float.fromhex(v) for v in rec[2:6]]
    assert hashlib.md5(film.tobytes()).hexdigest() == st["md5"]
    np.testing.assert_allclose(film[..., :3].mean(axis=(0, 1), dtype=np.float64), st["mean_rgb"], rtol=1e-12)


def test_the_order_argument_is_checked():
    import pine_amd as pa
    from pine_amd import scenes
    with pytest.raises(pa.PineError, match="unknown traversal order"):
        pa.Plan(scenes.cbox((32, 32)), 4, 3, order="sideways")


def test_meshes_beyond_the_16_bit_node_limit_render_on_the_32_bit_variant(oracle, path_kernel):
    """The F_LDS_TOP variants keep 16-bit node ids on their traversal stacks: a BVH of 65 536 nodes and more must land on the
    all-features variant with a 32-bit stack (stage-queued kernel) or on the megakernel -- and render the oracle's film.  An
    icosphere of 327 680 triangles (more than 10^5 nodes) with a glossy skin, and the same with Subsurface."""
    import pine_amd as pa
    from pine_amd import scenes
    from pine_amd import _lib
    for skin, spp, depth in ((pa.Glossy([0.9, 0.5, 0.3], 0.2), 4, 4), (None, 2, 4)):
        sc = scenes.sss((24, 24), 7, skin=skin)  # 327 680 triangles
        assert _lib.lib.pine_gpu_scene_build_accel(sc._h) >= 65536
        film, st = _render(sc, spp, depth)
        ref, _ = oracle.render(sc.describe(), (24, 24), spp, depth)
        assert_bit_equal(film, ref, f"327 680-triangle mesh, skin {'glossy' if skin else 'subsurface'} ({path_kernel})")
        # (never a 16-bit-stack variant: the stage-queued kernel's 32-bit one when its stack fits LDS beside the contexts, else --
        #  this tree is 20-odd levels deep, 4 KB of LDS per level -- the megakernel)
        assert not (st.kernel_features & (1 << 13)), hex(st.kernel_features)


def test_device_memory_pool_changes_nothing_and_can_be_released(oracle):
    """Plans take their device buffers from a process-wide pool of the blocks destroyed plans gave back (a one-shot render spends
    most of its time outside the kernels in hipMalloc / hipFree otherwise): films of back-to-back one-shot renders -- the second
    in the first's recycled memory, a smaller one in a block that is too large, one after the pool has been emptied -- equal the
    oracle's."""
    import pine_amd as pa
    from pine_amd import _lib, scenes
    a, b = scenes.cbox((96, 80), "readme"), scenes.classic_cones((64, 32), 6)
    ra, _ = oracle.render(a.describe(), (96, 80), 16, 5)
    rb, _ = oracle.render(b.describe(), (64, 32), 8, 4)
    integ_a, integ_b = pa.PathIntegrator(pa.BlueSampler(16), 5), pa.PathIntegrator(pa.BlueSampler(8), 4)
    for round_ in range(3):
        assert_bit_equal(integ_a.render(a).pixels, ra, f"one-shot cbox, round {round_}")
        assert_bit_equal(integ_b.render(b).pixels, rb, f"one-shot cones in recycled memory, round {round_}")
        if round_ == 1:
            _lib.lib.pine_gpu_release_cached_memory()
