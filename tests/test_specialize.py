"""Scene-specialised path kernels (PINE_GPU_FLAG_SPECIALIZE, pine_amd/csrc/pine_specialize.h): the kernel compiled for one
scene -- its BVH unrolled in pine's visiting order (bvh.cpp:405-446), boxes and primitive records as immediates -- must
render the film of the precompiled kernel, which is the reference's, bit for bit.
CPU: the generator's text (what qualifies, what is in it) and the compile step (hipcc cross-compiles; the kernel's symbol is
in the code object; the cache answers the second request).  GPU: films."""
import ctypes as C
import os
import shutil
import tempfile

import numpy as np
import pytest

from conftest import assert_bit_equal, load_film


def _source(scene):
    from pine_amd import _lib
    n = _lib.lib.pine_gpu_scene_specialized_source(scene._h, None, 0)
    assert n >= 0, _lib.last_error()
    if n == 0:
        return ""
    buf = C.create_string_buffer(n + 1)
    assert _lib.lib.pine_gpu_scene_specialized_source(scene._h, buf, n + 1) == n
    return buf.value.decode()


def test_generated_text_holds_the_scene():
    from pine_amd import scenes
    sc = scenes.cbox((64, 64), "readme")
    text = _source(sc)
    assert "scene_traverse_baked" in text
    # one record per primitive that goes through the generic tests (here: the two transformed boxes), each float an exact
    # hexadecimal literal of the device record; the axis-aligned Rects are inline code over their non-zero components
    from pine_amd import _lib
    nshapes = len(sc.describe().split("\nshape ")) - 1
    recs = 0
    for g in range(nshapes):
        rec = (C.c_float * 32)()
        _lib.check(_lib.lib.pine_gpu_scene_shape_record(sc._h, g, rec))
        line = [l for l in text.splitlines() if l.startswith(f"__device__ static constexpr float kBakedRec{g}[30]")]
        if not line:
            continue
        vals = [float.fromhex(t.rstrip("f")) for t in line[0][line[0].index("{") + 1:line[0].index("}")].split(", ")]
        assert np.array_equal(np.float32(vals).view(np.uint32), np.frombuffer(rec, dtype=np.uint32)[:30]), g
        recs += 1
    assert recs == 2 and text.count("const float denom = ") == nshapes - 2
    # every leaf primitive is tested exactly once per visit of its leaf, in stored order: the packed words assigned in the
    # text, in order of appearance, are the BVH's primitive order
    prims = np.zeros(64, dtype=np.int32)
    nodes = np.zeros(64 * 16, dtype=np.float32)
    n = _lib.lib.pine_gpu_scene_accel_dump(sc._h, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes, prims.ctypes.data_as(C.POINTER(C.c_int32)), 64)
    order = [int(l.split("geom_out = ")[1].split(";")[0]) & 0x03ffffff for l in text.splitlines() if "geom_out = " in l]
    assert order == prims[:n].tolist()


def test_literals_are_exact_for_denormals_signed_zeros_and_huge_values(tmp_path, monkeypatch):
    """Record floats are written from their bits (not with printf's locale-dependent %a): denormals, -0 and values next to
    FLT_MAX come back bit for bit, and the compiler accepts them.  (Bounds this large used to crash the host BVH build: the
    bucket index of an overflowing centroid, bvh.cpp:66-68, was undefined behaviour.)"""
    import pine_amd as pa
    from pine_amd import _lib
    s = pa.Scene()
    s.add("w", pa.Diffuse([0.7, 0.7, 0.7]))
    s.add(pa.Sphere([1e-40, -0.0, 3.4e38], 1.17549435e-38), "w")
    s.add(pa.Rect([0.0, 1.9, 1], [0.5, 0, 0], [0, 0, 0.5]), pa.Emissive([20.0, 18.0, 15.0]))
    s.set(pa.ThinLenCamera(pa.Film([16, 16]), [0, 1, -4], [0, 1, 0], 0.25))
    text = _source(s)
    line = next(l for l in text.splitlines() if l.startswith("__device__ static constexpr float kBakedRec0[30]"))
    vals = [float.fromhex(t.rstrip("f")) for t in line[line.index("{") + 1:line.index("}")].split(", ")]
    rec = (C.c_float * 32)()
    _lib.check(_lib.lib.pine_gpu_scene_shape_record(s._h, 0, rec))
    assert np.array_equal(np.float32(vals).view(np.uint32), np.frombuffer(rec, dtype=np.uint32)[:30])
    assert "0x0.022d84p-126f" in line and "-0x0p+0f" in line
    if os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc"):
        monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))
        out = C.create_string_buffer(1024)
        assert _lib.lib.pine_gpu_test_specialize_compile(s._h, 0x104, 1536, b"gfx950", out, 1024) == 0, _lib.last_error()


def test_scenes_that_do_not_qualify_generate_nothing():
    import pine_amd as pa
    from pine_amd import scenes
    assert _source(_two_mesh_scene()) == ""  # two meshes
    top = _source(scenes.sss((32, 32), 1))  # ONE mesh: its top level only (for the traversal-stage variants)
    assert "scene_traverse_baked_top" in top and "#define PINE_BAKED_TOP 1" in top and top.count("the mesh: traversed by the flat machine") == 1
    assert _source(scenes.classic_cones((90, 45), 200)) == ""  # too many primitives to unroll
    s = pa.Scene()
    s.add(pa.Plane([0, 0, 0], [0, 1, 0]), pa.Diffuse([0.5, 0.5, 0.5]))  # fine: finite records
    s.add(pa.Rect([0.0, 1.9, 1], [0.5, 0, 0], [0, 0, 0.5]), pa.Emissive([20.0, 18.0, 15.0]))
    s.set(pa.ThinLenCamera(pa.Film([16, 16]), [0, 1, -4], [0, 1, 0], 0.25))
    assert "scene_traverse_baked" in _source(s)


def test_specialised_kernel_compiles_for_gfx950_and_is_cached(tmp_path, monkeypatch):
    from pine_amd import _lib, scenes
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path / "cache"))
    sc = scenes.cbox((64, 64), "readme")
    out = C.create_string_buffer(1024)
    r = _lib.lib.pine_gpu_test_specialize_compile(sc._h, 258, 1536, b"gfx950", out, 1024)  # (pine_variants.h order 0: cbox's variant)
    assert r == 0, _lib.last_error()
    blob = open(out.value.decode(), "rb").read()
    assert b"pine_scene_kernel_131330_1536" in blob  # (the extern "C" entry: features | F_BAKED, contexts)
    assert _lib.lib.pine_gpu_test_specialize_compile(sc._h, 258, 1536, b"gfx950", out, 1024) == 1  # cache hit
    assert os.listdir(tmp_path / "cache") == [os.path.basename(out.value.decode())]  # (the build directory is gone)
    # the key is the GEOMETRY (BVH + records): the same room under another camera and lamp colour is the same kernel ...
    out2 = C.create_string_buffer(1024)
    same_room, other_room = scenes.cbox((48, 32), "committed"), scenes.cbox((64, 64), "readme", False)
    assert _lib.lib.pine_gpu_test_specialize_compile(same_room._h, 258, 1536, b"gfx950", out2, 1024) == 1
    assert out2.value == out.value
    # ... and another scene is another kernel
    assert _lib.lib.pine_gpu_test_specialize_compile(other_room._h, 258, 1536, b"gfx950", out2, 1024) == 0
    assert out2.value != out.value
    # no compiler: an error that says so, not a silent fallback
    monkeypatch.setenv("PINE_GPU_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path / "cache2"))
    assert _lib.lib.pine_gpu_test_specialize_compile(sc._h, 258, 1536, b"gfx950", out, 1024) < 0
    assert "hipcc" in _lib.last_error()


@pytest.mark.parametrize("features, ctx", [(0x100, 1536), (0x6080, 1024), (0x3a14, 1024), (0xe020, 1024)])
def test_feature_sets_without_a_precompiled_variant_compile(tmp_path, monkeypatch, features, ctx):
    """Level 1 compiles `need | layout` for whatever a scene contains (Rects only; analytic Subsurface shapes without a mesh;
    sphere + cone + node programs + other shapes + Sobol; a bare mesh with traversal stages): the kernel template must
    instantiate for sets no precompiled variant has.  tools/compile_sweep.py walks many more."""
    from pine_amd import _lib
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))
    out = C.create_string_buffer(1024)
    assert _lib.lib.pine_gpu_test_specialize_compile(None, features, ctx, b"gfx950", out, 1024) == 0, _lib.last_error()[-1500:]
    assert f"pine_scene_kernel_{features}_{ctx}".encode() in open(out.value.decode(), "rb").read()  # (the extern "C" entry)


def test_a_packaged_kernel_is_found_before_any_compiler(monkeypatch, tmp_path):
    """A deployment may ship code objects in pine_amd/lib/kernel_cache/ (a read-only install, a box without hipcc): an entry
    with the right content key is found there before a compiler or the user's cache is looked for.  (build() ships none:
    the benchmark's kernels are compiled on the box that runs it.)"""
    from pine_amd import _lib, scenes
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    cache = os.path.join(os.path.dirname(_lib.LIB_PATH), "kernel_cache")
    sc = scenes.cbox((64, 64), "readme")
    out = C.create_string_buffer(1024)
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))
    assert _lib.lib.pine_gpu_test_specialize_compile(sc._h, 258, 1536, b"gfx950", out, 1024) == 0, _lib.last_error()
    built = out.value.decode()
    os.makedirs(cache, exist_ok=True)
    packaged = os.path.join(cache, os.path.basename(built))
    shutil.copy(built, packaged)
    try:
        monkeypatch.delenv("PINE_GPU_CACHE_DIR", raising=False)
        monkeypatch.setenv("HOME", str(tmp_path / "home"))  # (an empty user cache)
        monkeypatch.delenv("XDG_CACHE_HOME", raising=False)
        monkeypatch.setenv("PINE_GPU_HIPCC", "/nonexistent/hipcc")  # (not needed: must not be looked for)
        assert _lib.lib.pine_gpu_test_specialize_compile(sc._h, 258, 1536, b"gfx950", out, 1024) == 1, _lib.last_error()
        assert os.path.realpath(out.value.decode()) == os.path.realpath(packaged)
    finally:
        os.unlink(packaged)
        if not os.listdir(cache):
            os.rmdir(cache)


def test_cache_directory_must_be_private(monkeypatch, tmp_path):
    """A code object found in the cache is loaded and launched: a directory that others can write to is not used."""
    from pine_amd import _lib, scenes
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    if os.geteuid() == 0:
        pass  # (root passes access() everywhere; the ownership / mode test still applies)
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(shared))
    out = C.create_string_buffer(1024)
    sc = scenes.cbox((64, 64), "readme")
    r = _lib.lib.pine_gpu_test_specialize_compile(sc._h, 258, 1536, b"gfx950", out, 1024)
    assert r in (0, 1), _lib.last_error()
    assert os.path.dirname(out.value.decode()) != str(shared) and not os.listdir(shared)
    st = os.lstat(os.path.dirname(out.value.decode()))
    assert st.st_uid == os.geteuid() and not (st.st_mode & 0o022)


def test_generated_unit_checks_the_headers_it_is_compiled_from(monkeypatch, tmp_path):
    """The run-time compile static_asserts sizeof / offsetof fingerprints of the kernel-argument and counter structures
    against the library's own: a kernel built from other headers (here: a flag that grows `Counters`) is refused at compile
    time instead of running with another argument layout."""
    from pine_amd import _lib, scenes
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("PINE_GPU_SPECIALIZE_EXTRA", "-DPINE_PROFILE_SECTIONS")
    out = C.create_string_buffer(1024)
    sc = scenes.cbox((64, 64), "readme")
    assert _lib.lib.pine_gpu_test_specialize_compile(sc._h, 258, 1536, b"gfx950", out, 1024) < 0
    assert "not the ones libpine_gpu.so was built from" in _lib.last_error()


# ---- GPU ------------------------------------------------------------------------------------------------------------
def _render(scene, spp, depth, **kw):
    import torch
    import pine_amd as pa
    w, h = scene.camera.film().size
    plan = pa.Plan(scene, spp, depth, **kw)
    film = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    plan.check()
    st = plan.stats()
    out = film.cpu().numpy()
    plan.close()
    return out, st


@pytest.mark.gpu
@pytest.mark.parametrize("name, look, boxes", [("cbox_readme_64_s16_d4", "readme", True), ("cbox_committed_64_s16_d4", "committed", True),
                                               ("cbox_readme_64_s256_d8", "readme", True), ("cbox_rect_readme_64_s64_d5", "readme", False)])
def test_specialised_cbox_equals_the_reference_film(name, look, boxes):
    from pine_amd import scenes
    ref, ps, spp, depth = load_film(name)
    sc = scenes.cbox((64, 64), look, boxes)
    assert sc.describe() == ps
    film, st = _render(sc, spp, depth, specialize=True)
    assert st.specialized == 2 and st.specialize_ms > 0  # (2: feature set + baked scene)
    assert_bit_equal(film, ref, f"specialised kernel vs the reference's film {name}")


def _adversarial_rays(rng, lo, hi, n):
    """Rays that probe what the axis-aligned Rect code relies on: zero direction components (signed), axis-parallel and
    diagonal directions, origins exactly on the walls' planes and edges, grazing directions with tiny components (huge or
    overflowing t), short and zero-length spans, and plain random ones."""
    lo, hi = np.float32(lo), np.float32(hi)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = n // 8
    grid = np.float32([lo[0], hi[0], 0.0, lo[1], hi[1], 1.0, lo[2], hi[2], 0.5, -0.0])
    o[:2 * k] = rng.choice(grid, (2 * k, 3))                       # origins on planes, edges, corners
    o[2 * k:3 * k, rng.integers(0, 3)] = rng.choice(grid, k)       # ... one coordinate on a plane
    d[k:4 * k] = np.where(rng.random((3 * k, 3)) < 0.45, np.float32(0.0), d[k:4 * k])     # zero components
    d[k:2 * k] = np.where(rng.random((k, 3)) < 0.3, np.float32(-0.0), d[k:2 * k])          # ... negative zeros
    d[4 * k:5 * k] *= np.float32(10.0) ** rng.integers(-44, -20, (k, 3)).astype(np.float32)  # tiny / denormal components
    d[5 * k:6 * k] = rng.choice(np.float32([-1, 0, 1]), (k, 3))    # axes and diagonals
    norm = np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)
    d = np.where(norm > 0, d / np.maximum(norm, 1e-300), d).astype(np.float32)
    d[4 * k:5 * k, 0] = np.float32(1e-42)                          # (a denormal after normalisation, too)
    tmin = np.zeros(n, dtype=np.float32)
    tmax = np.full(n, np.float32(3.0e38))
    tmax[6 * k:7 * k] = rng.uniform(0, 3, k).astype(np.float32)
    tmax[7 * k:7 * k + k // 2] = 0.0
    tmin[7 * k + k // 2:] = rng.uniform(0, 0.5, n - 7 * k - k // 2).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([o, d, tmin[:, None], tmax[:, None]], axis=1), dtype=np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["cbox", "cbox_rects", "axis_room"])
def test_baked_traversal_equals_the_generic_one_ray_by_ray(which):
    """The traversal compiled into a specialised kernel against the generic nested traversal (itself pinned against the
    reference's BVH, tests/test_bvh_fixtures.py): same hit, same geometry, same t bits, same any-hit answer for every ray."""
    import pine_amd as pa
    from pine_amd import _lib, scenes
    if which == "axis_room":
        # every kind of axis-aligned Rect: each normal direction, flipped and not, both edge orders, negative edges, non-unit lengths
        sc = pa.Scene()
        sc.add("w", pa.Diffuse([0.7, 0.7, 0.7]))
        for pos, ex, ey, flip in (([0, 0, 1], [2, 0, 0], [0, 0, 2], True), ([0, 2, 1], [0, 0, -2], [3, 0, 0], False), ([-1, 1, 1], [0, 0, 2], [0, -2, 0], True),
                                  ([1, 1, 1], [0, 0.5, 0], [0, 0, 2], False), ([0, 1, 2], [2, 0, 0], [0, 2, 0], True), ([0.25, 0.5, 0.75], [0, 0, -0.3], [0.7, 0, 0], False),
                                  ([0.3, 1.0, 1.2], [0.4, 0.1, 0], [0, 0, 0.5], False)):  # (the last one is NOT axis-aligned: generic path)
            sc.add(pa.Rect(pos, ex, ey, flip), "w")
        sc.add(pa.Rect([0.0, 1.9, 1], [0.5, 0, 0], [0, 0, 0.5]), pa.Emissive([20.0, 18.0, 15.0]))
        sc.set(pa.ThinLenCamera(pa.Film([16, 16]), [0, 1, -4], [0, 1, 0], 0.25))
    else:
        sc = scenes.cbox((16, 16), "readme", which == "cbox")
    rays = _adversarial_rays(np.random.default_rng(5), [-1, 0, 0], [1, 2, 2], 40000)
    cap = 40
    gen = np.zeros((len(rays), 2 * cap + 5), dtype=np.uint32)
    _lib.check(_lib.lib.pine_gpu_test_traverse(sc._h, 0, rays.ctypes.data_as(_lib.c_f_p), len(rays), 0, cap, gen.ctypes.data_as(C.POINTER(C.c_uint32))))
    plan = pa.Plan(sc, 1, 1, specialize=True)
    assert plan.stats().specialized == 2
    baked = np.zeros((len(rays), 4), dtype=np.uint32)
    _lib.check(_lib.lib.pine_gpu_plan_test_traverse_baked(plan._h, rays.ctypes.data_as(_lib.c_f_p), len(rays), baked.ctypes.data_as(C.POINTER(C.c_uint32))))
    plan.close()
    want = np.stack([gen[:, cap], gen[:, cap + 1], gen[:, cap + 3], gen[:, 2 * cap + 4]], axis=1)
    bad = np.nonzero((want != baked).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], rays[bad[:5]], want[bad[:5]], baked[bad[:5]])
    assert 0.2 < want[:, 0].mean() < 0.99 and want[:, 3].mean() > 0.1  # (the rays do hit things, and miss)


@pytest.mark.gpu
def test_specialised_random_scenes_equal_the_precompiled_kernels():
    """Seeded random scenes (every analytic shape, every material and light kind, meshes, BVHs with one and with two inner
    children).  Level 2 (at most 10 primitives, no mesh): feature set + baked scene.  Level 1 (everything else): the scene's
    exact feature set with the generic traversal.  Either way the film equals the precompiled kernel's, bit for bit."""
    from pine_amd import scenes
    done = {1: 0, 2: 0}
    for seed in range(3000, 3040):
        sc, spp, depth, sampler = scenes.random_scene(seed, variety=True)
        level = 2 if _source(sc) else 1
        if done[level] >= 5:
            continue
        a, st = _render(sc, spp, depth, sampler=sampler, specialize=True)
        b, st0 = _render(sc, spp, depth, sampler=sampler, specialize=False)
        assert st0.specialized == 0
        assert_bit_equal(a, b, f"random scene {seed}: specialised (level {level}) vs precompiled")
        assert st.specialized in (0, level)  # (0: the precompiled variant already is the scene's feature set)
        if st.specialized:
            assert st.kernel_features & ~st0.kernel_features & 0xffff == 0 and (level == 2 or st.kernel_features != st0.kernel_features)
            done[level] += 1
        if done[1] >= 5 and done[2] >= 5:
            break
    assert done[1] >= 5 and done[2] >= 5, done


def _two_mesh_scene():
    import pine_amd as pa
    from pine_amd import scenes
    sc = pa.Scene()
    sc.add("w", pa.Diffuse([0.8, 0.8, 0.8]))
    sc.add("skin", pa.Subsurface([0.9, 0.8, 0.7], 0.2, [20.0, 30.0, 40.0]))
    sc.add(pa.Rect([0, 0, 1], [2, 0, 0], [0, 0, 2], True), "w")
    for x in (-0.4, 0.4):
        v, f = scenes.icosphere(1, 0.3, (x, 0.5, 1.0))
        sc.add(pa.Mesh(v, f), "skin")
    sc.add(pa.Rect([0.0, 1.9, 1], [0.5, 0, 0], [0, 0, 0.5]), pa.Emissive([20.0, 18.0, 15.0]))
    sc.set(pa.ThinLenCamera(pa.Film([32, 32]), [0, 1, -4], [0, 1, 0], 0.25))
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("level, skin", [(1, "sss"), (2, "sss"), (2, "glossy_emissive"), (3, "sss")])
def test_one_mesh_scenes_bake_their_top_level(oracle, level, skin):
    """A scene with ONE mesh under a small top level (BASELINE's Subsurface icosphere): the top-level BVH becomes code run once
    per ray, the flat traversal keeps the mesh.  Primitives stored after the mesh in pine's order are
    settled by replaying the top level with the mesh's hit in its place: the film is the oracle's, bit for bit -- closed
    rooms (every ray that hits the mesh also hit a wall behind it), walk stage and sample tokens included; a scene with a
    second mesh is left to the flat traversal whole."""
    import pine_amd as pa
    from pine_amd import scenes
    kw = {} if skin == "sss" else dict(skin=pa.Glossy([0.9, 0.5, 0.3], 0.15), emissive_mesh=True)
    sc = scenes.sss((40, 40), level, **kw)
    a, st = _render(sc, 16, 6, specialize=True)
    ref, _ = oracle.render(sc.describe(), (40, 40), 16, 6)
    if skin == "sss":  # (ONE mesh; traversal stages chosen: the mesh's BVH fits the LDS cache)
        assert st.specialized == 2 and (st.kernel_features & 0x8000)
    else:              # (the emissive-mesh variant of the scene has a second mesh, the lamp: feature set only)
        assert st.specialized in (0, 1)
    assert_bit_equal(a, ref, f"top level baked, icosphere level {level}, {skin}")


@pytest.mark.gpu
def test_specialise_under_sharding_and_by_environment(monkeypatch):
    import torch
    import pine_amd as pa
    from pine_amd import _lib, scenes
    sc = scenes.cbox((72, 40), "readme")
    whole, _ = _render(sc, 16, 5, specialize=False)
    total = np.zeros_like(whole)
    for rank in range(3):
        part, st = _render(sc, 16, 5, shard_rank=rank, shard_world=3, specialize=True)
        assert st.specialized == 2
        total += part
    assert_bit_equal(total, whole, "specialised shards sum to the whole film")
    monkeypatch.setenv("PINE_GPU_SPECIALIZE", "1")
    f, st = _render(sc, 16, 5)
    assert st.specialized == 2
    assert_bit_equal(f, whole, "PINE_GPU_SPECIALIZE=1")
    monkeypatch.setenv("PINE_GPU_SPECIALIZE", "0")
    f, st = _render(sc, 16, 5, specialize=True)
    assert st.specialized == 0
    monkeypatch.delenv("PINE_GPU_SPECIALIZE")
    # the feature set only (geometry that changes every render): cbox without its boxes is Rects only -> F_OBB dropped
    rects = scenes.cbox((72, 40), "readme", False)
    a, st = _render(rects, 16, 5, flags=_lib.FLAG_SPECIALIZE | _lib.FLAG_SPECIALIZE_NO_BAKE)
    b, _ = _render(rects, 16, 5, specialize=False)
    assert st.specialized == 1 and st.kernel_features & 0xff == 0
    assert_bit_equal(a, b, "feature-set kernel without the baked scene")
    monkeypatch.setenv("PINE_GPU_SPECIALIZE", "0")
    # a scene with nothing to gain (two meshes: nothing to bake; its variant already is its feature set) renders with the
    # precompiled kernel, flag or not
    monkeypatch.delenv("PINE_GPU_SPECIALIZE")
    f, st = _render(_two_mesh_scene(), 8, 4, specialize=True)
    assert st.specialized == 0
    # a Subsurface mesh among other kinds: the everything kernel is replaced by the scene's own feature set (walk stage,
    # sample tokens and traversal stages included)
    sc = scenes.random_scene(4001, variety=2)[0]
    a, st = _render(sc, 16, 6, specialize=True)
    b, st0 = _render(sc, 16, 6, specialize=False)
    assert st0.specialized == 0
    assert_bit_equal(a, b, "exact feature set vs the all-features kernel")


@pytest.mark.gpu
def test_asynchronous_specialisation_swaps_the_kernel_in_between_launches(monkeypatch, tmp_path):
    """PINE_GPU_FLAG_SPECIALIZE_ASYNC: plan creation returns at once, the precompiled kernel renders while hipcc runs on a
    background thread, a later launch adopts the scene's kernel; every film along the way is the same film."""
    import time
    import torch
    import pine_amd as pa
    from pine_amd import _lib, scenes
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))  # (empty: the compiler must run)
    sc = scenes.cbox((48, 48), "readme")
    want, _ = _render(sc, 8, 4, specialize=False)
    t0 = time.perf_counter()
    plan = pa.Plan(sc, 8, 4, flags=_lib.FLAG_SPECIALIZE | _lib.FLAG_SPECIALIZE_ASYNC)
    created = time.perf_counter() - t0
    film = torch.zeros((48, 48, 4), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seen = set()
    deadline = time.perf_counter() + 120
    while time.perf_counter() < deadline:
        film.fill_(-1.0)
        plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        plan.check()
        level = plan.stats().specialized
        seen.add(level)
        assert_bit_equal(film.cpu().numpy(), want, f"film at specialisation level {level}")
        if level == 2 and len(seen) > 1 or level < 0:
            break
        if level == 2:
            break
        time.sleep(0.02)
    plan.close()
    assert 2 in seen, seen
    assert 0 in seen and created < 0.5, (seen, created)  # (the first launches ran the precompiled kernel; creation did not wait)
    # a build that fails: no exception, the precompiled kernel keeps rendering, stats say so
    monkeypatch.setenv("PINE_GPU_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path / "other"))
    plan = pa.Plan(sc, 8, 4, flags=_lib.FLAG_SPECIALIZE | _lib.FLAG_SPECIALIZE_ASYNC)
    level = 0
    deadline = time.perf_counter() + 60
    while level == 0 and time.perf_counter() < deadline:
        plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        level = plan.stats().specialized
        time.sleep(0.02)
    assert level == -1
    assert_bit_equal(film.cpu().numpy(), want, "film after a failed background build")
    plan.close()


@pytest.mark.gpu
def test_a_damaged_cache_entry_is_replaced(monkeypatch, tmp_path):
    import pine_amd as pa
    from pine_amd import scenes
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))
    sc = scenes.cbox((32, 32), "readme")
    a, st = _render(sc, 4, 3, specialize=True)
    (entry,) = os.listdir(tmp_path)
    good = os.path.getsize(tmp_path / entry)
    with open(tmp_path / entry, "r+b") as f:  # cut short, as by a full disk
        f.truncate(good // 3)
    from pine_amd import _lib
    _lib.lib.pine_gpu_release_cached_memory()  # (as a new process would find it: this one still has the kernel loaded)
    b, st = _render(sc, 4, 3, specialize=True)
    assert st.specialized == 2, "the plan did not get its kernel back"
    assert os.listdir(tmp_path) == [entry] and os.path.getsize(tmp_path / entry) > good // 2, (os.listdir(tmp_path), os.path.getsize(tmp_path / entry), good)
    assert_bit_equal(a, b, "after recompiling a damaged cache entry")


@pytest.mark.gpu
def test_specialise_fails_loudly_without_a_compiler(monkeypatch, tmp_path):
    import pine_amd as pa
    from pine_amd import scenes
    monkeypatch.setenv("PINE_GPU_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))
    with pytest.raises(pa.PineError, match="hipcc"):
        pa.Plan(scenes.cbox((32, 32), "readme"), 4, 3, specialize=True)


@pytest.mark.gpu
def test_default_mode_never_waits_never_fails_and_uses_the_cache(monkeypatch, tmp_path):
    """No flag: the scene's kernel comes from the cache when it is there; otherwise the compiler runs in the background while
    the precompiled kernel renders, a later launch adopts the result, and the NEXT plan of the same geometry -- another
    camera, film size, spp -- finds it in the cache at creation.  Without a compiler nothing fails.  Same film throughout."""
    import time
    import torch
    import pine_amd as pa
    from pine_amd import scenes
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))  # (empty: the compiler must run)
    monkeypatch.delenv("PINE_GPU_SPECIALIZE", raising=False)
    sc = scenes.cbox((48, 48), "readme", False)  # (Rect-only room: a geometry no other test of this process has tried to build)
    want, st = _render(sc, 8, 4, specialize=False)
    assert st.specialized == 0 and st.specialize_source == 0 and st.specialize_pending == 0
    t0 = time.perf_counter()
    plan = pa.Plan(sc, 8, 4)
    created = time.perf_counter() - t0
    film = torch.zeros((48, 48, 4), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seen = []
    deadline = time.perf_counter() + 120
    while time.perf_counter() < deadline:
        film.fill_(-1.0)
        plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        plan.check()
        st = plan.stats()
        seen.append((st.specialized, st.specialize_source, st.specialize_pending))
        assert_bit_equal(film.cpu().numpy(), want, f"film at {seen[-1]}")
        if st.specialized != 0 or not st.specialize_pending:
            break
        time.sleep(0.02)
    plan.close()
    assert seen[0] == (0, 3, 1) and created < 0.5, (seen[:2], created)  # creation did not wait; the precompiled kernel rendered first
    assert seen[-1] == (2, 3, 0), seen[-1]
    # the next plan of the same geometry (another camera, size, spp): from the cache, at creation
    other = scenes.cbox((40, 24), "committed", False)
    want2, _ = _render(other, 4, 3, specialize=False)
    got2, st = _render(other, 4, 3)
    assert (st.specialized, st.specialize_source, st.specialize_pending) == (2, 1, 0)
    assert_bit_equal(got2, want2, "kernel from the cache")
    # a damaged cache entry (cut short, as by a full disk): the default mode neither fails nor loads it -- the precompiled kernel
    # renders, the entry is rebuilt in the background
    (entry,) = [e for e in os.listdir(tmp_path) if e.endswith(".co")]
    good = os.path.getsize(tmp_path / entry)
    with open(tmp_path / entry, "r+b") as f:
        f.truncate(good // 3)
    from pine_amd import _lib
    _lib.lib.pine_gpu_release_cached_memory()  # (as a new process would find it: this one still has the kernel loaded)
    plan = pa.Plan(other, 4, 3)
    film2 = torch.zeros((24, 40, 4), device="cuda")
    deadline = time.perf_counter() + 120
    while time.perf_counter() < deadline:
        plan.launch(film2.data_ptr(), stream)
        torch.cuda.synchronize()
        st = plan.stats()
        assert_bit_equal(film2.cpu().numpy(), want2, "film while a damaged cache entry is being rebuilt")
        if st.specialized == 2:
            break
        assert st.specialized == 0 and st.specialize_pending == 1, (st.specialized, st.specialize_pending)
        time.sleep(0.02)
    plan.close()
    assert st.specialized == 2 and os.path.getsize(tmp_path / entry) > good // 2
    # no compiler, nothing cached: the default mode renders with the precompiled kernel and says so; nothing raises
    monkeypatch.setenv("PINE_GPU_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path / "empty"))
    plan = pa.Plan(sc, 8, 4)
    level, deadline = 0, time.perf_counter() + 60
    while time.perf_counter() < deadline:
        plan.launch(film.data_ptr(), stream)
        torch.cuda.synchronize()
        st = plan.stats()
        if not st.specialize_pending:
            break
        time.sleep(0.02)
    assert st.specialized == -1 and st.specialize_pending == 0
    assert_bit_equal(film.cpu().numpy(), want, "film without a compiler")
    plan.close()
    # ... and the one-shot entry point (what PathIntegrator.render calls) likewise
    f = pa.PathIntegrator(pa.BlueSampler(8), 4).render(sc).pixels
    assert_bit_equal(f, want, "one-shot render without a compiler")


@pytest.mark.gpu
def test_a_process_that_exits_while_the_compiler_runs_leaves_nothing_behind(tmp_path):
    """The default mode's compiler child outlives the plan that asked for it, not the process: at exit the library ends the
    child (its own process group), joins its workers and removes the half-made build directory -- no hang, no crash, no litter."""
    import subprocess
    import sys
    import time
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import pine_amd as pa\nfrom pine_amd import scenes\n"
            "sc = scenes.cbox((32, 32), 'readme')\n"
            "plan = pa.Plan(sc, 4, 3)\n"
            "st = plan.stats()\n"
            "print('pending', st.specialize_pending, flush=True)\n") % ROOT
    env = dict(os.environ, PINE_GPU_CACHE_DIR=str(tmp_path))
    env.pop("PINE_GPU_SPECIALIZE", None)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    took = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    assert "pending 1" in r.stdout, r.stdout
    left = os.listdir(tmp_path)
    assert not [e for e in left if e.startswith("build_")], left  # (a finished kernel may be there if the compiler won the race; a build directory may not)
    assert took < 60
