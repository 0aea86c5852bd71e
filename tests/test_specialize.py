"""Scene-specialised path kernels (PINE_GPU_FLAG_SPECIALIZE, pine_amd/csrc/pine_specialize.h): the kernel compiled for one
scene -- its BVH unrolled in pine's visiting order (bvh.cpp:405-446), boxes and primitive records as immediates -- must
render the film of the precompiled kernel, which is the reference's, bit for bit.
CPU: the generator's text (what qualifies, what is in it) and the compile step (hipcc cross-compiles; the kernel's symbol is
in the code object; the cache answers the second request).  GPU: films."""
import ctypes as C
import os
import shutil

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
    # one record per primitive of the scene, each float as an exact hexadecimal literal of the device record
    from pine_amd import _lib
    nshapes = len(sc.describe().split("\nshape ")) - 1
    for g in range(nshapes):
        rec = (C.c_float * 32)()
        _lib.check(_lib.lib.pine_gpu_scene_shape_record(sc._h, g, rec))
        line = next(l for l in text.splitlines() if l.startswith(f"__device__ static constexpr float kBakedRec{g}[30]"))
        vals = [float.fromhex(t.rstrip("f")) for t in line[line.index("{") + 1:line.index("}")].split(", ")]
        assert np.array_equal(np.float32(vals).view(np.uint32), np.frombuffer(rec, dtype=np.uint32)[:30]), g
    # every leaf primitive is tested exactly once per visit of its leaf, in stored order: the order of first appearance in
    # the text is the BVH's primitive order
    prims = np.zeros(64, dtype=np.int32)
    nodes = np.zeros(64 * 16, dtype=np.float32)
    n = _lib.lib.pine_gpu_scene_accel_dump(sc._h, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes, prims.ctypes.data_as(C.POINTER(C.c_int32)), 64)
    order = [int(l.split("kBakedRec")[1].split("[")[0]) for l in text.splitlines() if "rec.f[k] = kBakedRec" in l]
    assert sorted(order) == sorted(prims[:n].tolist()) and len(order) == n


def test_scenes_that_do_not_qualify_generate_nothing():
    import pine_amd as pa
    from pine_amd import scenes
    assert _source(scenes.sss((32, 32), 1)) == ""  # a mesh
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
    assert b"_ZN8pine_gpu17path_queue_kernelILj131330ELi1536EEEvNS_11DeviceSceneENS_10WorkParams" in blob
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
        b, st0 = _render(sc, spp, depth, sampler=sampler)
        assert st0.specialized == 0
        assert_bit_equal(a, b, f"random scene {seed}: specialised (level {level}) vs precompiled")
        assert st.specialized in (0, level)  # (0: the precompiled variant already is the scene's feature set)
        if st.specialized:
            assert st.kernel_features & ~st0.kernel_features & 0xffff == 0 and (level == 2 or st.kernel_features != st0.kernel_features)
            done[level] += 1
        if done[1] >= 5 and done[2] >= 5:
            break
    assert done[1] >= 5 and done[2] >= 5, done


@pytest.mark.gpu
def test_specialise_under_sharding_and_by_environment(monkeypatch):
    import torch
    import pine_amd as pa
    from pine_amd import scenes
    sc = scenes.cbox((72, 40), "readme")
    whole, _ = _render(sc, 16, 5)
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
    # a scene with nothing to gain (a mesh: nothing to bake; its variant already is its feature set) renders with the
    # precompiled kernel, flag or not
    monkeypatch.delenv("PINE_GPU_SPECIALIZE")
    f, st = _render(scenes.sss((32, 32), 1), 8, 4, specialize=True)
    assert st.specialized == 0
    # a Subsurface mesh among other kinds: the everything kernel is replaced by the scene's own feature set (walk stage,
    # sample tokens and traversal stages included)
    sc = scenes.random_scene(4001, variety=2)[0]
    a, st = _render(sc, 16, 6, specialize=True)
    b, st0 = _render(sc, 16, 6)
    assert_bit_equal(a, b, "exact feature set vs the all-features kernel")


@pytest.mark.gpu
def test_specialise_fails_loudly_without_a_compiler(monkeypatch, tmp_path):
    import pine_amd as pa
    from pine_amd import scenes
    monkeypatch.setenv("PINE_GPU_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("PINE_GPU_CACHE_DIR", str(tmp_path))
    with pytest.raises(pa.PineError, match="hipcc"):
        pa.Plan(scenes.cbox((32, 32), "readme"), 4, 3, specialize=True)
