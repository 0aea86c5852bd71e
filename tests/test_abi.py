"""CPU: the C-ABI library loads, exports every symbol include/pine_gpu.h declares, and its host-side
logic (scene building, host math, BVH build, error behaviour) works without a GPU.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_bit_equal, load_film


def _declared():
    text = open(os.path.join(ROOT, "include", "pine_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pine_gpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pine_amd import _lib
    names = _declared()
    assert len(names) > 35
    for n in names:
        assert hasattr(_lib.lib, n), f"libpine_gpu.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib.pine_gpu_abi_version() == 4


def test_host_math_matches_reference():
    import pine_amd as pa
    ref = np.load(os.path.join(GOLDEN, "host_math.npy"))
    m0 = pa.translate([0.0, 0.0, 0.6]) * pa.rotate_y(0.4) * pa.scale([0.6, 0.6, 0.6])
    m1 = pa.translate([-0.6, 0.0, 1.0]) * pa.rotate_y(-0.4) * pa.scale([0.6, 1.3, 0.6])
    got = []
    for m in (m0, pa.inverse(m0), m1, pa.inverse(m1), pa.look_at([0, 0, 0], [0, 0, 1]), pa.look_at([0, 1, -4], [0, 1, 0]),
              pa.look_at([0, 4, -8], [0, 1, 0]), pa.rotate_x(0.3) * pa.rotate_z(-1.1)):
        got += list(m.s)
    assert_bit_equal(np.float32(got), ref[:128], "mat4 helpers")


def test_scene_describe_equals_golden_pscene():
    """The API mirror + C ABI rebuild exactly the scene text the golden films were rendered from."""
    from pine_amd import scenes
    _, ps, _, _ = load_film("cbox_committed_64_s16_d4")
    assert scenes.cbox((64, 64), "committed").describe() == ps
    _, ps, _, _ = load_film("sss_48_s32_d8")
    assert scenes.sss((48, 48), 1).describe() == ps
    _, ps, _, _ = load_film("classic_cones12_90x45_s32_d6")
    assert scenes.classic_cones((90, 45), 12).describe() == ps
    _, ps, _, _ = load_film("xshapes_48_s16_d5")
    assert scenes.xshapes_zoo((48, 48)).describe() == ps


def test_error_behaviour_mirrors_reference():
    import pine_amd as pa
    s = pa.Scene()
    with pytest.raises(pa.PineError, match="Can't find material"):   # scene.cpp:53
        s.add(pa.Rect([0, 0, 0], [1, 0, 0], [0, 0, 1]), "nope")
    with pytest.raises(pa.PineError, match="degenerated"):           # geometry.cpp:266
        s.add("m", pa.Diffuse([1, 1, 1]))
        s.add(pa.Rect([0, 0, 0], [1, 0, 0], [2, 0, 0]), "m")
    with pytest.raises(pa.PineError, match="positive thickness"):    # geometry.cpp:177
        s.add(pa.Line([0, 0, 0], [1, 0, 0], 0.0), "m")
    with pytest.raises(pa.PineError, match="identical begin and end"):  # geometry.cpp:178
        s.add(pa.Line([1, 2, 3], [1, 2, 3], 0.1), "m")
    with pytest.raises(pa.PineError, match="degenerated normal"):    # geometry.cpp:32
        s.add(pa.Plane([0, 0, 0], [0, 0, 0]), "m")
    with pytest.raises(pa.PineError, match="Cylinder"):              # geometry.h:148-150: no sample/pdf/area
        s.add(pa.Cylinder([0, 0, 0], [0, 1, 0], 0.1), pa.Emissive([1, 1, 1]))
    with pytest.raises(pa.PineError, match="max_path_length"):       # path.cpp:12-13
        pa.PathIntegrator(pa.BlueSampler(4), 0)
    with pytest.raises(pa.PineError):
        pa.BlueSampler(0)
    assert pa.BlueSampler(1000).spp() == 256 and pa.BlueSampler(5).spp() == 8


def test_material_name_shadowing_and_light_list():
    import pine_amd as pa
    s = pa.Scene()
    s.add("a", pa.Diffuse([1, 0, 0]))
    s.add("a", pa.Diffuse([0, 1, 0]))  # map assignment: the later one wins
    s.add(pa.Rect([0, 0, 0], [1, 0, 0], [0, 0, 1]), "a")
    s.add(pa.Rect([0, 1, 0], [1, 0, 0], [0, 0, 1]), pa.Emissive([1, 1, 1]))
    d = s.describe()
    assert d.count("material a diffuse") == 2 and "shape rect a" in d and "emissive" in d


def test_bvh_build_invariants():
    """Host BVH build (no GPU): every primitive in exactly one leaf, child boxes inside the parent's."""
    from pine_amd import scenes, _lib
    sc = scenes.classic_cones((64, 32), 16)
    n_nodes = _lib.check(_lib.lib.pine_gpu_scene_build_accel(sc._h))
    assert n_nodes > 50
    nodes = np.zeros((n_nodes, 16), np.float32)
    prims = np.zeros(4096, np.int32)
    n_prims = _lib.lib.pine_gpu_scene_accel_dump(sc._h, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes,
                                                 prims.ctypes.data_as(C.POINTER(C.c_int32)), prims.size)
    n_geoms = 16 * 16 + 5
    assert n_prims == n_geoms
    assert sorted(prims[:n_prims].tolist()) == list(range(n_geoms))
    ints = nodes.view(np.int32)
    seen = np.zeros(n_prims, bool)
    for nd in range(n_nodes):
        for c in range(2):
            child, count = ints[nd, 12 + c], ints[nd, 14 + c]
            if count > 0:
                assert not seen[child:child + count].any()
                seen[child:child + count] = True
            else:
                lo, hi = nodes[nd, 6 * c:6 * c + 3], nodes[nd, 6 * c + 3:6 * c + 6]
                clo = np.minimum(nodes[child, 0:3], nodes[child, 6:9])
                chi = np.maximum(nodes[child, 3:6], nodes[child, 9:12])
                assert (clo >= lo).all() and (chi <= hi).all()
    assert seen.all()


def test_shard_mapping_is_a_partition():
    from pine_amd import _lib
    w, h, world = 45, 37, 3
    owner = np.array([[_lib.lib.pine_gpu_shard_of_pixel(w, x, y, world) for x in range(w)] for y in range(h)])
    tiles_x = (w + 7) // 8
    ys, xs = np.mgrid[0:h, 0:w]
    assert np.array_equal(owner, ((ys // 8) * tiles_x + xs // 8) % world)
    assert set(np.unique(owner)) == {0, 1, 2}


def test_film_finalize_u8():
    import pine_amd as pa
    f = pa.Film([4, 2], pa.Uncharted2())
    f.pixels[...] = 0.0
    f.pixels[0, 0, :3] = 11.2 / 2  # maps to the white point: uncharted2(11.2)/uncharted2(11.2) = 1
    out = f.finalize_u8()
    assert out.shape == (2, 4, 4) and (out[..., 3] == 255).all()
    assert tuple(out[1, 0, :3]) == (255, 255, 255)  # row 0 of the film is the bottom image row (y flip)
    assert tuple(out[0, 1, :3]) == (0, 0, 0)


def _decode_png_rgba8(path):
    """Minimal PNG reader for the files this repo writes (RGBA8, filter type 0 on every row)."""
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(b):
        n, tag = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        if tag == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", data[:10])
            assert (depth, colour) == (8, 6)
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 4 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.parametrize("tag", ["hdr", "cbox"])
@pytest.mark.parametrize("tm", ["uncharted2", "aces"])
def test_film_finalize_equals_the_reference_byte_for_byte(tag, tm, tmp_path):
    """pine_gpu_film_finalize_u8 against what the REAL reference's film.save() produces
    (Film::finalize film.cpp:19-25, tone mappers color.cpp:6-23, y flip fileio.h:33-38, gamma + x256 clamp
    fileio.cpp:42-54; tests/golden/finalize.npz is made by `pine_ref finalize`, whose PNG decodes to the same
    bytes): an HDR film with values far above 1, exact zeros, denormal-small values, w != 1 and a ragged size,
    and a rendered cbox film.  Then the PNG round trip of this repo's two writers."""
    import subprocess
    import pine_amd as pa
    d = np.load(os.path.join(ROOT, "tests", "golden", "finalize.npz"))
    film = d[tag + "_film"]
    h, w = film.shape[:2]
    f = pa.Film([w, h], pa.ACES() if tm == "aces" else pa.Uncharted2())
    f.pixels = film.copy()
    out = f.finalize_u8()
    want = d[f"{tag}_{tm}_u8"]
    assert out.shape == want.shape and np.array_equal(out, want)
    path = str(tmp_path / "a.png")
    f.save(path)
    assert np.array_equal(_decode_png_rgba8(path), want)
    # the C++ writer (pine_amd/host/png_writer.hpp) through a tiny program
    src = tmp_path / "w.cpp"
    src.write_text('#include "png_writer.hpp"\n#include <cstdio>\n#include <vector>\nint main(int c, char** v) { int w = atoi(v[2]), h = atoi(v[3]); '
                   'std::vector<unsigned char> p(size_t(w) * h * 4); FILE* f = fopen(v[1], "rb"); if (!f || fread(p.data(), 1, p.size(), f) != p.size()) return 2; '
                   'fclose(f); return png_writer::write_rgba8(v[4], w, h, p.data()) ? 0 : 1; }\n')
    exe = tmp_path / "w"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "pine_amd", "host"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want.tofile(tmp_path / "in.u8")
    assert subprocess.run([str(exe), str(tmp_path / "in.u8"), str(w), str(h), str(tmp_path / "b.png")]).returncode == 0
    assert np.array_equal(_decode_png_rgba8(str(tmp_path / "b.png")), want)


def test_no_gpu_means_loud_failure():
    """The product never falls back to a CPU path: without a device, rendering raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pine_amd as pa
    from pine_amd import scenes
    sc = scenes.cbox((16, 16))
    with pytest.raises(pa.PineError, match="no HIP device|hip"):
        pa.PathIntegrator(pa.BlueSampler(4), 4).render(sc)
    with pytest.raises(pa.PineError):
        pa.Plan(sc, 4, 4)


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pine_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path only: on a host without a GPU it stops with a message, it never falls
    back to timing the CPU oracle as if it were the product."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())  # no metric line


def test_bench_reference_baseline_reports_the_real_binary(oracle):
    """cpu_baseline kind "reference": oracle/_ref/pine_ref (when the build container made it) renders the scene
    description the product builds, through the code path bench.py uses -- here at 16 spp, against the port."""
    import hashlib
    import importlib.util
    exe = os.path.join(ROOT, "oracle", "_ref", "pine_ref")
    if not os.access(exe, os.X_OK):
        pytest.skip("oracle/_ref/pine_ref not built (no /root/reference here)")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from pine_amd import scenes
    sc = scenes.cbox((640, 640), "committed")
    port, _ = oracle.render(sc.describe(), (640, 640), 16, 8)
    md5 = hashlib.md5(port.tobytes()).hexdigest()
    out = bench.cpu_baseline_reference(sc, (640, 640), 16, 8, gpu_md5=md5, budget_s=1e9)
    assert out and out["kind"] == "reference" and out["cores"] >= 1 and out["value"] > 0
    assert out["film_md5"] == md5 and out["film_equals_gpu"] is True


def test_partition_formulations_agree():
    """The reference's Lomuto partition (src/psl/algorithm.h:394-402) decides the primitive order of every BVH leaf.  The
    device build does not run its swap loop: it computes the loop's RESULT with a prefix sum and pointer jumping
    (pine_bvh_build_device.h).  Both formulations, on random predicates of every density, all-true / all-false / single
    element / alternating inputs: identical permutations."""
    from pine_amd import _lib
    if os.environ.get("PINE_SANITIZER_RUN") == "1":
        pytest.skip("the hook lives in the kernels' translation unit")
    rng = np.random.default_rng(5)
    cases = [np.zeros(1, np.uint8), np.ones(1, np.uint8), np.zeros(7, np.uint8), np.ones(9, np.uint8), np.uint8([0, 1] * 50), np.uint8([1, 0] * 50)]
    cases += [(rng.random(int(rng.integers(2, 3000))) < rng.random()).astype(np.uint8) for _ in range(400)]
    cases += [(rng.random(200_000) < 0.37).astype(np.uint8)]
    for p in cases:
        a, b = np.zeros(len(p), np.int32), np.zeros(len(p), np.int32)
        left = _lib.lib.pine_gpu_test_lomuto(p.ctypes.data_as(C.POINTER(C.c_uint8)), len(p), a.ctypes.data_as(C.POINTER(C.c_int)), b.ctypes.data_as(C.POINTER(C.c_int)))
        assert left == int(p.sum()), _lib.last_error()
        assert np.array_equal(a, b) and sorted(a.tolist()) == list(range(len(p)))
        assert p[a[:left]].all() and not p[a[left:]].any() and np.array_equal(a[:left], np.sort(a[:left]))  # trues first, in order
