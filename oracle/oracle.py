"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes wrapper of liboracle.so (pine_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("PINE_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # (the override: the sanitizer build, tools/sanitize)
REF_BIN = os.path.join(_HERE, "_ref", "pine_ref")
TABLES = os.path.join(os.path.dirname(_HERE), "pine_amd", "data", "bluesobol_u8.bin")


class Stats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("camera_samples", C.c_uint64), ("vertices", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("bsdf_samples", C.c_uint64), ("threads", C.c_int),
                ("spp_effective", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


_lib = None
_tables = None


def lib():
    global _lib, _tables
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_last_error.restype = C.c_char_p
        _tables = np.fromfile(TABLES, dtype=np.uint8)
        assert _tables.size == 65536 + 9 * 262144
    return _lib


def _tp():
    lib()
    return _tables.ctypes.data_as(C.c_void_p)


def render(pscene: str, size, spp, depth, threads=0, rows=None, sampler="blue", order="pine"):
    """-> (film[h,w,4] float32, Stats).  sampler: "blue" = BlueSampler(spp), "sobol" = SobolSampler(spp), "halton" = HaltonSampler(spp).
    order: "pine" = the reference's BVH order (the parity oracle); "nearest" = nearest bounds first (PINE_GPU_FLAG_ORDER_NEAREST)."""
    w, h = size
    film = np.zeros((h, w, 4), np.float32)
    st = Stats()
    y0, y1 = rows if rows else (0, 0)
    lib().oracle_set_sampler({"sobol": 1, "halton": 2}.get(sampler, 0))
    lib().oracle_set_order({"nearest": 1, "embree": 2}.get(order, 0))
    try:
        rc = lib().oracle_render(pscene.encode(), _tp(), int(spp), int(depth), int(threads), int(y0), int(y1),
                                 film.ctypes.data_as(C.c_void_p), C.byref(st))
    finally:
        lib().oracle_set_sampler(0)
        lib().oracle_set_order(0)
    if rc:
        raise RuntimeError(f"oracle_render: {lib().oracle_last_error().decode()}")
    return film, st


def render_shard(pscene: str, size, spp, depth, rank, world, threads=0):
    w, h = size
    film = np.zeros((h, w, 4), np.float32)
    rc = lib().oracle_render_shard(pscene.encode(), _tp(), int(spp), int(depth), int(threads), int(rank), int(world),
                                   film.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"oracle_render_shard: {lib().oracle_last_error().decode()}")
    return film


def render_samples(pscene: str, size, spp_eff, spp, depth, threads=0):
    w, h = size
    out = np.zeros((h, w, spp_eff, 4), np.float32)
    rc = lib().oracle_render_samples(pscene.encode(), _tp(), int(spp), int(depth), int(threads),
                                     out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"oracle_render_samples: {lib().oracle_last_error().decode()}")
    return out


def effective_spp(spp):
    n, p = min(spp, 256), 1
    while p < n:
        p *= 2
    return p


def sampler_stream(spp):
    n = 6 * effective_spp(spp) * (260 + 270)
    out = np.zeros(n, np.float32)
    assert lib().oracle_sampler_stream(_tp(), int(spp), out.ctypes.data_as(C.c_void_p), C.c_int64(n)) == 0
    return out


def rng_stream():
    out = np.zeros(6 * 19, np.uint64)
    assert lib().oracle_rng_stream(out.ctypes.data_as(C.c_void_p), C.c_int64(out.size)) == 0
    return out


def host_math():
    out = np.zeros(8 * 16 + 5 * 9, np.float32)
    assert lib().oracle_host_math(out.ctypes.data_as(C.c_void_p), C.c_int64(out.size)) == 0
    return out


def shapes(pscene: str, rays: np.ndarray, n_geoms: int):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros((n_geoms, len(rays), 11), np.float32)
    rc = lib().oracle_shapes(pscene.encode(), rays.ctypes.data_as(C.c_void_p), C.c_int64(len(rays)),
                             out.ctypes.data_as(C.c_void_p), C.c_int64(out.size))
    assert rc == 0, rc
    return out


def have_ref():
    return os.path.exists(REF_BIN)


def ref_render(pscene: str, size, spp, depth, workdir="/tmp"):
    """Run the real reference (oracle/_ref/pine_ref). -> (film, json dict)"""
    import json
    w, h = size
    sp = os.path.join(workdir, f"_ref_{os.getpid()}.pscene")
    fp = os.path.join(workdir, f"_ref_{os.getpid()}.film")
    open(sp, "w").write(pscene)
    out = subprocess.run([REF_BIN, "render", sp, str(spp), str(depth), fp], capture_output=True, text=True, check=True)
    film = np.fromfile(fp, dtype=np.float32).reshape(h, w, 4)
    os.remove(sp)
    os.remove(fp)
    return film, json.loads(out.stdout.strip().splitlines()[-1])
