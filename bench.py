#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configs.

metric   : Msamples/s = W*H*spp_effective / t   
workload : --config c2 (default) = configs[1] = cbox 640x640, BlueSampler(256), depth 8, as-committed camera;
           c3 = cbox 1920x1080, BlueSampler(1024 -> 256 effective), depth 8 (BASELINE's 8-GPU workload);
           c4 = classic.pine + 10 000 cones 720x360, BlueSampler(64), depth 6;
           c5 = Subsurface icosphere in the Rect-only cbox 640x640, BlueSampler(512 -> 256 effective), depth 8.
step     : one full render of that film (prepass + path kernel + ordered resolve [+ RCCL slab gather and
           unpack on rank 0 when N > 1]); scene, BVH, sampler tables and work buffers are resident in HBM
           before the timed region.
N > 1    : 8x8-pixel tiles dealt round-robin to ranks (strong scaling: the film is fixed); every rank writes
           only its own tiles into a slab, the slabs are gathered to rank 0 (direct xGMI sends) and scattered
           into the row-major film there.  The gather + unpack of step i runs on a second stream while the path
           kernel of step i+1 runs (double-buffered slabs).
At N = 1 the default line also carries a `configs` array: C2 with the README camera, C3 / C4 /
C5 measured the same way at reduced steps, each with its own roofline entry and its film's md5 checked against
the reference's (tests/golden/stats_640.json); `host_resident` (the same renders with the film's device-to-host
copy inside the clock) and `including_build_and_upload` (one cold render: plan creation = host BVH build + flattening
+ table / scene upload, then the launch and the copy).  At N > 1 the headline stays C2 and `configs` carries C3 --
BASELINE's 8-GPU workload -- sharded over the same ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_VERTEX = 192.0  # algorithmic bytes per radiance() invocation
HBM_PEAK_GBS = 8000.0


def _configs():
    from pine_amd import scenes
    return {
        # name: (scene builder, requested spp, depth, workload text, key in tests/golden/stats_640.json)
        "c2": (lambda: scenes.cbox((640, 640), "committed"), 256, 8,
               "cbox 640x640 256spp depth=8, as-committed camera, BlueSampler, pine-BVH order", "C2_cbox_640_s256_d8_committed"),
        "c2r": (lambda: scenes.cbox((640, 640), "readme"), 256, 8,
                "cbox 640x640 256spp depth=8, README camera [0,1,-4]->[0,1,0] fov 0.25 (README.md:32: the whole room is in view), BlueSampler, pine-BVH order",
                "C2_cbox_640_s256_d8_readme"),
        "c3": (lambda: scenes.cbox((1920, 1080), "committed"), 1024, 8,
               "cbox 1920x1080 BlueSampler(1024) = 256 effective spp depth=8, as-committed camera, pine-BVH order",
               "C3_cbox_1920x1080_s1024_d8"),
        "c4": (lambda: scenes.classic_cones((720, 360), 100), 64, 6,
               "classic.pine + 10 000 procedurally placed cones 720x360 64spp depth=6", "C4_classic_10k_cones_720x360_s64_d6"),
        "c5": (lambda: scenes.sss((640, 640), 3), 512, 8,
               "Subsurface icosphere (1280 triangles) + emissive Rect in the Rect-only cbox 640x640 BlueSampler(512) = 256 effective spp depth=8",
               "C5_sss_640_s512_d8"),
    }


def golden_md5(key):
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "stats_640.json")))[key]["md5"]
    except Exception:
        return None


def cpu_baseline(scene, size, spp, depth, budget_s=12.0):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded,
    stratified sample of the same workload: 8-row bands spread evenly over the film."""
    from oracle import oracle
    W, H = size
    ps = scene.describe()
    threads = os.cpu_count() or 1
    # calibrate on two bands (one in the lower half, one in the upper half)
    t0 = time.time()
    n = 0
    for y in (H // 4 // 8 * 8, 3 * H // 4 // 8 * 8):
        _, st = oracle.render(ps, (W, H), spp, depth, threads, rows=(y, y + 8))
        n += st.camera_samples
    rate = n / max(time.time() - t0, 1e-6)
    total = W * H * min(spp, 256)
    bands_all = H // 8
    nb = int(max(2, min(bands_all, budget_s * rate / (8 * W * min(spp, 256)))))
    nb -= nb % 2  # keep the two halves of the film equally represented
    nb = max(nb, 2)
    step = bands_all / nb
    ys = sorted({int(i * step) * 8 for i in range(nb)})
    samples = 0
    secs = 0.0
    verts = 0
    for y in ys:
        _, st = oracle.render(ps, (W, H), spp, depth, threads, rows=(y, y + 8))
        samples += st.camera_samples
        secs += st.seconds
        verts += st.vertices
    return {
        "value": samples / secs * 1e-6, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{len(ys)} of {bands_all} 8-row bands of the same {W}x{H} render "
                  f"({samples / total:.3f} of the workload, {secs:.1f} s), evenly spaced over the film",
        "vertices_per_sample": verts / samples,
    }


def cpu_baseline_reference(scene, size, spp, depth, gpu_md5, budget_s=40.0):
    """The REAL reference timed beside the GPU: oracle/_ref/pine_ref is pine's own PathIntegrator + BVH,
    compiled from the reference's sources by oracle/Makefile in the build container (the binary travels,
    the sources do not).  Renders the same scene description; the whole workload when a 16-spp
    calibration run says it fits the budget, else the largest power-of-two spp that does.  Returns None when
    the binary is absent or fails (the caller then reports the CPU restatement, kind "port")."""
    import subprocess
    import tempfile
    W, H = size
    spp_full = min(spp, 256)
    exe = os.path.join(ROOT, "oracle", "_ref", "pine_ref")
    if not os.access(exe, os.X_OK):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            sp, fp = os.path.join(tmp, "s.pscene"), os.path.join(tmp, "s.film")
            open(sp, "w").write(scene.describe())

            def run(n, limit):
                r = subprocess.run([exe, "render", sp, str(n), str(depth), fp], capture_output=True, text=True,
                                   timeout=limit)
                if r.returncode != 0:
                    raise RuntimeError(r.stderr[-300:])
                return json.loads(r.stdout.strip().splitlines()[-1])
            cal = run(16, 240)
            n = spp_full
            while n > 16 and cal["seconds"] * n / 16 > budget_s:
                n //= 2
            res = cal if n == 16 else run(n, 4 * budget_s + 60)
            md5 = hashlib.md5(open(fp, "rb").read()).hexdigest()
        whole = n == spp_full
        what = "oracle/_ref/pine_ref = the reference's own PathIntegrator(BVH, BlueSampler, UniformLightSampler)"
        return {
            "value": res["msamples_per_s"], "unit": "Msamples/s", "cores": res["threads"], "kind": "reference",
            "sample": (f"the whole workload ({W}x{H}x{n}spp depth {depth}) rendered by {what} in {res['seconds']:.1f} s" if whole else
                       f"{W}x{H}x{n}spp depth {depth} ({n}/{spp_full} of the samples per pixel, {res['seconds']:.1f} s) rendered by {what}"),
            "film_md5": md5,
            "film_equals_gpu": (md5 == gpu_md5) if whole else None,
        }
    except Exception as e:  # the reference binary is optional test infrastructure: report, do not fail the bench
        print(f"[bench] reference CPU baseline unavailable: {e}", file=sys.stderr)
        return None


def pmc_figures(name):
    """VALU figures of the path kernel from the latest committed PMC summary of this config (profiles/rNN_[cX_]pmc_summary.txt,
    written by tools/profile_round.sh on the GPU box) -- collected off-line with rocprofv3 --pmc, NOT in this run."""
    pat = "r*_pmc_summary.txt" if name in ("c2", "c3") else f"r*_{name}_pmc_summary.txt"
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pat)) if name not in ("c2", "c3") or "_c" not in os.path.basename(f)[3:])
    if not files:
        return None
    m = {}
    for line in open(files[-1]):
        t = line.split()
        if len(t) >= 2 and t[0].startswith(("SQ_", "TCC_", "FETCH", "WRITE")):
            try:
                m[t[0]] = float(t[1])
            except ValueError:
                pass
    need = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU")
    if any(k not in m for k in need):
        return None
    simd_quads = m["SQ_WAVE_CYCLES"] / 4.0  # four resident waves per SIMD for the whole launch (one 1024-thread workgroup per CU)
    return {
        # share of the SIMDs' time in which SOME wave has a VALU instruction in flight (sum over a SIMD's four waves: can exceed 1)
        "valu_active_per_simd": m["SQ_ACTIVE_INST_VALU"] / simd_quads,
        # lanes doing work per executed VALU instruction
        "valu_lane_utilisation": m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64.0),
        # issue slots: measured on gfx950 (tools/valu_rate.hip, profiles/r03_issue_rates.txt) a SIMD issues one full-rate VALU
        # wave-instruction per ~2.2 cycles and one half-rate (compare, select, shift, v_div_*, f64) or scalar one per ~4
        "valu_issue_share_at_2_cycles": m["SQ_INSTS_VALU"] * 2.0 / (simd_quads * 4.0),
        "valu_issue_share_at_4_cycles": m["SQ_INSTS_VALU"] * 4.0 / (simd_quads * 4.0),
        "salu_issue_share_at_4_cycles": m["SQ_INSTS_SALU"] * 4.0 / (simd_quads * 4.0),
        "source": os.path.relpath(files[-1], ROOT) + " (rocprofv3 --pmc passes of an earlier run of this command, not this run)",
    }


def roofline_entry(st, local_samples, verts, traffic=None, name=None):
    """Roofline of the dominant kernel (the path kernel) on this rank: algorithmic bytes per launch =
    192 B x radiance() invocations + 16 B x pixels of this rank's shard, over the kernel's average launch
    duration from HIP events recorded on the launch stream inside the library."""
    k_ms = st.trace_ms
    kernel = "path_queue_kernel" if st.block_threads == 1024 else "path_trace_kernel"
    alg_bytes = B_VERTEX * verts + 16.0 * local_samples / st.spp_effective
    achieved = alg_bytes / (k_ms * 1e-3) * 1e-9
    pm = pmc_figures(name) if name else None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_unit": "bytes per launch (rocprofv3 2*FETCH_SIZE+WRITE_SIZE, fabric side incl. Infinity-Cache hits); read from the latest profiles/rNN_traffic*.json (separate --pmc passes of this command), not measured in this run",
            "valu_utilisation": pm["valu_active_per_simd"] if pm else None, "valu": pm,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel": kernel, "kernel_ms": k_ms,
            "bytes_per_vertex": B_VERTEX, "note": "algorithmic bytes of the streaming formulation (SURVEY.md 8(d)); the kernel keeps path state in LDS and the fold stack in L2/Infinity Cache, so it is latency/VALU bound, not HBM bound"}


def measured_traffic(name, kernel, world):
    """HBM-side traffic of the path kernel per launch: measured off-line with rocprofv3 PMC passes
    (tools/profile_round.sh -> profiles/rNN_traffic[_cX].json, latest round); None if not measured."""
    if world != 1:
        return None
    pat = "r*_traffic.json" if name == "c2" else f"r*_traffic_{name}.json"
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if tfiles:
        tj = json.load(open(tfiles[-1]))
        if tj.get("kernel", "").startswith(kernel):
            return tj["traffic_bytes_per_launch"]
    return None


def want_specialised():
    """The scene-specialised path kernel (PINE_GPU_FLAG_SPECIALIZE: same arithmetic, bit-identical film) is what the bench
    runs where the scene qualifies; PINE_BENCH_SPECIALIZE=0 measures the precompiled kernels only."""
    return os.environ.get("PINE_BENCH_SPECIALIZE", "1") != "0"


def make_plan(scene, spp, depth, specialize, **kw):
    """(plan, note): the plan with the scene-specialised kernel if asked for and buildable here, else the precompiled one
    with the reason (a box without hipcc can still run the bench; the library itself fails loudly, as it should)."""
    import pine_amd
    if specialize:
        try:
            return pine_amd.Plan(scene, spp, depth, specialize=True, **kw), None
        except pine_amd.PineError as e:
            if "specialisation" not in str(e):
                raise
            return pine_amd.Plan(scene, spp, depth, **kw), str(e)[:300]
    return pine_amd.Plan(scene, spp, depth, **kw), None


def kernel_note(st, err=None):
    if st.specialized:
        return "scene-specialised (PINE_GPU_FLAG_SPECIALIZE: compiled for this scene at plan creation -- %s; %.0f ms; bit-identical film)" % (
            ("exact feature set + BVH and primitive records baked in" if not (st.kernel_features & 0x8000) else
             "exact feature set + the top-level BVH and its primitive records baked in (the mesh stays with the flat traversal)") if st.specialized == 2 else "exact feature set",
            st.specialize_ms)
    return "precompiled" + (" (specialisation failed: " + err + ")" if err else "")


def side_config(name, steps, warmup, device, fast=False, specialize=None):
    """One of the other BASELINE configs on this GPU, measured like the headline (N = 1): K timed renders
    bracketed by synchronisation, kernel time from the library's HIP events, md5 against the reference's.
    fast=True: the same with PINE_GPU_FLAG_FAST (declared-tolerance arithmetic) -- reported beside the exact
    numbers with its distance from the exact film; never the headline."""
    import torch
    import pine_amd
    from pine_amd import _lib
    build, spp, depth, text, key = _configs()[name]
    scene = build()
    W, H = scene.camera.film().size
    if specialize is None:
        specialize = want_specialised() and not fast
    plan, spec_err = make_plan(scene, spp, depth, specialize, device=device, timing=True, flags=_lib.FLAG_FAST if fast else 0)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        plan.launch(film.data_ptr(), stream)
    torch.cuda.synchronize()
    plan.stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.launch(film.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = plan.stats()
    host_film = film.cpu().numpy()
    md5 = hashlib.md5(host_film.tobytes()).hexdigest()
    want = golden_md5(key)
    kernel_used = kernel_note(st, spec_err)
    rl = roofline_entry(st, st.camera_samples, st.vertices, None if fast else measured_traffic(name, "path_queue_kernel" if st.block_threads == 1024 else "path_trace_kernel", 1),
                        None if fast else name)
    if fast:
        # distance from the exact film of the same scene (SURVEY.md 8(d)'s metric): rendered here, once
        import numpy as np
        ex = pine_amd.Plan(scene, spp, depth, device=device)
        film2 = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        ex.launch(film2.data_ptr(), stream)
        torch.cuda.synchronize()
        e = np.minimum(film2.cpu().numpy()[..., :3].astype(np.float64), 8.0)
        f = np.minimum(host_film[..., :3].astype(np.float64), 8.0)
        rel = np.linalg.norm(f - e, axis=-1) / (np.linalg.norm(e, axis=-1) + 1e-3)
        ex.close()
        plan.close()
        return {"config": name + "_fast", "mode": "PINE_GPU_FLAG_FAST (declared tolerance; not the parity gate)", "workload": text,
                "value": st.camera_samples * steps / dt * 1e-6, "unit": "Msamples/s", "steps": steps, "warmup": warmup,
                "ms_per_step": dt / steps * 1e3, "vertices_per_sample": st.vertices / st.camera_samples, "roofline": rl,
                "kernels_ms": {"prepass": st.prepass_ms, "path_trace": st.trace_ms, "resolve": st.resolve_ms},
                "distance_from_exact_film": {"pixels_within_rel_l2_1e-4": float((rel <= 1e-4).mean()),
                                             "rmse": float(np.sqrt(((f - e) ** 2).mean())),
                                             "bit_identical_pixels": float((film2.cpu().numpy().view(np.uint32) == host_film.view(np.uint32)).all(axis=-1).mean())}}
    out = {"config": name, "workload": text, "film": [W, H], "spp_effective": st.spp_effective, "max_path_length": depth,
           "value": st.camera_samples * steps / dt * 1e-6, "unit": "Msamples/s", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "vertices_per_sample": st.vertices / st.camera_samples,
           "walk_steps_per_sample": st.walk_steps / st.camera_samples,
           "samples_per_item": st.samples_per_item, "kernel": kernel_used, "roofline": rl,
           "kernels_ms": {"prepass": st.prepass_ms, "path_trace": st.trace_ms, "resolve": st.resolve_ms},
           "plan_ms": {"accel_build_host": st.accel_build_ms, "upload": st.upload_ms},
           "film_md5": md5, "reference_md5": want, "film_equals_reference": (md5 == want) if want else None}
    plan.close()
    del film
    torch.cuda.empty_cache()
    return out


def measure(name, steps, warmup, env, spi=0, specialize=None):
    """K timed renders of one config, sharded over the job's ranks (every rank calls this): W untimed steps, barrier +
    synchronize, K steps without host synchronisation, barrier + synchronize, MAX over ranks.  Returns a dict on every rank
    (film / statistics of rank 0's view; totals summed over ranks)."""
    import torch
    import pine_amd
    rank, world, local_rank, dist, use_dist, cdev = env["rank"], env["world"], env["local_rank"], env["dist"], env["use_dist"], env["cdev"]
    build, SPP, DEPTH, workload, stats_key = _configs()[name]
    scene = build()
    W, H = scene.camera.film().size
    plan, spec_err = make_plan(scene, SPP, DEPTH, want_specialised() if specialize is None else specialize, device=local_rank,
                               shard_rank=rank, shard_world=world, samples_per_item=spi, timing=True)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    main_stream = torch.cuda.current_stream()
    stream = main_stream.cuda_stream
    # N > 1: every rank writes only its own tiles, tile-major, into a slab (1/N of the film); the
    # slabs are gathered to rank 0 (direct sends over xGMI) and scattered into the row-major film there.
    # Slabs are double-buffered and the gather + unpack run on a second stream, so that they overlap the next
    # step's path kernel (PINE_BENCH_OVERLAP=0 puts everything back on one stream).
    # PINE_BENCH_COLLECTIVE=reduce selects the simpler form instead: full-size zero-initialised films summed to
    # rank 0 (exact: x + 0), a whole film per rank.
    collective = os.environ.get("PINE_BENCH_COLLECTIVE", "gather") if use_dist else "none"
    overlap = collective == "gather" and cdev == "cuda" and os.environ.get("PINE_BENCH_OVERLAP", "1") != "0"
    if collective == "gather":
        nbuf = 2 if overlap else 1
        slab = [torch.empty(plan.slab_floats(), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        slabs = [torch.empty((world, plan.slab_floats()), dtype=torch.float32, device="cuda") if rank == 0 else None for _ in range(nbuf)]
        slabs_c = [s if cdev == "cuda" else (torch.empty(s.shape, dtype=torch.float32) if rank == 0 else None) for s in slabs]
        slab_list = [list(s.unbind(0)) if rank == 0 else None for s in slabs_c]
        comm_stream = torch.cuda.Stream() if overlap else main_stream
        rendered = [torch.cuda.Event() for _ in range(nbuf)]
        consumed = [torch.cuda.Event() for _ in range(nbuf)]
    step_no = [0]

    def step():
        if collective == "gather":
            b = step_no[0] % len(slab)
            step_no[0] += 1
            if overlap:
                main_stream.wait_event(consumed[b])  # (the gather that read this slab two steps ago; a no-op the first time)
            plan.launch_packed(slab[b].data_ptr(), stream)
            if overlap:
                rendered[b].record(main_stream)
                comm_stream.wait_event(rendered[b])
            with torch.cuda.stream(comm_stream):
                dist.gather(slab[b] if cdev == "cuda" else slab[b].cpu(), slab_list[b], dst=0)
                if rank == 0:
                    if cdev != "cuda":
                        slabs[b].copy_(slabs_c[b])
                    pine_amd.film_unpack((W, H), world, slabs[b].data_ptr(), film.data_ptr(), local_rank, comm_stream.cuda_stream)
                if overlap:
                    consumed[b].record(comm_stream)
        else:
            plan.launch(film.data_ptr(), stream)
            if collective == "reduce":
                if cdev == "cuda":
                    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
                else:
                    host = film.cpu()
                    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                    film.copy_(host)

    def barrier():
        torch.cuda.synchronize()  # every stream of this device, the second one included
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    plan.stats()  # (discard the warm-up launches' kernel timings)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()  # no host synchronisation inside the timed loop
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel HIP-event timings (events recorded on the launch stream inside the library): mean
    # over the timed launches (the library keeps the last 64)
    st = plan.stats()
    local_samples = st.camera_samples
    verts = st.vertices
    if use_dist:
        t = torch.tensor([local_samples, verts], dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        total_samples, total_verts = float(t[0]), float(t[1])
    else:
        total_samples, total_verts = float(local_samples), float(verts)
    res = {"name": name, "scene": scene, "plan": plan, "film": film, "st": st, "dt": dt, "steps": steps, "warmup": warmup,
           "size": (W, H), "spp": SPP, "depth": DEPTH, "workload": workload, "stats_key": stats_key,
           "local_samples": local_samples, "verts": verts, "total_samples": total_samples, "total_verts": total_verts,
           "collective": collective, "overlap": bool(overlap) if use_dist else None, "stream": stream, "spec_err": spec_err}
    return res


def host_side_figures(res, steps, device):
    """'t_render spans kernel launch -> film resident on host (report with and without host BVH build +
    table upload)'.  N = 1 only.  (a) the same K renders with the film's device-to-host copy (pinned memory) inside the
    clock; (b) one cold render: plan creation from the built scene description (host BVH build + flattening, device
    allocation, upload of scene and sampler tables), the launch, the copy."""
    import torch
    import pine_amd
    W, H = res["size"]
    plan, film, stream = res["plan"], res["film"], res["stream"]
    host = torch.empty((H, W, 4), dtype=torch.float32).pin_memory()
    plan.launch(film.data_ptr(), stream)
    host.copy_(film, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.launch(film.data_ptr(), stream)
        host.copy_(film, non_blocking=True)  # (same stream: the copy follows the resolve kernel)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    samples = res["local_samples"]
    out = {"host_resident": {"value": samples * steps / dt * 1e-6, "unit": "Msamples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                             "what": "launch -> film in pinned host memory (6.55 MB device-to-host copy per render inside the clock); scene resident"}}
    build, SPP, DEPTH, _, _ = _configs()[res["name"]]
    specialised = bool(res["st"].specialized)

    def cold(specialize):
        scene2 = build()  # (a fresh scene object: its BVH is not built yet)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan2, _ = make_plan(scene2, SPP, DEPTH, specialize, device=device)
        plan2.launch(film.data_ptr(), stream)
        host.copy_(film, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st2 = plan2.stats()
        plan2.close()
        return {"value": samples / dt * 1e-6, "unit": "Msamples/s", "ms": dt * 1e3, "accel_build_host_ms": st2.accel_build_ms,
                "alloc_and_upload_ms": st2.upload_ms, "specialize_ms": st2.specialize_ms, "specialized": bool(st2.specialized)}

    out["including_build_and_upload"] = cold(specialised)
    out["including_build_and_upload"]["what"] = (
        "ONE cold render: plan creation (host BVH build + flattening, device allocation, upload of scene records and the 320 KB of sampler "
        "tables" + (", the scene's kernel fetched from the on-disk cache and loaded" if specialised else "") + "), launch, film to pinned host memory; "
        "the process's HIP module is already loaded")
    if specialised:
        # ... the same with the precompiled kernel, and with an EMPTY kernel cache (hipcc runs: the first render of a new scene ever)
        out["including_build_and_upload_precompiled"] = cold(False)
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            keep = os.environ.get("PINE_GPU_CACHE_DIR")
            os.environ["PINE_GPU_CACHE_DIR"] = tmp
            try:
                out["including_build_upload_and_kernel_compile"] = cold(True)
            finally:
                if keep is None:
                    del os.environ["PINE_GPU_CACHE_DIR"]
                else:
                    os.environ["PINE_GPU_CACHE_DIR"] = keep
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=["c2", "c2r", "c3", "c4", "c5"], help="the workload (default: BASELINE configs[1])")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the other configs' entries of the default line")
    ap.add_argument("--spi", type=int, default=0, help="samples per work item (0 = auto)")
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline loop: no host-side figures, no precompiled-kernel leg, "
                    "no other configs, no CPU leg (counter passes: every path-kernel dispatch of the run is the headline's)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu = args.no_configs = True

    import torch
    import pine_amd  # noqa: F401

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the PathIntegrator hot path has no CPU fallback")
    # Rehearsal knobs (not used by the driver): PINE_BENCH_BACKEND=gloo runs the N > 1 code path with the
    # collectives staged through host memory, PINE_BENCH_DEVICE pins every rank to one GPU -- together
    # they let a one-GPU box execute the multi-rank logic end to end (tools/rehearse_ranks.sh).
    backend = os.environ.get("PINE_BENCH_BACKEND", "nccl")
    if os.environ.get("PINE_BENCH_DEVICE"):
        local_rank = int(os.environ["PINE_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("PINE_BENCH_FORCE_DIST") == "1"  # the latter: exercise the RCCL calls at N=1
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"  # where collective buffers live
    env = {"rank": rank, "world": world, "local_rank": local_rank, "dist": dist, "use_dist": use_dist, "cdev": cdev}

    res = measure(args.config, args.steps, args.warmup, env, args.spi)
    st = res["st"]
    W, H = res["size"]
    # N > 1: BASELINE names C3 (1920x1080) as the 8-GPU workload; the headline stays C2 so that N = 1 equals the single-GPU
    # record, and C3 is measured right after it on the same ranks (every rank takes part; rank 0 reports)
    c3 = None
    if world > 1 and args.config == "c2" and not args.no_configs:
        res["plan"].close()
        res["plan"] = None
        c3 = measure("c3", max(2, args.steps // 3), 1, env, args.spi)

    if rank == 0:
        def line_of(r):
            s_ = r["st"]
            kernel = "path_queue_kernel" if s_.block_threads == 1024 else "path_trace_kernel"
            md5 = hashlib.md5(r["film"].cpu().numpy().tobytes()).hexdigest()
            want = golden_md5(r["stats_key"])
            return {
                "value": r["total_samples"] * r["steps"] / r["dt"] * 1e-6, "unit": "Msamples/s", "n_gpus": world,
                "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["dt"] / r["steps"] * 1e3,
                "config": {"workload": r["workload"], "name": r["name"],
                           "film": list(r["size"]), "spp_effective": s_.spp_effective, "max_path_length": r["depth"],
                           "parallelism": f"tiles8x8-roundrobin x{world}", "collective": r["collective"],
                           "overlap_gather_with_next_render": r["overlap"],
                           "kernel": kernel_note(s_, r.get("spec_err")),
                           "samples_per_item": s_.samples_per_item, "serial_tiles": s_.serial_tiles,
                           "grid_blocks": s_.grid_blocks, "vertices_per_sample": r["total_verts"] / r["total_samples"]},
                "roofline": roofline_entry(s_, r["local_samples"], r["verts"], measured_traffic(r["name"], kernel, world), r["name"] if world == 1 else None),
                "film_md5": md5, "reference_md5": want, "film_equals_reference": (md5 == want) if want else None,
                "kernels_ms": {"prepass": s_.prepass_ms, "path_trace": s_.trace_ms, "resolve": s_.resolve_ms, "launches_averaged": s_.timed_launches},
                "plan_ms": {"accel_build_host": s_.accel_build_ms, "upload": s_.upload_ms,
                            "note": "one-time host cost of plan creation, outside the timed region (scene resident before it starts)"},
            }
        head = line_of(res)
        out = {"metric": "Msamples/s", "value": head["value"], "unit": "Msamples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic (the reference's scene geometry rebuilt through the API; sampler tables are the published BlueSobol data)"}
        for k in ("config", "roofline", "film_md5", "reference_md5", "film_equals_reference", "kernels_ms", "plan_ms"):
            out[k] = head[k]
        film_md5 = head["film_md5"]
        if c3 is not None:
            e = line_of(c3)
            e["config_name"] = "c3"
            e["note"] = "BASELINE's multi-GPU workload (configs[2]) on the same ranks, measured after the headline; scaling: strong"
            out["configs"] = [e]
        if world == 1 and not args.headline_only:
            try:
                out.update(host_side_figures(res, max(2, args.steps // 2), local_rank))
            except Exception as e:  # report, keep the headline
                out["host_resident"] = {"error": str(e)[:300]}
        if world == 1 and st.specialized and not args.headline_only:
            # the same K renders with the precompiled kernel (what a caller without the flag, or a box without hipcc, gets)
            res["plan"].close()
            res["plan"] = None
            try:
                g_res = measure(args.config, max(2, args.steps // 2), 1, env, args.spi, specialize=False)
                g = line_of(g_res)
                g_res["plan"].close()
                out["precompiled_kernel"] = {k: g[k] for k in ("value", "unit", "ms_per_step", "steps", "kernels_ms", "film_equals_reference")}
                out["precompiled_kernel"]["roofline_frac"] = g["roofline"]["frac"]
            except Exception as e:
                out["precompiled_kernel"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_configs and args.config == "c2":
            if res["plan"] is not None:
                res["plan"].close()
            res["plan"] = None
            out["configs"] = []
            spec = want_specialised()
            for name, k, w, fast, specialize in (("c2r", 5, 1, False, spec), ("c3", 3, 1, False, spec), ("c4", 5, 1, False, spec), ("c5", 2, 1, False, spec),
                                                 ("c2r", 5, 1, False, False), ("c3", 3, 1, False, False), ("c5", 2, 1, False, False),
                                                 ("c2", 5, 1, True, False), ("c4", 5, 1, True, False), ("c5", 2, 1, True, False)):
                if not specialize and not fast and name in ("c2r", "c3", "c5") and not spec:
                    continue  # (already measured with the precompiled kernel above)
                try:
                    e = side_config(name, k, w, local_rank, fast, specialize)
                    if not fast and not specialize and spec and name in ("c2r", "c3", "c5"):
                        e["config"] = name + "_precompiled"
                    out["configs"].append(e)
                except Exception as e:  # report, keep the headline
                    out["configs"].append({"config": name + ("_fast" if fast else ""), "error": str(e)[:300]})
        if world == 1 and not args.no_cpu:
            port = cpu_baseline(res["scene"], (W, H), res["spp"], res["depth"])
            ref = cpu_baseline_reference(res["scene"], (W, H), res["spp"], res["depth"], film_md5)
            cb = ref if ref else port
            out["cpu_baseline"] = cb
            if ref:
                out["cpu_baseline_port"] = port  # this repo's CPU restatement of the same path, for comparison
            out["speedup_vs_cpu"] = out["value"] / cb["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
