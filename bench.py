#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configs.

metric   : Msamples/s = W*H*spp_effective / t_render, t_render = launch -> film resident in (pinned) HOST memory
workload : --config c2 (default) = configs[1] = cbox 640x640, BlueSampler(256), depth 8, as-committed camera;
           c2r = the same with the README camera; c3 = cbox 1920x1080, BlueSampler(1024 -> 256 effective), depth 8 (BASELINE's
           8-GPU workload); c4 = classic.pine + 10 000 cones 720x360, BlueSampler(64), depth 6; c5 = Subsurface icosphere in
           the Rect-only cbox 640x640, BlueSampler(512 -> 256 effective), depth 8.
step     : one full render of that film through the library's DEFAULT path (what `PathIntegrator(sampler, n).render(scene)`
           runs: path kernel + ordered resolve), then the film's device-to-host copy [N > 1: RCCL slab gather and
           unpack on rank 0 first]; scene, BVH, sampler tables, work buffers and the per-item RNG checkpoints -- a function of the
           film partition and the sample counts alone, computed by the plan's first launch (a warm-up step) -- are resident in HBM
           before the timed region.  `checkpoints_every_launch` is the same loop with that prepass back in every launch.
           The copy of step i runs on a second stream beside the render of step i+1 (two device films, two host films).
kernel   : the library's default mode: the scene's own kernel from the on-disk cache, else the precompiled
           kernel while the scene's kernel compiles in the background.  The bench is a steady-state measurement: after the W
           warm-up steps it waits (untimed, bounded) for a pending background build and warms up again; `config.kernel_mode`
           says what ran and where the kernel came from -- precompiled | specialised(cache: warm | cold, compiled in .. s).
           The precompiled kernel's figures are measured beside it (`precompiled`); PINE_BENCH_SPECIALIZE=0 makes them the headline.
N > 1    : `python bench.py --gpus N` without RANK in the environment starts N ranks itself (child processes of
           `python -m torch.distributed.run`, decided before anything touches a GPU; fewer than N visible devices is an
           error); under the driver's own torchrun launch the ranks are just this file.  8x8-pixel tiles dealt round-robin
           to ranks (strong scaling: the film is fixed); every rank writes only its own tiles into a slab, the slabs are
           gathered to rank 0 (direct xGMI sends), scattered into the row-major film there and copied to the host.  The
           gather + unpack + copy of step i run on a second stream while the path kernel of step i+1 runs.

Prints ONE compact JSON line (< 4 KB) on rank 0's stdout; everything else -- per-config dicts, cold renders, the fast mode,
the CPU restatement's figure, notes -- goes to bench_detail.json next to this file (--detail PATH) and to stderr.
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_VERTEX = 192.0  # algorithmic bytes per radiance() invocation
HBM_PEAK_GBS = 8000.0
LINE_LIMIT = 4096
# cycles per issued VALU wave-instruction at four waves per SIMD, measured on the MI355X (tools/valu_rate.hip,
# profiles/r03_issue_rates.txt): full rate 2.1 - 2.5, half rate (compare, select, shift, v_div_*, f64, conversions) 3.7 - 3.8,
# transcendental 7.3.  `other` = the static mix of the path kernels outside f64 / transcendental: 60 % full, 40 % half rate.
ISSUE_CYCLES = {"trans": 7.3, "f64": 3.75, "other": 0.6 * 2.3 + 0.4 * 3.75}


def _configs():
    from pine_amd import scenes
    return {
        # name: (scene builder, requested spp, depth, workload text, key in tests/golden/stats_640.json)
        "c2": (lambda: scenes.cbox((640, 640), "committed"), 256, 8,
               "cbox 640x640 256spp depth=8, as-committed camera, BlueSampler, pine-BVH order", "C2_cbox_640_s256_d8_committed"),
        "c2r": (lambda: scenes.cbox((640, 640), "readme"), 256, 8,
                "cbox 640x640 256spp depth=8, README camera [0,1,-4]->[0,1,0] fov 0.25, BlueSampler, pine-BVH order",
                "C2_cbox_640_s256_d8_readme"),
        "c3": (lambda: scenes.cbox((1920, 1080), "committed"), 1024, 8,
               "cbox 1920x1080 BlueSampler(1024) = 256 effective spp depth=8, as-committed camera, pine-BVH order",
               "C3_cbox_1920x1080_s1024_d8"),
        "c4": (lambda: scenes.classic_cones((720, 360), 100), 64, 6,
               "classic.pine + 10 000 procedurally placed cones 720x360 64spp depth=6", "C4_classic_10k_cones_720x360_s64_d6"),
        "c5": (lambda: scenes.sss((640, 640), 3), 512, 8,
               "Subsurface icosphere (1280 triangles) + emissive Rect in the Rect-only cbox 640x640 BlueSampler(512) = 256 effective spp depth=8",
               "C5_sss_640_s512_d8"),
        # the same workloads as a `.pine` script renders them on real pine: EmbreeAccel (PINE_GPU_FLAG_ORDER_EMBREE; the md5 is that of the
        # reference built WITH Embree, tests/golden/stats_640_embree.json) -- side figures, never the headline
        "c2e": (lambda: scenes.cbox((640, 640), "committed"), 256, 8,
                "cbox 640x640 256spp depth=8, as-committed camera, BlueSampler, EmbreeAccel's order (the .pine default)", "embree:C2_cbox_640_s256_d8_committed"),
        "c4e": (lambda: scenes.classic_cones((720, 360), 100), 64, 6,
                "classic.pine + 10 000 cones 720x360 64spp depth=6, EmbreeAccel's order (the .pine default)", "embree:C4_classic_10k_cones_720x360_s64_d6"),
    }


def golden_md5(key):
    try:
        if key.startswith("embree:"):
            return json.load(open(os.path.join(ROOT, "tests", "golden", "stats_640_embree.json")))[key[7:]]["md5"]
        return json.load(open(os.path.join(ROOT, "tests", "golden", "stats_640.json")))[key]["md5"]
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------------
# the line: built by a pure function so that a CPU test can hold it to its size (tests/test_bench_line.py)
# ---------------------------------------------------------------------------------------------------------------------
def _r(x, digits=4):
    """Round floats for the line (the detail file keeps full precision)."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}") if abs(x) < 1e15 else x
    return x


def compact_line(full):
    """The ONE stdout line from the full result dict: the contract's keys, `roofline`, `cpu_baseline`, and one short summary per
    side measurement.  Never more than LINE_LIMIT bytes: optional parts are dropped, last first, until it fits."""
    head = full["headline"]
    rl = head["roofline"]
    line = {
        "metric": "Msamples/s", "value": _r(head["value"], 6), "unit": "Msamples/s", "n_gpus": full["n_gpus"], "steps": full["steps"],
        "warmup": full["warmup"], "ms_per_step": _r(head["ms_per_step"], 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": head["workload"][:120], "name": head["name"], "film": head["film"], "spp_effective": head["spp_effective"],
                   "max_path_length": head["max_path_length"], "parallelism": head["parallelism"], "collective": head["collective"],
                   "kernel_mode": head["kernel_mode"][:100], "clock": "launch -> film in pinned host memory",
                   "resident": "scene, BVH, tables, rng checkpoints"},
        "roofline": {"bound": "hbm", "achieved": _r(rl["achieved"], 5), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": _r(rl["frac"], 4),
                     "traffic": _r(rl.get("traffic"), 5), "kernel": rl["kernel"][:48], "kernel_ms": _r(rl["kernel_ms"], 5),
                     "algorithmic_bytes_per_launch": _r(rl["algorithmic_bytes_per_launch"], 6),
                     "valu_issue_frac": _r(rl.get("valu_issue_frac"), 3), "valu_lane_utilisation": _r(rl.get("valu_lane_utilisation"), 3),
                     "counters_from": rl.get("counters_from")},
        "film_equals_reference": head.get("film_equals_reference"),
    }
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"], 5), "unit": "Msamples/s", "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"][:96], "film_equals_gpu": cb.get("film_equals_gpu")}
        line["speedup_vs_cpu"] = _r(head["value"] / cb["value"], 5) if cb["value"] else None
    optional = []  # (key, value), most important first
    if full.get("device_resident"):
        d = full["device_resident"]
        optional.append(("device_resident", {"value": _r(d["value"], 5), "ms_per_step": _r(d["ms_per_step"], 5)}))
    if full.get("checkpoints_every_launch") and "value" in full["checkpoints_every_launch"]:
        d = full["checkpoints_every_launch"]
        optional.append(("checkpoints_every_launch", {k: _r(d.get(k), 5) for k in ("value", "ms_per_step")}))
    if full.get("precompiled"):
        d = full["precompiled"]
        optional.append(("precompiled", {k: _r(d.get(k), 5) for k in ("value", "ms_per_step", "kernel_ms", "roofline_frac", "film_equals_reference")}
                         if "error" not in d else {"error": d["error"][:80]}))
    side = []
    for e in full.get("configs", []):
        if str(e.get("config", "")).endswith("e"):  # (the EmbreeAccel-mode side figures, c2e / c4e, stay in the detail file)
            continue
        if "error" in e:
            side.append({"name": e.get("config"), "error": e["error"][:60]})
        else:
            side.append({"name": e["config"], "value": _r(e["value"], 5), "ms": _r(e["ms_per_step"], 5), "kernel_ms": _r(e["roofline"]["kernel_ms"], 5),
                         "frac": _r(e["roofline"]["frac"], 3), "mode": e.get("kernel_mode_short"), "ok": e.get("film_equals_reference")})
    if side:
        optional.append(("configs", side))
    if full.get("cold"):
        optional.append(("cold_ms", {k: _r(v.get("ms"), 5) for k, v in full["cold"].items() if isinstance(v, dict)}))
    optional.append(("detail", full.get("detail_file")))
    for k, v in optional:
        line[k] = v
    text = json.dumps(line, separators=(",", ":"))
    for k, _ in reversed(optional):  # never over the limit: drop optional parts, last first
        if len(text) < LINE_LIMIT:
            break
        line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:  # (cannot happen with the truncations above; a hard guarantee all the same)
        line["config"]["workload"] = line["config"]["workload"][:40]
        line["roofline"].pop("counters_from", None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(scene, size, spp, depth, budget_s=12.0):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded,
    stratified sample of the same workload: 8-row bands spread evenly over the film."""
    from oracle import oracle
    W, H = size
    ps = scene.describe()
    threads = os.cpu_count() or 1
    t0 = time.time()
    n = 0
    for y in (H // 4 // 8 * 8, 3 * H // 4 // 8 * 8):  # calibrate on two bands (one in each half of the film)
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
        "sample": f"{len(ys)} of {bands_all} 8-row bands of the {W}x{H} render ({samples / total:.3f} of the workload, {secs:.1f} s)",
        "vertices_per_sample": verts / samples,
    }


def cpu_baseline_reference(scene, size, spp, depth, gpu_md5, budget_s=40.0):
    """The REAL reference timed beside the GPU: oracle/_ref/pine_ref is pine's own PathIntegrator + BVH, compiled from the
    reference's sources by oracle/Makefile in the build container (the binary travels, the sources do not).  Renders the same
    scene description; the whole workload when a 16-spp calibration run says it fits the budget, else the largest power-of-two
    spp that does.  Returns None when the binary is absent or fails (the caller then reports the CPU restatement, kind "port")."""
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
                r = subprocess.run([exe, "render", sp, str(n), str(depth), fp], capture_output=True, text=True, timeout=limit)
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
        return {
            "value": res["msamples_per_s"], "unit": "Msamples/s", "cores": res["threads"], "kind": "reference",
            "sample": (f"whole workload {W}x{H}x{n}spp d{depth} by oracle/_ref/pine_ref in {res['seconds']:.1f} s" if whole else
                       f"{W}x{H}x{n}spp d{depth} ({n}/{spp_full} of the spp, {res['seconds']:.1f} s) by oracle/_ref/pine_ref"),
            "what": "oracle/_ref/pine_ref = the reference's own PathIntegrator(BVH, BlueSampler, UniformLightSampler), its sources compiled in place by oracle/Makefile",
            "film_md5": md5,
            "film_equals_gpu": (md5 == gpu_md5) if whole else None,
        }
    except Exception as e:  # the reference binary is optional test infrastructure: report, do not fail the bench
        print(f"[bench] reference CPU baseline unavailable: {e}", file=sys.stderr)
        return None


# ---------------------------------------------------------------------------------------------------------------------
# counters measured off-line (rocprofv3 --pmc passes, tools/profile_round.sh), stamped with the kernel sources' hash
# ---------------------------------------------------------------------------------------------------------------------
def kernel_source_hash():
    """sha1 over the device sources the path kernels are built from: what a committed counter file was measured on."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "pine_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:12]


def offline_counters(name, kernel_mode_short):
    """VALU issue / lane figures and fabric traffic of this config's path kernel from the latest committed
    profiles/rNN_counters.json (separate rocprofv3 --pmc passes of `bench.py --headline-only`; NOT measured in this run).  The
    entry names the kernel sources it was measured on; `stale` says whether they are still the ones in this tree."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))
    if not files:
        return None
    try:
        allc = json.load(open(files[-1]))
    except Exception:
        return None
    e = allc.get(f"{name}:{kernel_mode_short}") or allc.get(name)
    if not e or e.get("mode", kernel_mode_short) != kernel_mode_short:
        return None
    c = e["counters"]
    other = c["SQ_INSTS_VALU"] - c.get("SQ_INSTS_VALU_TRANS_F32", 0.0) - c.get("SQ_INSTS_VALU_FMA_F64", 0.0) - c.get("SQ_INSTS_VALU_MUL_F64", 0.0)
    f64 = c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0)
    issue_cycles = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0) * ISSUE_CYCLES["trans"] + f64 * ISSUE_CYCLES["f64"] + other * ISSUE_CYCLES["other"]
    # SQ_WAVE_CYCLES counts 4-cycle quanta summed over waves; at four resident waves per SIMD for the whole launch that IS the
    # launch's SIMD-cycles summed over the chip's SIMDs
    out = {"valu_issue_frac": issue_cycles / c["SQ_WAVE_CYCLES"],
           "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0),
           "wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
           "salu_issue_frac": c.get("SQ_INSTS_SALU", 0.0) * 3.7 / c["SQ_WAVE_CYCLES"],
           "traffic": e.get("traffic_bytes_per_launch"),
           "counters_from": f"{os.path.basename(files[-1])}@{e.get('kernel_source_hash', '?')}",
           "stale": e.get("kernel_source_hash") != kernel_source_hash()}
    if out["stale"]:
        out["counters_from"] += " (STALE: kernels changed since)"
    return out


def kernel_name(st):
    if st.block_threads != 1024:
        return "path_trace_kernel"
    return f"pine_scene_kernel_{st.kernel_features}" if st.specialized > 0 else f"path_queue_kernel<{st.kernel_features}>"


def mode_of(st, compile_wait_s=None):
    """(long, short) description of the path kernel that rendered, from the plan's statistics."""
    if st.specialized > 0:
        what = "scene baked in" if st.specialized == 2 else "exact feature set"
        if st.specialize_source == 1:
            return f"specialised(cache: warm; {what})", "specialised"
        if st.specialize_source == 3:
            return f"specialised(cache: cold, compiled in {compile_wait_s or 0:.1f} s during warm-up; {what})", "specialised"
        return f"specialised(compiled at plan creation in {st.specialize_ms / 1e3:.1f} s; {what})", "specialised"
    if st.specialized < 0:
        return "precompiled (background build failed)", "precompiled"
    return "precompiled", "precompiled"


def roofline_entry(st, local_samples, verts, name, mode_short, world=1):
    """Roofline of the dominant kernel (the path kernel) on this rank: algorithmic bytes per launch = 192 B x radiance()
    invocations + 16 B x pixels of this rank's shard, over the kernel's average launch duration from HIP events recorded on the
    launch stream inside the library."""
    k_ms = st.trace_ms
    alg_bytes = B_VERTEX * verts + 16.0 * local_samples / st.spp_effective
    achieved = alg_bytes / (k_ms * 1e-3) * 1e-9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "algorithmic_bytes_per_launch": alg_bytes, "kernel": kernel_name(st), "kernel_ms": k_ms, "bytes_per_vertex": B_VERTEX}
    oc = offline_counters(name, mode_short) if world == 1 else None
    if oc:
        out.update(oc)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# measurement
# ---------------------------------------------------------------------------------------------------------------------
def want_specialised():
    """PINE_BENCH_SPECIALIZE=0: precompiled kernels only (the headline too).  Default: the library's default mode."""
    return os.environ.get("PINE_BENCH_SPECIALIZE", "1") != "0"


def measure(name, steps, warmup, env, spi=0, specialize=None, host_copy=True):
    """K timed renders of one config, sharded over the job's ranks (every rank calls this): W untimed steps [+ an untimed,
    bounded wait for a pending background kernel build and W more], barrier + synchronize, K steps without host
    synchronisation, barrier + synchronize, MAX over ranks.  host_copy: the film's device-to-host copy inside the clock
    (rank 0), overlapped with the next step's render.  Returns a dict on every rank."""
    import torch
    import pine_amd
    rank, world, local_rank, dist, use_dist, cdev = env["rank"], env["world"], env["local_rank"], env["dist"], env["use_dist"], env["cdev"]
    build, SPP, DEPTH, workload, stats_key = _configs()[name]
    scene = build()
    W, H = scene.camera.film().size
    plan = pine_amd.Plan(scene, SPP, DEPTH, device=local_rank, shard_rank=rank, shard_world=world, samples_per_item=spi, timing=True,
                         specialize=specialize, order="embree" if stats_key.startswith("embree:") else "pine")
    main_stream = torch.cuda.current_stream()
    stream = main_stream.cuda_stream
    # N > 1: every rank writes only its own tiles, tile-major, into a slab (1/N of the film); the slabs are gathered to rank 0
    # (direct sends over xGMI) and scattered into the row-major film there.  PINE_BENCH_COLLECTIVE=reduce selects the simpler
    # form instead: full-size zero-initialised films summed to rank 0 (exact: x + 0), a whole film per rank.
    collective = os.environ.get("PINE_BENCH_COLLECTIVE", "gather") if use_dist else "none"
    overlap = os.environ.get("PINE_BENCH_OVERLAP", "1") != "0" and (collective != "reduce")
    nbuf = 2 if overlap else 1
    film = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(nbuf if (rank == 0 or collective != "gather") else 1)]
    host = [torch.empty((H, W, 4), dtype=torch.float32).pin_memory() for _ in range(nbuf)] if (host_copy and rank == 0) else None
    side_stream = torch.cuda.Stream() if overlap else main_stream  # gather + unpack + device-to-host copy
    rendered = [torch.cuda.Event() for _ in range(nbuf)]
    consumed = [torch.cuda.Event() for _ in range(nbuf)]
    if collective == "gather":
        slab = [torch.empty(plan.slab_floats(), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
        slabs = [torch.empty((world, plan.slab_floats()), dtype=torch.float32, device="cuda") if rank == 0 else None for _ in range(nbuf)]
        slabs_c = [s if cdev == "cuda" else (torch.empty(s.shape, dtype=torch.float32) if rank == 0 else None) for s in slabs]
        slab_list = [list(s.unbind(0)) if rank == 0 else None for s in slabs_c]
    step_no = [0]

    def step():
        b = step_no[0] % nbuf
        step_no[0] += 1
        if overlap:
            main_stream.wait_event(consumed[b])  # (whoever read this buffer two steps ago; a no-op the first time)
        if collective == "gather":
            plan.launch_packed(slab[b].data_ptr(), stream)
        else:
            plan.launch(film[b].data_ptr(), stream)
            if collective == "reduce":
                if cdev == "cuda":
                    dist.reduce(film[b], dst=0, op=dist.ReduceOp.SUM)
                else:
                    h_ = film[b].cpu()
                    dist.reduce(h_, dst=0, op=dist.ReduceOp.SUM)
                    film[b].copy_(h_)
        if overlap:
            rendered[b].record(main_stream)
            side_stream.wait_event(rendered[b])
        with torch.cuda.stream(side_stream):
            if collective == "gather":
                dist.gather(slab[b] if cdev == "cuda" else slab[b].cpu(), slab_list[b], dst=0)
                if rank == 0:
                    if cdev != "cuda":
                        slabs[b].copy_(slabs_c[b])
                    pine_amd.film_unpack((W, H), world, slabs[b].data_ptr(), film[b].data_ptr(), local_rank, side_stream.cuda_stream)
            if host is not None:
                host[b].copy_(film[b], non_blocking=True)
            if overlap:
                consumed[b].record(side_stream)

    def barrier():
        torch.cuda.synchronize()  # every stream of this device, the second one included
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    # a background build of the scene's kernel (cold cache): steady state is what is measured -- wait for it (untimed, bounded),
    # then warm up again with the kernel that will be timed
    torch.cuda.synchronize()
    st0 = plan.stats()
    compile_wait_s = None
    if st0.specialize_pending:
        t0 = time.perf_counter()
        limit = float(os.environ.get("PINE_BENCH_COMPILE_WAIT_S", "180"))
        while plan.stats().specialize_pending and time.perf_counter() - t0 < limit:
            time.sleep(0.05)
        compile_wait_s = time.perf_counter() - t0
        for _ in range(max(warmup, 1)):
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

    # per-kernel HIP-event timings (events recorded on the launch stream inside the library): mean over the timed launches
    st = plan.stats()
    local_samples = st.camera_samples
    verts = st.vertices
    if use_dist:
        t = torch.tensor([local_samples, verts], dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        total_samples, total_verts = float(t[0]), float(t[1])
    else:
        total_samples, total_verts = float(local_samples), float(verts)
    last = (step_no[0] - 1) % nbuf
    final = None
    if rank == 0:
        final = host[last].numpy() if host is not None else film[last].cpu().numpy()
    return {"name": name, "scene": scene, "plan": plan, "film": film, "film_host": final, "st": st, "dt": dt, "steps": steps, "warmup": warmup,
            "size": (W, H), "spp": SPP, "depth": DEPTH, "workload": workload, "stats_key": stats_key,
            "local_samples": local_samples, "verts": verts, "total_samples": total_samples, "total_verts": total_verts,
            "collective": collective, "overlap": bool(overlap), "stream": stream, "compile_wait_s": compile_wait_s, "host_copy": host_copy,
            "world": world}


def entry_of(r):
    """Result dict of one measured config (rank 0)."""
    s_ = r["st"]
    world = r["world"]
    md5 = hashlib.md5(r["film_host"].tobytes()).hexdigest()
    want = golden_md5(r["stats_key"])
    mode_long, mode_short = mode_of(s_, r["compile_wait_s"])
    return {
        "config": r["name"], "name": r["name"], "workload": r["workload"], "film": list(r["size"]), "spp_effective": s_.spp_effective,
        "max_path_length": r["depth"], "value": r["total_samples"] * r["steps"] / r["dt"] * 1e-6, "unit": "Msamples/s", "n_gpus": world,
        "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["dt"] / r["steps"] * 1e3,
        "clock": "launch -> film in pinned host memory (copy overlapped with the next render)" if r["host_copy"] else "launch -> film in device memory",
        "parallelism": f"tiles8x8-roundrobin x{world}", "collective": r["collective"], "overlap_side_stream": r["overlap"],
        "kernel_mode": mode_long, "kernel_mode_short": mode_short, "compile_wait_s": r["compile_wait_s"],
        "samples_per_item": s_.samples_per_item, "serial_tiles": s_.serial_tiles, "grid_blocks": s_.grid_blocks,
        "vertices_per_sample": r["total_verts"] / r["total_samples"], "walk_steps_per_sample": s_.walk_steps / max(1, s_.camera_samples),
        "roofline": roofline_entry(s_, r["local_samples"], r["verts"], r["name"], mode_short, world),
        "film_md5": md5, "reference_md5": want, "film_equals_reference": (md5 == want) if want else None,
        "kernels_ms": {"prepass": s_.prepass_ms, "path_trace": s_.trace_ms, "resolve": s_.resolve_ms, "launches_averaged": s_.timed_launches},
        "plan_ms": {"accel_build_host": s_.accel_build_ms, "upload": s_.upload_ms, "specialize": s_.specialize_ms,
                    "note": "one-time host cost of plan creation, outside the timed region (scene resident before it starts)"},
    }


def fast_mode_entry(name, steps, warmup, device):
    """PINE_GPU_FLAG_FAST (declared-tolerance arithmetic) beside the exact numbers, with its distance from the exact film; never
    the headline, detail file only."""
    import numpy as np
    import torch
    import pine_amd
    from pine_amd import _lib
    build, spp, depth, text, _ = _configs()[name]
    scene = build()
    W, H = scene.camera.film().size
    plan = pine_amd.Plan(scene, spp, depth, device=device, timing=True, flags=_lib.FLAG_FAST)
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
    ex = pine_amd.Plan(scene, spp, depth, device=device, specialize=False)
    film2 = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    ex.launch(film2.data_ptr(), stream)
    torch.cuda.synchronize()
    e = np.minimum(film2.cpu().numpy()[..., :3].astype(np.float64), 8.0)
    f = np.minimum(host_film[..., :3].astype(np.float64), 8.0)
    rel = np.linalg.norm(f - e, axis=-1) / (np.linalg.norm(e, axis=-1) + 1e-3)
    out = {"config": name + "_fast", "mode": "PINE_GPU_FLAG_FAST (declared tolerance; not the parity gate)", "workload": text,
           "value": st.camera_samples * steps / dt * 1e-6, "unit": "Msamples/s", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "kernel_ms": st.trace_ms,
           "distance_from_exact_film": {"pixels_within_rel_l2_1e-4": float((rel <= 1e-4).mean()), "rmse": float(np.sqrt(((f - e) ** 2).mean())),
                                        "bit_identical_pixels": float((film2.cpu().numpy().view(np.uint32) == host_film.view(np.uint32)).all(axis=-1).mean())}}
    ex.close()
    plan.close()
    return out


def cold_renders(res, device):
    """'report with and without host BVH build + table upload'.  N = 1 only.  ONE cold render each way: plan
    creation from the built scene description (host BVH build + flattening, device allocation, upload of scene and sampler
    tables [+ the scene's kernel read from the cache and loaded]), the launch, the film's copy to pinned host memory."""
    import torch
    import pine_amd
    W, H = res["size"]
    build, SPP, DEPTH, _, _ = _configs()[res["name"]]
    film = res["film"][0]
    stream = res["stream"]
    host = torch.empty((H, W, 4), dtype=torch.float32).pin_memory()
    samples = res["local_samples"]

    def cold(specialize):
        scene2 = build()  # (a fresh scene object: its BVH is not built yet)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan2 = pine_amd.Plan(scene2, SPP, DEPTH, device=device, specialize=specialize)
        plan2.launch(film.data_ptr(), stream)
        host.copy_(film, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st2 = plan2.stats()
        plan2.close()
        return {"value": samples / dt * 1e-6, "unit": "Msamples/s", "ms": dt * 1e3, "accel_build_host_ms": st2.accel_build_ms,
                "alloc_and_upload_ms": st2.upload_ms, "specialize_ms": st2.specialize_ms, "specialized": st2.specialized,
                "specialize_source": st2.specialize_source}

    out = {"default_mode": cold(None), "precompiled": cold(False),
           "what": "ONE cold render: plan creation (host BVH build + flattening, device allocation, upload of scene records and the 320 KB of "
                   "sampler tables; default mode: the scene's kernel read from the on-disk cache and loaded when it is there), launch, film to "
                   "pinned host memory; the process's HIP module is already loaded"}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` started plainly runs N ranks
# ---------------------------------------------------------------------------------------------------------------------
def visible_gpus():
    """Number of HIP devices this process tree may use, found WITHOUT initialising a GPU runtime here: a child process asks
    torch (device_count() does not create a context), so the parent stays free to start rank processes afterwards."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def launch_ranks(args, argv):
    """Start N fresh rank processes (torch.distributed.run, rendezvous on 127.0.0.1) and pass their output through.  Returns the
    exit code.  Nothing in THIS process has touched a GPU or imported torch."""
    import socket
    n = args.gpus
    rehearsal = os.environ.get("PINE_BENCH_DEVICE") is not None  # every rank on one device, collectives through gloo (tools/rehearse_ranks.sh)
    if not rehearsal:
        have = visible_gpus()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible: refusing to report n_gpus={n} from fewer devices "
                  f"(PINE_BENCH_DEVICE=0 PINE_BENCH_BACKEND=gloo rehearses the N-rank logic on one GPU)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=["c2", "c2r", "c3", "c4", "c5"], help="the workload (default: BASELINE configs[1])")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the other configs' entries")
    ap.add_argument("--spi", type=int, default=0, help="samples per work item (0 = auto)")
    ap.add_argument("--detail", default=os.environ.get("PINE_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")), help="where the full result dict goes")
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline loop: no side figures, no precompiled-kernel leg, "
                    "no other configs, no CPU leg (counter passes: every path-kernel dispatch of the run is the headline's)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu = args.no_configs = True
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    import torch
    import pine_amd  # noqa: F401

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: n_gpus would not be the number of ranks that rendered")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the PathIntegrator hot path has no CPU fallback")
    # Rehearsal knobs (not used by the driver): PINE_BENCH_BACKEND=gloo runs the N > 1 code path with the collectives staged
    # through host memory, PINE_BENCH_DEVICE pins every rank to one GPU -- together they let a one-GPU box execute the
    # multi-rank logic end to end (tools/rehearse_ranks.sh).
    backend = os.environ.get("PINE_BENCH_BACKEND", "nccl")
    if os.environ.get("PINE_BENCH_DEVICE"):
        local_rank = int(os.environ["PINE_BENCH_DEVICE"])
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("PINE_BENCH_FORCE_DIST") == "1"  # the latter: exercise the RCCL calls at N=1
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:  # (PINE_BENCH_FORCE_DIST=1 started plainly: a one-rank group of our own)
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"  # where collective buffers live
    env = {"rank": rank, "world": world, "local_rank": local_rank, "dist": dist, "use_dist": use_dist, "cdev": cdev}
    spec = None if want_specialised() else False

    res = measure(args.config, args.steps, args.warmup, env, args.spi, specialize=spec)
    # N > 1: BASELINE names C3 (1920x1080) as the 8-GPU workload; the headline stays C2 so that N = 1 equals the single-GPU
    # record, and C3 is measured right after it on the same ranks (every rank takes part; rank 0 reports)
    c3 = None
    if world > 1 and args.config == "c2" and not args.no_configs:
        res["plan"].close()
        res["plan"] = None
        c3 = measure("c3", max(2, args.steps // 3), 1, env, args.spi, specialize=spec)

    if rank == 0:
        head = entry_of(res)
        full = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "headline": head, "configs": [],
                "detail_file": os.path.basename(args.detail), "git_kernel_source_hash": kernel_source_hash(),
                "notes": {"metric": "Msamples/s = W*H*spp_effective / t, t = launch -> film in pinned host memory (SURVEY.md 8(d)); scene resident",
                          "roofline": "algorithmic bytes of the streaming formulation (SURVEY.md 8(d): 192 B per radiance() invocation + 16 B per pixel) over the path "
                                      "kernel's mean launch duration (HIP events on the launch stream); the kernel keeps path state in LDS, so it is bound by "
                                      "instruction issue and latency, not by HBM: `valu_issue_frac` = issued VALU wave-instructions x measured cycles per issue by "
                                      "class / SIMD-cycles of the launch, from separate rocprofv3 --pmc passes (counters_from names file and kernel-source hash)",
                          "traffic": "bytes per launch, rocprofv3 2*FETCH_SIZE+WRITE_SIZE (fabric side incl. Infinity-Cache hits), same off-line passes",
                          "data": "the reference's scene geometry rebuilt through the API; sampler tables are the published BlueSobol data"}}
        film_md5 = head["film_md5"]
        if c3 is not None:
            e = entry_of(c3)
            e["note"] = "BASELINE's multi-GPU workload (configs[2]) on the same ranks, measured after the headline; scaling: strong"
            full["configs"].append(e)
            c3["plan"].close()
        if world == 1 and not args.headline_only:
            # the same K renders without the host copy (film stays in HBM)
            res["plan"].close()
            res["plan"] = None
            try:
                d = measure(args.config, max(2, args.steps // 2), 1, env, args.spi, specialize=spec, host_copy=False)
                de = entry_of(d)
                full["device_resident"] = {k: de[k] for k in ("value", "ms_per_step", "steps", "kernels_ms", "clock")}
                full["cold"] = cold_renders(d, local_rank)
                d["plan"].close()
            except Exception as e:  # report, keep the headline
                full["device_resident"] = None
                full["errors"] = full.get("errors", []) + ["device_resident/cold: " + str(e)[:300]]
            # ... and with the RNG-checkpoint prepass in EVERY launch, as before round 4 (the table is a function of the film partition and
            # the sample counts alone: the plan's first launch -- a warm-up step here -- computes it, later launches reuse it)
            try:
                os.environ["PINE_GPU_CKPT_EVERY_LAUNCH"] = "1"
                q_res = measure(args.config, max(2, args.steps // 2), 1, env, args.spi, specialize=spec)
                q = entry_of(q_res)
                q_res["plan"].close()
                full["checkpoints_every_launch"] = {"value": q["value"], "ms_per_step": q["ms_per_step"], "steps": q["steps"]}
            except Exception as e:
                full["checkpoints_every_launch"] = {"error": str(e)[:300]}
            finally:
                os.environ.pop("PINE_GPU_CKPT_EVERY_LAUNCH", None)
            if head["kernel_mode_short"] == "specialised":
                # ... and with the precompiled kernel (what a first-sight scene runs while its kernel compiles, or a box without hipcc)
                try:
                    g_res = measure(args.config, max(2, args.steps // 2), 1, env, args.spi, specialize=False)
                    g = entry_of(g_res)
                    g_res["plan"].close()
                    full["precompiled"] = {"value": g["value"], "ms_per_step": g["ms_per_step"], "steps": g["steps"], "kernel_ms": g["roofline"]["kernel_ms"],
                                           "roofline_frac": g["roofline"]["frac"], "roofline": g["roofline"], "film_equals_reference": g["film_equals_reference"]}
                except Exception as e:
                    full["precompiled"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_configs and args.config == "c2":
            if res["plan"] is not None:
                res["plan"].close()
                res["plan"] = None
            full["fast_mode"] = []
            for name, k in (("c2r", 5), ("c3", 3), ("c4", 10), ("c5", 2), ("c2e", 5), ("c4e", 10)):
                try:
                    r2 = measure(name, k, 1, env, args.spi, specialize=spec)
                    e = entry_of(r2)
                    r2["plan"].close()
                    del r2
                    torch.cuda.empty_cache()
                    full["configs"].append(e)
                    if e["kernel_mode_short"] == "specialised" and not name.endswith("e"):
                        r3 = measure(name, max(2, k // 2), 1, env, args.spi, specialize=False)
                        e3 = entry_of(r3)
                        r3["plan"].close()
                        del r3
                        torch.cuda.empty_cache()
                        e["precompiled"] = {"value": e3["value"], "ms_per_step": e3["ms_per_step"], "kernel_ms": e3["roofline"]["kernel_ms"],
                                            "roofline_frac": e3["roofline"]["frac"], "film_equals_reference": e3["film_equals_reference"]}
                except Exception as e:  # report, keep the headline
                    full["configs"].append({"config": name, "error": str(e)[:300]})
            for name, k in (("c2", 5), ("c4", 5), ("c5", 2)):
                try:
                    full["fast_mode"].append(fast_mode_entry(name, k, 1, local_rank))
                except Exception as e:
                    full["fast_mode"].append({"config": name + "_fast", "error": str(e)[:300]})
        if world == 1 and not args.no_cpu:
            W, H = res["size"]
            port = cpu_baseline(res["scene"], (W, H), res["spp"], res["depth"])
            ref = cpu_baseline_reference(res["scene"], (W, H), res["spp"], res["depth"], film_md5)
            full["cpu_baseline"] = ref if ref else port
            if ref:
                full["cpu_baseline_port"] = port  # this repo's CPU restatement of the same path, for comparison
        text = compact_line(full)
        try:
            with open(args.detail, "w") as f:
                json.dump(full, f, indent=1, default=str)
        except OSError as e:
            print(f"[bench] cannot write {args.detail}: {e}", file=sys.stderr)
        print("[bench detail] " + json.dumps(full, default=str), file=sys.stderr, flush=True)
        print(text, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
