#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric   : Msamples/s = W*H*spp_effective / t   
workload : configs[1] = cbox 640x640, BlueSampler(256), depth 8, as-committed camera
step     : one full render of that film (prepass + path kernel + ordered resolve [+ RCCL film
           reduce when N > 1]); scene, BVH, sampler tables and work buffers are resident in HBM
           before the timed region.
N > 1    : 8x8-pixel tiles dealt round-robin to ranks (strong scaling: the film is fixed), each rank
           renders its tiles into a zero-initialised full-size film, one RCCL reduce(sum) to rank 0
           per step (exact: x + 0).

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP, DEPTH = 640, 640, 256, 8
B_VERTEX = 192.0  # algorithmic bytes per radiance() invocation
HBM_PEAK_GBS = 8000.0


def cpu_baseline(scene, budget_s=12.0):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded,
    stratified sample of the same workload: 8-row bands spread evenly over the film."""
    from oracle import oracle
    ps = scene.describe()
    threads = os.cpu_count() or 1
    # calibrate on two bands (one in the empty lower half, one in the lit upper half)
    t0 = time.time()
    n = 0
    for y in (160, 480):
        _, st = oracle.render(ps, (W, H), SPP, DEPTH, threads, rows=(y, y + 8))
        n += st.camera_samples
    rate = n / max(time.time() - t0, 1e-6)
    total = W * H * SPP
    bands_all = H // 8
    nb = int(max(2, min(bands_all, budget_s * rate / (8 * W * SPP))))
    nb -= nb % 2  # keep the two halves of the film equally represented
    nb = max(nb, 2)
    step = bands_all / nb
    ys = sorted({int(i * step) * 8 for i in range(nb)})
    samples = 0
    secs = 0.0
    verts = 0
    for y in ys:
        _, st = oracle.render(ps, (W, H), SPP, DEPTH, threads, rows=(y, y + 8))
        samples += st.camera_samples
        secs += st.seconds
        verts += st.vertices
    return {
        "value": samples / secs * 1e-6, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{len(ys)} of {bands_all} 8-row bands of the same 640x640x256spp depth-8 render "
                  f"({samples / total:.3f} of the workload, {secs:.1f} s), evenly spaced over the film",
        "vertices_per_sample": verts / samples,
    }


def cpu_baseline_reference(scene, gpu_md5, budget_s=40.0):
    """The REAL reference timed beside the GPU: oracle/_ref/pine_ref is pine's own PathIntegrator + BVH,
    compiled from the reference's sources by oracle/Makefile in the build container (the binary travels,
    the sources do not).  Renders the same scene description; the whole C2 workload when a 16-spp
    calibration run says it fits the budget, else the largest power-of-two spp that does.  Returns None when
    the binary is absent or fails (the caller then reports the CPU restatement, kind "port")."""
    import hashlib
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_ref", "pine_ref")
    if not os.access(exe, os.X_OK):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            sp, fp = os.path.join(tmp, "s.pscene"), os.path.join(tmp, "s.film")
            open(sp, "w").write(scene.describe())

            def run(spp, limit):
                r = subprocess.run([exe, "render", sp, str(spp), str(DEPTH), fp], capture_output=True, text=True,
                                   timeout=limit)
                if r.returncode != 0:
                    raise RuntimeError(r.stderr[-300:])
                return json.loads(r.stdout.strip().splitlines()[-1])
            cal = run(16, 120)
            spp = SPP
            while spp > 16 and cal["seconds"] * spp / 16 > budget_s:
                spp //= 2
            res = cal if spp == 16 else run(spp, 4 * budget_s + 60)
            md5 = hashlib.md5(open(fp, "rb").read()).hexdigest()
        whole = spp == SPP
        return {
            "value": res["msamples_per_s"], "unit": "Msamples/s", "cores": res["threads"], "kind": "reference",
            "sample": (f"the whole workload (640x640x{spp}spp depth {DEPTH}) rendered by oracle/_ref/pine_ref = the reference's own "
                       f"PathIntegrator(BVH, BlueSampler, UniformLightSampler) in {res['seconds']:.1f} s" if whole else
                       f"640x640x{spp}spp depth {DEPTH} ({spp}/{SPP} of the samples per pixel, {res['seconds']:.1f} s) rendered by "
                       f"oracle/_ref/pine_ref = the reference's own PathIntegrator(BVH, BlueSampler, UniformLightSampler)"),
            "film_md5": md5,
            "film_equals_gpu": (md5 == gpu_md5) if whole else None,
        }
    except Exception as e:  # the reference binary is optional test infrastructure: report, do not fail the bench
        print(f"[bench] reference CPU baseline unavailable: {e}", file=sys.stderr)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--spi", type=int, default=0, help="samples per work item (0 = auto)")
    args = ap.parse_args()

    import torch
    import pine_amd
    from pine_amd import scenes

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

    scene = scenes.cbox((W, H), "committed")
    plan = pine_amd.Plan(scene, SPP, DEPTH, device=local_rank, shard_rank=rank, shard_world=world,
                         samples_per_item=args.spi, timing=True)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # N > 1: every rank writes only its own tiles, tile-major, into a slab (1/N of the film); the
    # slabs are gathered to rank 0 (direct sends over xGMI, 6.55/N MB each) and scattered into the
    # row-major film there.  PINE_BENCH_COLLECTIVE=reduce selects the simpler form instead: full-size
    # zero-initialised films summed to rank 0 (exact: x + 0), 6.55 MB per rank.
    collective = os.environ.get("PINE_BENCH_COLLECTIVE", "gather") if use_dist else "none"
    if collective == "gather":
        slab = torch.empty(plan.slab_floats(), dtype=torch.float32, device="cuda")
        slabs = torch.empty((world, plan.slab_floats()), dtype=torch.float32, device="cuda") if rank == 0 else None
        slabs_c = slabs if cdev == "cuda" else (torch.empty(slabs.shape, dtype=torch.float32) if rank == 0 else None)
        slab_list = list(slabs_c.unbind(0)) if rank == 0 else None

    def step():
        if collective == "gather":
            plan.launch_packed(slab.data_ptr(), stream)
            dist.gather(slab if cdev == "cuda" else slab.cpu(), slab_list, dst=0)
            if rank == 0:
                if cdev != "cuda":
                    slabs.copy_(slabs_c)
                pine_amd.film_unpack((W, H), world, slabs.data_ptr(), film.data_ptr(), local_rank, stream)
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
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    plan.stats()  # (discard the warm-up launches' kernel timings)
    t0 = time.perf_counter()
    for _ in range(args.steps):
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
    spp_eff = st.spp_effective
    local_samples = st.camera_samples
    verts = st.vertices
    if use_dist:
        t = torch.tensor([local_samples, verts], dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        total_samples, total_verts = float(t[0]), float(t[1])
    else:
        total_samples, total_verts = float(local_samples), float(verts)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_samples * args.steps / dt * 1e-6
        vbar = total_verts / total_samples
        # roofline of the dominant kernel (path_trace_kernel) on this rank: algorithmic bytes per
        # launch = (192 B * vertices + 16 B * pixels) of this rank's shard / average launch duration
        k_ms = st.trace_ms
        kernel = "path_queue_kernel" if st.block_threads == 1024 else "path_trace_kernel"
        # HBM-side traffic of that kernel per launch: measured off-line with rocprofv3 PMC passes
        # (tools/profile_round.sh -> profiles/rNN_traffic.json, latest round); null if not measured
        # for this kernel / this GPU count
        traffic = None
        tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_traffic.json")))
        if tfiles and world == 1:
            tj = json.load(open(tfiles[-1]))
            if tj.get("kernel", "").startswith(kernel):
                traffic = tj["traffic_bytes_per_launch"]
        alg_bytes = B_VERTEX * verts + 16.0 * local_samples / spp_eff
        achieved = alg_bytes / (k_ms * 1e-3) * 1e-9
        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (scenes/cbox.pine geometry rebuilt through the API; sampler tables are the published BlueSobol data)",
            "config": {"workload": "cbox 640x640 256spp depth=8, as-committed camera, BlueSampler, pine-BVH order",
                       "film": [W, H], "spp_effective": spp_eff, "max_path_length": DEPTH,
                       "parallelism": f"tiles8x8-roundrobin x{world}", "collective": collective, "samples_per_item": st.samples_per_item,
                       "grid_blocks": st.grid_blocks, "vertices_per_sample": vbar},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (rocprofv3 2*FETCH_SIZE+WRITE_SIZE, fabric side incl. Infinity-Cache hits)",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel": kernel, "kernel_ms": k_ms,
                         "bytes_per_vertex": B_VERTEX, "note": "algorithmic bytes of the streaming formulation (SURVEY.md 8(d)); the kernel keeps path state in LDS and the fold stack in L2/Infinity Cache, so it is latency/VALU bound, not HBM bound"},
            "film_md5": __import__("hashlib").md5(film.cpu().numpy().tobytes()).hexdigest(),
            "kernels_ms": {"prepass": st.prepass_ms, "path_trace": k_ms, "resolve": st.resolve_ms, "launches_averaged": st.timed_launches},
        }
        if world == 1 and not args.no_cpu:
            port = cpu_baseline(scene)
            ref = cpu_baseline_reference(scene, out["film_md5"])
            cb = ref if ref else port
            out["cpu_baseline"] = cb
            if ref:
                out["cpu_baseline_port"] = port  # this repo's CPU restatement of the same path, for comparison
            out["speedup_vs_cpu"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
