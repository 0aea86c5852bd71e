"""bench.py's stdout contract: ONE JSON line under 4 KB that round-trips through json.loads and carries `roofline` and
`cpu_baseline` (round 3's 21.9 KB line could not be parsed by the driver), and a `--gpus N` entry that really starts N ranks
or fails loudly."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _roofline(frac=0.66, long_strings=False):
    return {"bound": "hbm", "achieved": 5283.123456789, "peak": 8000.0, "unit": "GB/s", "frac": frac, "traffic": 2.24e10,
            "algorithmic_bytes_per_launch": 53212345678.0, "kernel": "pine_scene_kernel_131330" + ("x" * 400 if long_strings else ""), "kernel_ms": 10.0712345,
            "bytes_per_vertex": 192.0, "valu_issue_frac": 0.8123456, "valu_lane_utilisation": 0.79123, "wait_any_frac": 0.32, "salu_issue_frac": 0.4,
            "counters_from": "r04_counters.json@0123456789ab" + (" (STALE: kernels changed since)" if long_strings else ""), "stale": long_strings}


def _entry(name, long_strings=False):
    pad = "y" * 700 if long_strings else ""
    return {"config": name, "name": name, "workload": "cbox 640x640 256spp depth=8, as-committed camera, BlueSampler, pine-BVH order" + pad, "film": [640, 640],
            "spp_effective": 256, "max_path_length": 8, "value": 10030.123456789, "unit": "Msamples/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 10.4512345, "clock": "launch -> film in pinned host memory" + pad, "parallelism": "tiles8x8-roundrobin x1", "collective": "none",
            "overlap_side_stream": True, "kernel_mode": "specialised(cache: cold, compiled in 1.7 s during warm-up; scene baked in)" + pad,
            "kernel_mode_short": "specialised", "compile_wait_s": 1.7, "samples_per_item": 2, "serial_tiles": 0, "grid_blocks": 256,
            "vertices_per_sample": 2.6425, "walk_steps_per_sample": 0.0, "roofline": _roofline(long_strings=long_strings), "film_md5": "0" * 32,
            "reference_md5": "0" * 32, "film_equals_reference": True,
            "kernels_ms": {"prepass": 0.3, "path_trace": 10.07, "resolve": 0.33, "launches_averaged": 20},
            "plan_ms": {"accel_build_host": 0.01, "upload": 20.0, "specialize": 1700.0, "note": "n" * (500 if long_strings else 10)}}


def _full(long_strings=False, nconfigs=4):
    full = {"n_gpus": 1, "steps": 20, "warmup": 5, "headline": _entry("c2", long_strings),
            "configs": [_entry(n, long_strings) for n in ("c2r", "c3", "c4", "c5", "c6", "c7", "c8", "c9", "c10", "c11")[:nconfigs]],
            "detail_file": "bench_detail.json", "git_kernel_source_hash": "0123456789ab", "notes": {"x": "z" * 3000},
            "device_resident": {"value": 10300.1, "ms_per_step": 10.18, "steps": 10, "kernels_ms": {}, "clock": "c"},
            "precompiled": {"value": 7779.0, "ms_per_step": 13.48, "steps": 10, "kernel_ms": 12.9, "roofline_frac": 0.51, "roofline": _roofline(0.51),
                            "film_equals_reference": True},
            "cold": {"default_mode": {"ms": 14.1}, "precompiled": {"ms": 15.0}, "what": "w" * 600},
            "fast_mode": [{"config": "c2_fast", "value": 1.0}],
            "cpu_baseline": {"value": 9.6229, "unit": "Msamples/s", "cores": 256, "kind": "reference",
                             "sample": "whole workload 640x640x256spp d8 by oracle/_ref/pine_ref in 10.9 s" + ("s" * 800 if long_strings else ""),
                             "what": "w" * 500, "film_md5": "0" * 32, "film_equals_gpu": True},
            "cpu_baseline_port": {"value": 9.9, "cores": 256, "kind": "port", "sample": "p" * 300}}
    for e in full["configs"]:
        e["precompiled"] = {"value": 1.0, "ms_per_step": 2.0}
    return full


@pytest.mark.parametrize("long_strings, nconfigs", [(False, 4), (True, 4), (True, 10), (False, 0)])
def test_the_line_is_small_parses_and_carries_the_contract(long_strings, nconfigs):
    import bench
    text = bench.compact_line(_full(long_strings, nconfigs))
    assert "\n" not in text and len(text) < 4096, len(text)
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "Msamples/s" and d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("cbox 640x640") and "model" not in d["config"] and d["config"]["kernel_mode"].startswith("specialised(")
    rl = d["roofline"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    assert {"kernel", "kernel_ms", "algorithmic_bytes_per_launch", "traffic", "valu_issue_frac"} <= set(rl)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 256 and cb["value"] > 0 and cb["unit"] == "Msamples/s" and cb["sample"]
    assert abs(d["speedup_vs_cpu"] - d["value"] / cb["value"]) / d["speedup_vs_cpu"] < 1e-3
    if not long_strings and nconfigs:
        assert [c["name"] for c in d["configs"]] == ["c2r", "c3", "c4", "c5"] and d["precompiled"]["value"] == 7779.0


def test_a_line_without_cpu_baseline_or_side_figures_still_parses():
    import bench
    full = _full()
    for k in ("cpu_baseline", "cpu_baseline_port", "precompiled", "device_resident", "cold", "fast_mode"):
        full.pop(k)
    full["configs"] = [{"config": "c3", "error": "e" * 500}]
    full["headline"]["roofline"].update(traffic=None, valu_issue_frac=None, counters_from=None)
    d = json.loads(bench.compact_line(full))
    assert d["roofline"]["traffic"] is None and d["configs"][0]["error"] and "cpu_baseline" not in d


def _run(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_n_without_a_launcher_starts_n_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` with no RANK in the environment: never a silent one-GPU run that prints n_gpus 2.  On a box
    with fewer devices it refuses; in rehearsal mode (every rank on device 0, gloo) it starts two rank processes -- which, without
    any GPU, both say so and the launcher exits non-zero."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box: the launcher would really run")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and "only" in r.stderr and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    if torch.cuda.is_available():
        return  # (the rehearsal itself runs in the gpu-marked test below)
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-configs", "--no-cpu"], {"PINE_BENCH_DEVICE": "0", "PINE_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") == 2, r.stderr[-2000:]  # two ranks were started
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_two_rank_rehearsal_through_the_launcher_prints_one_parsable_line(tmp_path):
    """The N-rank code path end to end on ONE GPU (both ranks on device 0, collectives through gloo): the launcher starts two
    ranks, rank 0 prints one line with n_gpus 2 whose film equals the reference's."""
    detail = tmp_path / "detail.json"
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-configs", "--no-cpu", "--detail", str(detail)],
             {"PINE_BENCH_DEVICE": "0", "PINE_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["collective"] == "gather" and d["film_equals_reference"] is True
    assert json.load(open(detail))["headline"]["n_gpus"] == 2


@pytest.mark.gpu
def test_the_default_line_on_the_gpu(tmp_path):
    detail = tmp_path / "detail.json"
    r = _run(["--steps", "2", "--warmup", "1", "--no-configs", "--no-cpu", "--detail", str(detail)], {}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["film_equals_reference"] is True and d["roofline"]["frac"] > 0.2
    assert d["config"]["kernel_mode"].startswith(("specialised(", "precompiled"))
