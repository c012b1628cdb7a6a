"""bench.py must end with ONE parseable JSON line whatever happens to the distributed run: a failing
init_process_group / RCCL error / hung collective may cost the numbers, never the line.

CPU part (no GPU here): the watchdog, and a 2-rank launch whose ranks cannot even select a device.  GPU part: two ranks forced onto the
one GPU of the box with the RCCL backend (which refuses two ranks per device) — rank 0 must report `comm.error` next to its LOCAL
single-GPU numbers."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}:\n{stdout[-2000:]}"
    return json.loads(lines[0])


def test_line_is_compact():
    """BENCH_r04.parsed was null because the line had grown to 23 KB.  The stdout line is a pure function of the full record
    (bench.compact_line); built from the round-4 record — and from one inflated well past it — it must stay under 7000 bytes (the driver keeps an
    ~8 KB tail of stdout) and keep the contract's fields, `roofline` and `cpu_baseline`."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r4_n", "bench.json")))
    assert len(json.dumps(full)) > 20000          # the record that did not parse
    for inflate in (False, True):
        rec = json.loads(json.dumps(full))
        if inflate:
            rec["passes"]["note"] = "x" * 20000
            rec["roofline"]["note"] = "y" * 5000
            rec["config"]["workload"] += " " + "z" * 3000
            rec["cpu_baseline"]["sample"] = "s" * 4000
            rec["error"] = "e" * 4000
            rec["stages"].update({f"extra_{i}": dict(rec["stages"]["ray_trace"]) for i in range(40)})
        line = json.dumps(bench.compact_line(rec))
        assert len(line.encode()) < bench.LINE_LIMIT <= 7000, len(line)
        d = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert k in d, k
        assert d["config"]["workload"].startswith("1920x1080")
        assert d["roofline"]["bound"] in ("hbm", "mfma", "valu", "latency") and d["roofline"]["frac"] > 0 and d["roofline"]["peak"] == 8000.0 and "traffic" in d["roofline"]
        assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["sample"]
        if not inflate:
            assert set(d["passes"]["4k"]) == set(d["passes"]["1080p"]) == {"shadows", "ao", "reflections", "ddgi"}
            assert all(set(v) >= {"ms", "frac", "bound"} for v in d["passes"]["4k"].values())
            assert d["passes"]["hybrid_frame_ms"]["4k"]["serial"] > 0 and d["passes"]["hard_tier"]["ms_per_frame"] > 0
    # a record with nothing in it (the watchdog's stub) goes through as well
    assert json.loads(json.dumps(bench.compact_line({"metric": "m", "value": 0.0, "error": "boom"})))["error"] == "boom"


def test_stdout_carries_only_the_line(tmp_path):
    """what a library writes to stdout behind Python's back — through C stdio, where it sits in libc's buffer until the process exits (RCCL's
    banner in a --gpus N run) — must not follow the JSON line: bench.isolate_stdout() points descriptor 1 at stderr and emit() writes the line to
    the real stdout"""
    code = ("import sys, ctypes\nsys.path.insert(0, %r)\nimport bench\nbench.ROOT = %r\nbench.isolate_stdout()\n"
            "ctypes.CDLL(None).printf(b'banner via C stdio\\n')\nprint('a python print')\nbench.emit({'metric': 'm', 'value': 1.0})\n" % (ROOT, str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    assert out.stdout.count("\n") == 1 and json.loads(out.stdout)["value"] == 1.0, out.stdout
    assert "banner via C stdio" in out.stderr and "a python print" in out.stderr


def test_watchdog_prints_the_line_and_exits_nonzero():
    env = dict(os.environ, HR_BENCH_TIMEOUT_S="0.3")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 3, (out.returncode, out.stderr[-1000:])
    d = _last_json(out.stdout)
    assert "watchdog" in d["error"] and d["comm"]["stage"] and d["metric"] and d["unit"] == "Mrays/s" and d["n_gpus"] == 1
    assert "watchdog: rank 0/1" in out.stderr


def test_two_ranks_that_cannot_start_still_give_one_line():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU part: on a GPU box test_gpu_two_ranks_on_one_gpu_report_comm_error covers the same path with RCCL")
    env = dict(os.environ, HR_DIST_BACKEND="gloo", HR_BENCH_TIMEOUT_S="240")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-passes", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=400)
    d = _last_json(out.stdout)
    assert d["comm"]["error"] and d["requested_gpus"] == 2, d
    assert "error" in d   # no GPU here: the local fallback cannot measure anything either, and says so


@pytest.mark.gpu
def test_gpu_two_ranks_on_one_gpu_report_comm_error():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box (RCCL refuses two ranks on one device)")
    # RCCL mostly refuses at once ("duplicate GPU"), but has been seen to sit in its bootstrap for minutes on a loaded box: the watchdog is
    # set short, and BOTH endings are accepted — the refusal (local fallback numbers) and the watchdog (the line as far as it is known)
    env = dict(os.environ, HR_DIST_BACKEND="nccl", HR_FORCE_DEVICE="0", HR_BENCH_TIMEOUT_S="90")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29732",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-passes", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=400)
    d = _last_json(out.stdout)
    assert d["comm"]["error"], d
    if "watchdog" not in d["comm"]["error"]:
        assert d["requested_gpus"] == 2 and d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0, d


@pytest.mark.gpu
def test_gpu_bench_line_as_the_driver_runs_it():
    """the driver's own command (BENCH_rNN: `python3 bench.py --gpus 1 --steps 20 --warmup 5`): the LAST stdout line is one JSON object < 7000 bytes
    with the contract's fields, a live `roofline` and a `cpu_baseline`; the full record sits in bench_detail.json"""
    env = dict(os.environ, HR_BENCH_CPU_SECONDS="2", HR_BENCH_TEST_C_STDOUT="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # stdout is the line and nothing else: what libraries write through C stdio (RCCL's banner sits in libc's buffer until the process exits,
    # i.e. AFTER the line) goes to stderr (bench.isolate_stdout)
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout[-1500:]
    assert "a library's banner" in out.stderr
    last = out.stdout.strip().splitlines()[-1]
    assert len(last.encode()) < 7000, len(last)
    d = _last_json(out.stdout)
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu", "latency") and 0 < r["frac"] and r["achieved"] > 0 and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["frac_is_requested_bytes"] is True
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["config"]["nodes_per_ray"] <= d["config"]["nodes_per_ray_full_walk"]      # the occluder cache only ever shortens the walk
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"]
    assert set(d["passes"]["4k"]) == {"shadows", "ao", "reflections", "ddgi"}
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert detail["value"] == d["value"] and "kernels" in detail["passes"]["ao"]


def test_an_eight_gpu_line_keeps_the_fields_the_scale_record_needs():
    """first contact with a real 8-GPU node must not be lost to formatting (r04's 1-GPU line was): a record shaped like the one
    `bench.py --gpus 8` assembles — the round-4 full record + the `comm` block from bench_report.comm_block (the function bench.py itself calls)
    + the strong-scaling `hybrid_4k` block — goes through compact_line(): n_gpus, comm.ranks_seen, comm.band_bounds, comm.exchange_us_per_frame,
    hybrid_4k.comm_us_per_frame survive, the contract's fields are there, and the line stays under 7000 bytes"""
    import bench, bench_report
    full = json.load(open(os.path.join(ROOT, "profiles", "r4_n", "bench.json")))
    bounds = [0, 520, 980, 1370, 1720, 2060, 2400, 2750, 3056]           # nine edges of eight cost-balanced bands of a 5432 x 3056 frame
    full.update(n_gpus=8, requested_gpus=8, scaling="weak")
    full["config"] = dict(full["config"], workload="5432x3056 procedural Sponza-like (277968 tris) ray-traced shadows 1spp + SVGF denoise, 8x the pixels of 1920x1080 (same view), "
                                                   "row-tiled into 8 cost-balanced bands")
    full["comm"] = bench_report.comm_block("nccl", 8, bounds, 40, 37.4, 5432)
    full.pop("passes", None)
    full["hybrid_4k"] = {"workload": "ONE 3840x2160 hybrid frame row-tiled over the GPUs", "n_gpus": 8, "ms_per_frame": 0.61, "frames_per_s": 1639.3, "Mrays_per_s": 9100.0,
                         "bands": [[0, 270], [270, 540], [540, 810], [810, 1080], [1080, 1350], [1350, 1620], [1620, 1890], [1890, 2160]], "scaling": "strong", "forked_streams": True,
                         "comm_us_per_frame": {"history_exchange": 41.0, "ddgi_all_gather": 18.5, "note": "x" * 500}}
    line = json.dumps(bench.compact_line(full))
    assert len(line.encode()) < bench.LINE_LIMIT, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "dropped" not in d
    assert d["comm"]["ranks_seen"] == 8 and d["comm"]["band_bounds"] == bounds and d["comm"]["exchange_us_per_frame"] == 37.4 and d["comm"]["backend"] == "nccl"
    assert d["hybrid_4k"]["n_gpus"] == 8 and d["hybrid_4k"]["comm_us_per_frame"] == {"history_exchange": 41.0, "ddgi_all_gather": 18.5} and len(d["hybrid_4k"]["bands"]) == 8
