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
