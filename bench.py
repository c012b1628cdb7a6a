#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native ray-trace + denoise hot path.

One "step" = one frame of RayTracedShadows::render (1 spp soft-shadow trace + SVGF temporal + 4 x a-trous) on synthetic 1080p
G-buffers of the procedural Sponza-like scene (~278k triangles), inputs resident in HBM (BASELINE.json configs[1]), in the
shipping arithmetic mode (hr_shadows_params.exact = 0: bit-exact masks, fp16 images within the stated tolerance; the bit-for-bit
parity mode is timed next to it and reported as `exact_mode`).  Prints ONE JSON line (see the driver contract).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Besides the headline the line carries (outside the timed region, short runs):
  passes       N = 1: BASELINE configs[2], [3] and the passes of [4] on one GPU — AO 4 spp, half-res reflections, DDGI 16x8x16x256, the
               whole hybrid frame at 1080p and at 4K: ms, Mrays/s, frames/s and, per kernel, the HIP-event time and the fraction of the
               8 TB/s HBM roofline its ALGORITHMIC bytes amount to;
  hybrid_4k    N > 1: BASELINE configs[4] itself — ONE 3840x2160 hybrid frame row-tiled over the N GPUs (strong scaling), max over ranks;
  roofline     dominant kernel of the headline; `frac` = algorithmic bytes / time / peak, `dram_frac` = rocprofv3 PMC traffic / time / peak
               and `bound` = what the committed SQ counters of this build say limits it (profiles/, tools/profile_round.sh);
  cpu_baseline the SAME ray batch replayed through the oracle's scalar BVH traversal on the host cores (`trace_replay`) and the oracle's
               denoise chain timed separately (`denoise_ms`), BASELINE.md §3.

N > 1 (weak scaling of the headline): the SAME view is rendered with N times the pixels — 1920*sqrt(N) x 1080*sqrt(N), rounded to
multiples of 8 — so rays and pixels per frame grow with N while the content statistics stay those of the N = 1 frame.
The frame is row-tiled, one band per GPU; band boundaries are chosen from a calibration frame so that every band
carries the same share of the cost model  pixels + 1.8 * rays  (tiling.balanced_bounds: sky rows fire no ray, the
floor fires one per pixel — equal-height bands would leave most GPUs idle).  Every band re-traces / re-filters 24 halo
rows locally and, once per frame, exchanges the 40 history rows next to each band boundary with its neighbours over
RCCL (hybrid_rendering_amd/tiling.py), overlapped with the next frame's trace; band rows are bit-identical to the
single-GPU result in exact mode (tests/test_gpu_tiling.py).  `value` counts only the rays of band rows (halo work is overhead).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import re
import sys
import threading
import time

import numpy as np

# kernel arguments in device memory (the HIP runtime's default on this image; measured: headline 0.196 ms with it, 0.205 ms without —
# the pass kernels take 200-500 byte argument blocks).  Must be set before the HIP runtime initialises.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
NODE_BYTES, TRI_BYTES = 80, 48
# stage name of a pass profiler -> EXACT kernel instance (normalised rocprofv3 name: no "void", namespaces or argument list), so that
# every a-trous step / template instance carries its own counters (a substring match gave steps 1..8 step 1's traffic)
def kernel_of(pass_, stage, exact):
    step = {"atrous_0": 1, "atrous_1": 2, "atrous_2": 4, "atrous_3": 8}
    if pass_ == "shadows":
        if stage == "ray_trace": return "k_shadows_trace<false>"
        if stage == "temporal_accumulation": return "k_shadows_temporal" if exact else "kf_shadows_temporal<1>"   # <1>: reprojection from the pass's geometry records
        if stage == "atrous_01": return "kf_shadows_atrous01<16, true>"
        if stage in step:
            if exact: return "k_shadows_atrous<1, true>"       # one instance serves the four iterations in the parity mode
            return ("kf_shadows_atrous_lds<%d, true>" if step[stage] <= 2 else "kf_shadows_atrous<%d, true>") % step[stage]
    if pass_ == "ao":
        if stage == "ray_trace": return "k_ao_trace<false>"
        if stage == "temporal_accumulation": return "k_ao_temporal<true>" if exact else "kf_ao_temporal<true, 2>"
        if stage == "blur_xy": return "kf_ao_blur_xy<4, 16>"
        if stage in ("blur_x", "blur_y"): return "k_ao_blur<4>" if exact else "kf_ao_blur<4>"   # two launches of one instance: the counters average X and Y
    if pass_ == "ddgi":
        return {"ray_trace": "k_ddgi_trace<false>", "irradiance_probe_update": "k_ddgi_probe_update<false, false>", "depth_probe_update": "k_ddgi_probe_update<true, true>",
                "sample_probe_grid": "k_ddgi_sample" if exact else "kf_ddgi_sample"}.get(stage)
    if pass_ == "reflections":
        if stage == "ray_trace": return "k_refl_trace<false, false>" if exact else "k_refl_trace<false, true>"   # <., FAST>: tolerance-mode irradiance gathers
        if stage == "temporal_accumulation": return "k_refl_temporal" if exact else "kf_refl_temporal<1>"
        if stage == "atrous_01": return "kf_refl_atrous01<16, true>"
        if stage in step: return "k_refl_atrous<1>" if exact else "kf_refl_atrous<%d, true>" % step[stage]
        if stage == "upsample": return "k_upsample<4>" if exact else "kf_upsample<4>"
    return None


def norm_kernel(name):
    n = re.sub(r"^void\s+", "", name.strip())
    n = n.replace("(anonymous namespace)::", "").replace("hr::", "")
    return n.split("(")[0].strip()


N_SIMD = 1024            # 256 CUs x 4 SIMDs
N_XCD = 8                # GRBM_GUI_ACTIVE is summed over the 8 XCDs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--detail", type=float, default=1.0, help="scene tessellation (1.0 = ~278k triangles)")
    ap.add_argument("--tier", default="standard", choices=["standard", "hard"], help="hard: the ~2.5 M triangle scene with layered fabric, foliage cards and a grazing sun as the headline workload")
    ap.add_argument("--obj", default=None, help="Wavefront OBJ (+MTL) to render instead of the procedural scene (hybrid_rendering_amd/assets.py)")
    ap.add_argument("--ring", type=int, default=8, help="distinct camera positions cycled through")
    ap.add_argument("--exact", type=int, default=0, help="1: time the bit-for-bit parity arithmetic as the headline instead of the tolerance mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-passes", action="store_true", help="skip the `passes` / `hybrid_4k` blocks (profiling runs)")
    ap.add_argument("--cpu-frames", type=int, default=4)
    return ap.parse_args()


def load_profile(suffix=""):
    """latest profiles/r*/ directory of this build that holds the counters for this frame size (suffix "" = 1080p, "_4k" = 3840x2160;
    tools/profile_round.sh): per-kernel PMC traffic, raw SQ counters and rocprofv3's average duration, keyed by the EXACT normalised
    kernel name.  PMC cannot be sampled from inside this process: the files are committed with the build, and every number taken
    from them is marked stale when the kernel's live HIP-event time is more than 10 % off the profiled duration.
      traffic     = 2 * FETCH_SIZE + WRITE_SIZE   FETCH_SIZE tallies every L2 -> fabric read request at 64 B; a streaming read's requests
                                                  are 128 B (profiles/r3_calib: streams read back exactly 1/2 of their bytes, writes 1/1)
      traffic_lo  = FETCH_SIZE + WRITE_SIZE       a sparse gather's requests are 64 B (r3_calib: 4 B gathers, one per line, tally 64 B per
                                                  lane; two lanes on the two halves of a line tally 64 B per pair): BVH-walking kernels lie
                                                  between the two
      valu_issue_frac  = 4 * SQ_INSTS_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)   share of the SIMD cycles a VALU instruction issues in
      lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)             active lanes per issued VALU instruction
    (the derived VALUBusy of this rocprofv3 falls back to gfx94x formulas and exceeds 100 %: not used)"""
    prof = {"dir": None, "traffic": {}, "traffic_lo": {}, "sq": {}, "avg_us": {}, "suffix": suffix}
    try:
        pd = os.path.join(ROOT, "profiles")
        dirs = sorted(d for d in os.listdir(pd) if os.path.exists(os.path.join(pd, d, f"pmc_summary{suffix}.json")))
        if not dirs:
            return prof
        d = os.path.join(pd, dirs[-1])
        prof["dir"] = "profiles/" + dirs[-1]
        for k, v in json.load(open(os.path.join(d, f"pmc_summary{suffix}.json"))).items():
            if "FETCH_SIZE_KB_avg_per_launch" in v and "WRITE_SIZE_KB_avg_per_launch" in v:
                f_, w_ = v["FETCH_SIZE_KB_avg_per_launch"] * 1024, v["WRITE_SIZE_KB_avg_per_launch"] * 1024
                prof["traffic"][norm_kernel(k)] = int(2 * f_ + w_)
                prof["traffic_lo"][norm_kernel(k)] = int(f_ + w_)
        sq = os.path.join(d, f"sq_counters{suffix}.json")
        if os.path.exists(sq):
            prof["sq"] = {norm_kernel(k): v for k, v in json.load(open(sq)).items()}
        import csv
        for fn in (f"kernel_stats{suffix}.csv", f"frame_kernel_stats{suffix}.csv"):   # the frame file wins: same command as the counters
            fp = os.path.join(d, fn)
            if os.path.exists(fp):
                for row in csv.DictReader(open(fp)):
                    prof["avg_us"][norm_kernel(row["Name"])] = float(row["AverageNs"]) / 1e3
    except Exception as e:
        prof["error"] = repr(e)[:200]
    return prof


def classify(prof, kernel, ms, alg_bytes, gather=False):
    """-> dict(frac, traffic, dram_frac, valu_issue_frac, lane_utilisation, valu_frac, bound, ...) for ONE kernel instance (exact name).
    bound: `valu` if VALU instructions issue in > 70 % of the SIMD cycles, `hbm` if the counter traffic moves at > 50 % of peak, else
    `latency` (dependent fetches / too little in flight).  valu_frac = issue share x lane utilisation = the part of the VALU roof that does
    useful work; it is the operative roofline figure of a `valu` kernel (an HBM fraction says little about it)."""
    out = {"kernel": kernel, "frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 and alg_bytes else None}
    tr, sq, avg = prof["traffic"].get(kernel), prof["sq"].get(kernel), prof["avg_us"].get(kernel)
    state = None
    if avg is not None and ms > 0:
        out["profile_avg_us"] = round(avg, 2)
        state = "fresh" if abs(ms * 1e3 - avg) <= 0.10 * avg + 3.0 else "stale"   # + 3 us: the HIP-event pair includes the launch gap
        out["profile_state"] = state
    if tr is not None:
        out["traffic"] = tr
        out["dram_frac"] = round(tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None
        if gather:
            out["traffic_lo"] = prof["traffic_lo"].get(kernel)
    if sq and sq.get("SQ_INSTS_VALU") and sq.get("GRBM_GUI_ACTIVE"):
        raw = 4.0 * sq["SQ_INSTS_VALU"] / (N_SIMD * sq["GRBM_GUI_ACTIVE"] / N_XCD)
        # the 4-cycles-per-wave64-instruction model over-counts kernels with long EXEC-masked stretches (a fully masked VALU instruction
        # retires faster): raw values of 1.0-1.4 were measured on the trace and probe-update kernels; they mean "saturated"
        out["valu_issue_frac"] = round(min(raw, 1.0), 3)
        if raw > 1.0:
            out["valu_issue_raw"] = round(raw, 3)
    if sq and sq.get("SQ_ACTIVE_INST_VALU"):
        out["lane_utilisation"] = round(sq.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * sq["SQ_ACTIVE_INST_VALU"]), 3)
    if "valu_issue_frac" in out and "lane_utilisation" in out:
        out["valu_frac"] = round(out["valu_issue_frac"] * out["lane_utilisation"], 3)
    if "valu_issue_frac" not in out and tr is None:
        out["bound"] = None
    elif out.get("valu_issue_frac", 0.0) > 0.70:
        out["bound"] = "valu"
    elif (out.get("dram_frac") or 0.0) > 0.5:
        out["bound"] = "hbm"
    else:
        out["bound"] = "latency"
    return out


# ---- one JSON line, whatever happens (first multi-GPU contact must not be able to end without a parseable line) -------------
PROGRESS = {"stage": "start", "rank": 0, "world": 1}     # where this rank is (the watchdog prints it)
PARTIAL = {}                                             # rank 0: the line as far as it is known
_printed = threading.Event()


def stage(name):
    PROGRESS["stage"] = name


LINE_LIMIT = 7000        # bytes: the driver keeps an ~8 KB tail of stdout; round 4's 23 KB line was not parsed (BENCH_r04.parsed = null)
DETAIL_FILE = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


def _short(s_, n=160):
    return s_ if not isinstance(s_, str) or len(s_) <= n else s_[:n - 1] + "…"


_PASS_KEYS = ("ms", "frac", "dram_frac", "valu_frac", "bound")


def _pass_summary(entry, roof):
    """{ms, frac, dram_frac, valu_frac, bound} of one pass from its pass_roofline() aggregate (+ wall-clock ms / Mrays/s when known)"""
    out = {}
    if isinstance(entry, dict):
        if entry.get("ms_per_frame") is not None:
            out["wall_ms"] = entry["ms_per_frame"]
        if entry.get("Mrays_per_s") is not None:
            out["Mrays_per_s"] = entry["Mrays_per_s"]
    if isinstance(roof, dict):
        out.update(ms=roof.get("ms"), frac=roof.get("frac"), dram_frac=roof.get("dram_frac"), valu_frac=roof.get("valu_frac"), bound=roof.get("binding"))
    return {k: v for k, v in out.items() if v is not None}


def compact_line(full):
    """the ONE stdout line: the contract's fields + config + roofline + cpu_baseline + a compact per-pass summary, < LINE_LIMIT bytes.  Everything
    else (per-kernel blocks, notes, timed-region arrays) lives in bench_detail.json / on stderr.  Pure function of the full record (CPU-tested:
    tests/test_bench_robustness.py::test_line_is_compact)."""
    c = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"))
    c["vs_baseline"] = full.get("vs_baseline")
    if "ms_per_step" not in c:
        c["ms_per_step"] = None
    cfg = full.get("config") or {}
    c["config"] = {k: _short(v, 200) for k, v in cfg.items()}
    for k in ("error", "requested_gpus", "passes_error"):
        if full.get(k) is not None:
            c[k] = _short(full[k], 400)
    c.update(_pick(full, ("timed_repeats", "timed_total_ms", "denoised_frames_per_s", "trace_only_Mrays_per_s")))
    r = full.get("roofline")
    if isinstance(r, dict):
        c["roofline"] = _pick(r, ("kernel", "kernel_name", "bound", "achieved", "peak", "unit", "frac", "traffic", "dram_frac", "valu_issue_frac", "lane_utilisation",
                                  "valu_frac", "binding_frac", "frac_is_requested_bytes", "frac_full_walk", "algorithmic_bytes", "live_event_us", "profile_avg_us",
                                  "profile_state", "counters"))
        c["roofline"].setdefault("traffic", None)
    st = full.get("stages")
    if isinstance(st, dict):
        c["stages"] = {n: _pick(v, _PASS_KEYS) for n, v in st.items() if isinstance(v, dict)}
    for k in ("exact_mode", "tolerance_mode"):
        if isinstance(full.get(k), dict):
            c[k] = _pick(full[k], ("ms_per_step", "value"))
    ps = full.get("passes")
    if isinstance(ps, dict):
        summ = {"1080p": {}, "4k": {}}
        for n in ("shadows", "ao", "reflections", "ddgi"):
            if isinstance(ps.get(n), dict):
                summ["1080p"][n] = _pass_summary(ps[n], ps[n].get("roofline"))
        h4 = ps.get("hybrid_4k_one_gpu") or {}
        for n, roof in (h4.get("roofline") or {}).items():
            summ["4k"][n] = _pass_summary(None, roof)
        def frame(hb):
            return {"serial": hb.get("ms_per_frame"), "streams": (hb.get("concurrent_streams") or {}).get("ms_per_frame"), "graph": (hb.get("hip_graph") or {}).get("ms_per_frame"),
                    "Mrays_per_s": hb.get("Mrays_per_s")} if hb else None
        summ["hybrid_frame_ms"] = {"1080p": frame(ps.get("hybrid_1080p")), "4k": frame(h4)}
        if isinstance(ps.get("reflections_full_res"), dict):
            summ["reflections_full_res"] = _pick(ps["reflections_full_res"], ("ms_per_frame", "Mrays_per_s"))
        if isinstance(ps.get("hard_tier"), dict):
            summ["hard_tier"] = _pick(ps["hard_tier"], ("ms_per_frame", "Mrays_per_s", "trace_only_Mrays_per_s", "nodes_per_ray", "tris_per_ray"))
        summ["keys"] = "per pass: ms = sum of its kernels' HIP-event times, frac = algorithmic bytes / ms / 8 TB/s, dram_frac = counter traffic, valu_frac = issue x lanes, bound = of its longest kernel"
        c["passes"] = summ
    h = full.get("hybrid_4k")
    if isinstance(h, dict):
        c["hybrid_4k"] = _pick(h, ("n_gpus", "ms_per_frame", "frames_per_s", "Mrays_per_s", "bands", "scaling", "forked_streams"))
        cm = h.get("comm_us_per_frame")
        if isinstance(cm, dict):
            c["hybrid_4k"]["comm_us_per_frame"] = {k: v for k, v in cm.items() if k != "note"}
    cm = full.get("comm")
    if isinstance(cm, dict):
        c["comm"] = {k: _short(v, 300) for k, v in cm.items() if k != "note"}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        b = _pick(cb, ("value", "unit", "cores", "kind", "error"))
        if cb.get("sample"):
            b["sample"] = _short(cb["sample"], 260)
        for k, keys in (("trace_replay", ("value", "rays_per_batch", "batches", "seconds", "nodes_per_ray_bvh2")),
                        ("trace_replay_same_tree", ("value", "unit", "cores", "nodes_per_ray", "tris_per_ray", "masks_equal", "error")),
                        ("denoise_ms", ("temporal", "atrous_x4")), ("whole_frames", ("frames_per_s", "Mrays_per_s")),
                        ("reference_shaders", ("value", "unit", "cores", "bit_identical_to_port", "error"))):
            if isinstance(cb.get(k), dict):
                b[k] = _pick(cb[k], keys)
        c["cpu_baseline"] = b
    c["detail"] = DETAIL_FILE
    # belt and braces: should the line still be too long (a future field, a long error), drop the optional blocks, least important first
    for k in ("stages", "exact_mode", "tolerance_mode", "passes", "hybrid_4k", "comm"):
        if len(json.dumps(c)) < LINE_LIMIT:
            break
        c.pop(k, None)
        c["dropped"] = c.get("dropped", []) + [k]
    return c


_LINE_FD = None   # isolate_stdout(): the descriptor the ONE line goes to


def isolate_stdout():
    """stdout must carry exactly one line, and it must be the last thing on it.  Libraries write there behind Python's back — RCCL prints a five-line
    banner through C stdio at its first communicator, which sits in libc's buffer until the process EXITS, i.e. lands after the JSON line of a
    `--gpus N` run (and every other rank's copy lands in the launcher's merged stdout whenever that rank exits).  So every rank points descriptor 1
    at stderr for the whole run and keeps the real stdout aside for emit()."""
    global _LINE_FD
    if _LINE_FD is not None:
        return
    try:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)
    except OSError:
        _LINE_FD = None


def flush_c_stdio():
    """whatever libraries have written through C stdio so far leaves libc's buffer NOW (before the line), not when the process exits (after it) —
    for a driver that reads stdout and stderr as one stream"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def emit(out):
    """prints the bench line once per process: the full record goes to bench_detail.json (next to this script and, when that directory exists,
    under gpurun_out/) and to stderr; stdout gets ONE compact line (compact_line)"""
    if _printed.is_set():
        return
    _printed.set()
    full = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    f.write(full + "\n")
        except OSError:
            pass
    print("[bench detail] " + full, file=sys.stderr, flush=True)
    try:
        line = json.dumps(compact_line(out))
    except Exception as e:   # never lose the line to its own summariser
        line = json.dumps({**stub_line_from(out), "error": f"compact_line failed: {e!r}"[:300]})
    sys.stdout.flush()
    flush_c_stdio()
    if _LINE_FD is not None:
        os.write(_LINE_FD, (line + "\n").encode())
    else:
        print(line, flush=True)


def stub_line_from(out):
    return {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")}


def stub_line(args, world, error):
    """the contract's fields with no measurement in them: what rank 0 prints when nothing could be measured"""
    return {"metric": "shadow Mrays/s over the fully denoised frame (1 spp trace + SVGF temporal + 4x a-trous)", "value": 0.0, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": f"{args.width}x{args.height} procedural Sponza-like ray-traced shadows 1spp + SVGF denoise"}, "error": error}


def start_watchdog(args, rank, world):
    """HR_BENCH_TIMEOUT_S (default 900): a rank that has not finished by then says where it is (stage, last collective it posted), rank 0 prints
    the line as far as it is known with `error`, and the process exits with code 3 instead of hanging in a collective its peer never posts."""
    limit = float(os.environ.get("HR_BENCH_TIMEOUT_S", "900"))

    def fire():
        last = None
        try:
            from hybrid_rendering_amd import tiling
            last = dict(tiling.LAST_COLLECTIVE)
        except Exception:
            pass
        msg = f"watchdog: rank {rank}/{world} not finished after {limit:.0f} s at stage '{PROGRESS['stage']}'; last collective posted: {last}"
        print("[bench] " + msg, file=sys.stderr, flush=True)
        if rank == 0:
            out = dict(PARTIAL) if PARTIAL else stub_line(args, world, msg)
            out["error"] = msg
            out.setdefault("comm", {})
            if isinstance(out["comm"], dict):
                out["comm"].update({"error": msg, "stage": PROGRESS["stage"], "last_collective": last})
            emit(out)
        os._exit(3)
    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


def binding_frac(entry):
    """the fraction of the roof that `bound` names (VERDICT r3 #5c: so that `frac` of a VALU-bound kernel is not read as "x % of HBM")"""
    b = entry.get("bound")
    if b == "valu":
        return entry.get("valu_frac")
    if b == "hbm":
        return entry.get("dram_frac")
    c = [v for v in (entry.get("valu_frac"), entry.get("dram_frac")) if v is not None]
    return max(c) if c else None


def pass_roofline(kernels):
    """aggregate of one pass's kernels (the unit north_star's "each pass at >= 40 % of the HBM roofline" is stated in): algorithmic bytes and time
    summed over its kernels; `binding` = the bound of the kernel the pass spends most of its time in, `valu_frac` time-weighted"""
    ks = [k for k in kernels.values() if k.get("ms")]
    ms = sum(k["ms"] for k in ks)
    if not ks or ms <= 0:
        return None
    b = sum(k.get("alg_bytes") or 0 for k in ks)
    top = max(ks, key=lambda k: k["ms"])
    vf = [(k["ms"], k["valu_frac"]) for k in ks if k.get("valu_frac") is not None]
    df = [(k["ms"], k["dram_frac"]) for k in ks if k.get("dram_frac") is not None]
    return {"alg_bytes": int(b), "ms": round(ms, 4), "frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "binding": top.get("bound"),
            "valu_frac": round(sum(m * v for m, v in vf) / sum(m for m, _ in vf), 3) if vf else None,
            "dram_frac": round(sum(m * v for m, v in df) / sum(m for m, _ in df), 3) if df else None,
            "kernels_missing_bytes": [n for n, k in kernels.items() if k.get("ms") and not k.get("alg_bytes")]}


def main():
    args = parse()
    isolate_stdout()
    if os.environ.get("HR_BENCH_TEST_C_STDOUT"):   # test hook (tests/test_bench_robustness.py): what a library's C stdio does — buffered, flushed when the process exits
        import ctypes
        ctypes.CDLL(None).printf(b"a library's banner, written through C stdio\n")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    PROGRESS.update(rank=rank, world=world)
    wd = start_watchdog(args, rank, world)
    import torch
    import torch.distributed as dist
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if os.environ.get("HR_FORCE_DEVICE") is not None:  # developer switch: several ranks on ONE GPU (functional test of the N>1 path)
        local_rank = int(os.environ["HR_FORCE_DEVICE"])
    comm_error = None
    try:
        torch.cuda.set_device(local_rank)
        if world > 1:
            stage("init_process_group")
            import datetime
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # a collective timeout LONGER than the watchdog's: the watchdog (which prints the line) fires first
            dist.init_process_group(os.environ.get("HR_DIST_BACKEND", "nccl"),   # "nccl" is RCCL on ROCm
                                    timeout=datetime.timedelta(seconds=float(os.environ.get("HR_BENCH_TIMEOUT_S", "900")) + 600))
            if os.environ.get("HR_BENCH_FAIL_AT") == "init":   # test hook (tests/test_bench_robustness.py)
                raise RuntimeError("HR_BENCH_FAIL_AT=init")
    except Exception as e:
        comm_error = f"{PROGRESS['stage']}: {e!r}"[:400]
    out = None
    if comm_error is None:
        try:
            out = run(args, torch, dist, rank, world, local_rank)
        except Exception as e:
            import traceback
            traceback.print_exc()
            if world == 1:
                out = dict(PARTIAL) if PARTIAL else stub_line(args, world, "")
                out["error"] = f"{PROGRESS['stage']}: {e!r}"[:400]
                emit(out)
                sys.exit(1)
            comm_error = f"{PROGRESS['stage']}: {e!r}"[:400]
    if comm_error is not None:
        # the distributed run failed: rank 0 reports the N = 1-equivalent local numbers next to the error, the other ranks leave quietly
        # (exit code 0: a launcher that tears the job down on the first non-zero exit would take rank 0 with it before it has printed)
        print(f"[bench] rank {rank}: distributed run failed ({comm_error})" + ("; falling back to a local single-GPU run" if rank == 0 else "; leaving"), file=sys.stderr, flush=True)
        if rank != 0:
            os._exit(0)
        seen = PROGRESS.get("ranks_seen", 1)
        try:
            stage("local_fallback")
            out = run(args, torch, None, 0, 1, local_rank)
        except Exception as e:
            out = dict(PARTIAL) if PARTIAL else stub_line(args, 1, "")
            out["error"] = f"local fallback after the distributed failure also failed: {e!r}"[:400]
        out["requested_gpus"] = world
        out.setdefault("error", f"distributed run failed ({comm_error}); every number on this line is rank 0's LOCAL single-GPU result (n_gpus = 1)")   # visible at top level
        out["comm"] = {"error": comm_error, "ranks_seen": seen, "backend": os.environ.get("HR_DIST_BACKEND", "nccl"),
                       "note": "the distributed run could not start / complete; `value` and everything else on this line are rank 0's LOCAL single-GPU numbers (n_gpus = 1)"}
    wd.cancel()
    flush_c_stdio()
    if world > 1 and comm_error is None:
        try:
            dist.barrier()   # every rank has emptied its C stdio buffer before rank 0 writes the line
        except Exception:
            pass
    if rank == 0:
        emit(out)
    if world > 1 and comm_error is None:
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    if comm_error is not None:
        os._exit(0)   # no orderly teardown of a process group that never worked


def run(args, torch, dist, rank, world, local_rank):
    """the bench proper -> the line (a dict); world == 1 never touches torch.distributed"""
    from hybrid_rendering_amd import api as hr
    from hybrid_rendering_amd import synth, tiling
    from hybrid_rendering_amd.frame import HybridFrame

    # N x the pixels of the N = 1 frame, same aspect and view (multiples of 8: tile / band alignment)
    sc = math.sqrt(world)
    W, H = (int(round(args.width * sc / 8)) * 8, int(round(args.height * sc / 8)) * 8) if world > 1 else (args.width, args.height)
    if args.obj:
        from hybrid_rendering_amd import assets
        sd = assets.load_obj(args.obj)
        scene_name = f"{os.path.basename(args.obj)}"
    else:
        sd = synth.sponza_like(args.detail, tier=args.tier)
        scene_name = "procedural Sponza-like" + (" (hard tier: layered fabric, foliage cards, grazing sun)" if args.tier == "hard" else "")
    stage("scene")
    ctx = hr.Context(local_rank)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_hard_light() if args.tier == "hard" else synth.sponza_light()
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    exact = 1 if args.exact else 0

    # ring of camera positions (dolly 0.5 units/frame)
    R = max(2, args.ring)
    cam_of = (lambda f: synth.camera_for_bounds(sd.bounds(), W / H, frame=f, dolly=0.5)) if args.obj else (lambda f: synth.sponza_camera(W / H, frame=f, dolly=0.5))
    cams = [cam_of(f) for f in range(R + 1)]
    # G-buffers: ring position i rendered with prev = i-1 (forward sweep) and with prev = i+1 (backward sweep)
    gbs = {}
    ubos = {}
    for i in range(R):
        for direction, j in (("f", i - 1 if i > 0 else 0), ("b", i + 1)):
            ubo = synth.make_ubo(cams[i], cams[j], light)
            ubos[(i, direction)] = ubo
            gbs[(i, direction)] = scene.gbuffer(ubo, W, H)
    torch.cuda.synchronize()

    # sequence of (cur, prev) keys: 0f,1f,...,R-1f, R-2b, ..., 0b, 1f, ...
    seq = [(i, "f") for i in range(R)] + [(i, "b") for i in range(R - 2, -1, -1)]
    seq = seq[1:] if len(seq) > 1 else seq

    def inputs_for(k):
        key = seq[k % len(seq)]
        pk = seq[(k - 1) % len(seq)]
        return hr.frame_inputs(gbs[key], gbs[pk], ubos[key], k, k & 1, sob_d, sr_d)

    cycle = [inputs_for(k) for k in range(len(seq) * 2)]  # even length: ping_pong parity preserved when cycling
    bounds = None
    if world > 1:
        stage("calibration")
        if os.environ.get("HR_BENCH_FAIL_AT") == "calibration":   # test hook
            raise RuntimeError("HR_BENCH_FAIL_AT=calibration")
        seen = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)               # first collective: how many ranks the backend really connects
        PROGRESS["ranks_seen"] = int(seen.item())
        # calibration (outside the timed region, identical on every rank): rays per tile row of the whole frame + geometry
        # pixels per tile row -> band boundaries of equal modelled cost (tiling.shadow_cost_per_tile_row)
        cal = hr.RayTracedShadows(ctx, W, H)
        cal.ray_trace(scene, cycle[0])
        cost = tiling.shadow_cost_per_tile_row(gbs[seq[0]]["depth"], cal.tile_ray_counts())
        cal.close()
        bounds = tiling.balanced_bounds(cost, world, H)
        tb = torch.tensor(bounds, dtype=torch.int64, device="cuda")
        dist.broadcast(tb, src=0)      # belt and braces: every rank uses rank 0's partition
        bounds = [int(v) for v in tb.cpu()]

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_run(tiled, steps, warmup, min_total_s=1.0):
        def step(k):
            fi = cycle[k % len(cycle)]
            fi.num_frames = k
            tiled.render(scene, fi)
        # frames in flight: the host enqueues a step in ~40 us, the GPU needs 200; left alone it runs a thousand launches ahead, and
        # past some depth the HIP queue makes the host wait in coarse steps (seen on the hybrid frame: sporadic +1 ms per frame in a
        # 30-frame run).  A renderer keeps a few frames in flight; so does this loop: before step k the host waits for step k - 32,
        # which the GPU finished long ago — the GPU never idles.
        fences = [torch.cuda.Event() for _ in range(4)]

        def paced(k):
            if k % 8:                       # a fence every 8th step (an event record costs the stream ~2 us): <= 32 steps in flight
                return step(k)
            f = fences[(k // 8) % len(fences)]
            if k >= 8 * len(fences):
                f.synchronize()
            step(k)
            f.record()
        for k in range(warmup):
            paced(k)

        def region(k_first):
            """EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks"""
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(k_first, k_first + steps):
                paced(k)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el
        # A region of K steps of a ~0.2 ms frame lasts a few ms: too short for a driver that samples GPU activity every few seconds, and one
        # host hiccup is a large share of it.  The region is therefore repeated until >= 1 s has been timed in total (round 5; round 4: 50 ms —
        # the driver's activity sampler still saw an idle GPU in all 3 samples of the 14 s run) — same K steps each, same brackets — and the
        # MEDIAN region is reported (`timed_repeats`, VERDICT r3 #5d); one region if K steps already take that long.
        els = [region(warmup)]
        n_rep = 1 if min_total_s <= 0 else int(min(500, max(1, math.ceil(min_total_s / max(els[0], 1e-6)))))
        if world > 1:   # every rank repeats the same number of times
            t = torch.tensor([n_rep], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n_rep = int(t.item())
        for r in range(1, n_rep):
            els.append(region(warmup + r * steps))
        el = float(np.median(els))
        timed_run.last = {"timed_repeats": len(els), "timed_region_ms": [round(e * 1e3, 3) for e in els][:16], "timed_total_ms": round(sum(els) * 1e3, 2)}
        return el, step

    # ------------------------------------------------------------------------------------------------ the timed region (headline)
    stage("headline")
    tiled = tiling.TiledShadows(ctx, W, H, rank, world, bounds=bounds)
    tiled.params.exact = exact
    shadows = tiled.pass_
    b0, b1 = tiled.b0, tiled.b1
    elapsed, step = timed_run(tiled, args.steps, args.warmup)
    timing = dict(timed_run.last)
    stage("per_kernel")

    # ---- per-kernel HIP events on the launch stream, right AFTER the timed region (same stream, same frame cycle: 12 event records per
    # ~0.2 ms frame inside it cost 18 % of the throughput; the per-kernel averages agree within 1.5 % either way, and rocprofv3 agrees
    # with both — DESIGN.md §5), plus ray counts
    acc = {}
    rays_total, n_prof = 0, min(args.steps, 60)
    k0 = args.warmup + args.steps
    shadows.set_profiling(True)
    for k in range(k0, k0 + n_prof):
        step(k)
        rays_total += shadows.ray_count()
        for name, ms, nbytes in shadows.stage_times():
            a = acc.setdefault(name, [0.0, nbytes])
            a[0] += ms / n_prof
    shadows.set_profiling(False)
    rays_per_frame = rays_total / n_prof
    if world > 1:
        # useful rays = rays of the band rows only: count them with a halo-free pass on the same inputs
        counter = hr.RayTracedShadows(ctx, W, H, hr.SCALE_FULL_RES, band=(b0, b1, 0, 0))
        tot = 0
        for k in range(k0, k0 + 8):
            fi = cycle[k % len(cycle)]
            fi.num_frames = k
            counter.ray_trace(scene, fi)
            tot += counter.ray_count()
        rays_per_frame = tot / 8
        counter.close()
    stages = {n: dict(ms=v[0], bytes=v[1]) for n, v in acc.items()}
    # instrumented trace (node visits / triangle tests) on a few frames of the cycle
    # (VERDICT r4 #1a) the counts `frac` divides are those of the TIMED kernel: occluder cache ON, in the state the previous frame of the
    # cycle left it in (hr_shadows_trace_stats_timed).  The full walk (cache bypassed: comparable between frames, builds and rounds) is
    # reported beside it.  Frame kk is counted, then rendered, so that frame kk + 1 meets the cache a timed frame meets.
    nn = nt = nr = wn = wt = wr = 0
    kk0 = k0 + n_prof
    for kk in range(kk0, kk0 + 4):
        fi = cycle[kk % len(cycle)]
        fi.num_frames = kk
        r, a, b = shadows.trace_stats(scene, fi)                    # the walk
        wr, wn, wt = wr + r, wn + a, wt + b
        r, a, b = shadows.trace_stats(scene, fi, timed=True)        # what the timed kernel does on this frame
        nr, nn, nt = nr + r, nn + a, nt + b
        step(kk)
    nodes_per_ray, tris_per_ray = nn / max(nr, 1), nt / max(nr, 1)
    walk_nodes_per_ray, walk_tris_per_ray = wn / max(wr, 1), wt / max(wr, 1)
    px = W * (b1 - b0)
    trace_bytes = px * 12.125 + rays_per_frame * (nodes_per_ray * NODE_BYTES + tris_per_ray * TRI_BYTES)
    walk_bytes = px * 12.125 + rays_per_frame * (walk_nodes_per_ray * NODE_BYTES + walk_tris_per_ray * TRI_BYTES)
    if "ray_trace" in stages:
        stages["ray_trace"]["bytes"] = int(trace_bytes)
    prof = load_profile()
    profiled_config = world == 1 and (W, H) == (1920, 1080) and not args.obj and args.tier == "standard"
    if not profiled_config:
        # another frame size / scene than the profiled one: neither the per-launch counter traffic nor the durations carry over
        prof = dict(prof, traffic={}, traffic_lo={}, avg_us={}, sq={})
    for n, s in stages.items():
        kern = kernel_of("shadows", n, exact) or n
        s.update(classify(prof, kern, s["ms"], s["bytes"], gather=(n == "ray_trace")))
        s["GBps"] = s["bytes"] / (s["ms"] * 1e-3) / 1e9 if s["ms"] > 0 else 0.0

    total_rays = rays_per_frame * args.steps * world
    if world > 1:
        t = torch.tensor([rays_per_frame], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_rays = float(t.item()) * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays / elapsed / 1e6
    comm_info = None
    if world > 1:
        # what the driver's SCALE record needs to tell compute from communication: how many ranks RCCL really saw, the band cut, and
        # the cost of one history exchange posted and waited for back to back (max over ranks; in the frame it hides under the next trace)
        seen = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ex = torch.tensor([tiled.time_exchange(20)], dtype=torch.float64, device="cuda")
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        comm_info = {"backend": dist.get_backend(), "ranks_seen": int(seen.item()), "band_bounds": bounds, "history_rows_exchanged": tiled.history_rows,
                     "exchange_us_per_frame": round(float(ex.item()), 1),
                     "exchange_bytes_per_boundary": int(tiled.history_rows * W * (4 + 8)),
                     "note": "one grouped ncclSend/ncclRecv pair per neighbour and frame (feedback image 4 B/px + moments 8 B/px of the 40 rows next to each band boundary), "
                             "posted after the a-trous chain and waited for before the NEXT frame's temporal kernel; exchange_us_per_frame is its stand-alone cost"}
    dom = max(stages.items(), key=lambda kv: kv[1]["ms"])
    out = {
        "metric": "shadow Mrays/s over the fully denoised frame (1 spp trace + SVGF temporal + 4x a-trous)",
        "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{W}x{H} {scene_name} ({sd.n_tris} tris) ray-traced shadows 1spp + SVGF denoise"
                               + (f", {world}x the pixels of 1920x1080 (same view), row-tiled into {world} cost-balanced bands" if world > 1 else ""),
                   "math": ("exact=1: every stage image bit-identical to the oracle" if exact else
                            "exact=0 (shipping mode): masks / ray counts bit-exact, fp16 images within tolerance (tests/test_gpu_tolerance.py)"),
                   "trace_launch_order": "blockIdx" if os.environ.get("HR_TILE_ORDER") == "0" else "heaviest tiles of the last frame first (csrc/tile_order.h)",
                   "rays_per_frame_per_gpu": int(rays_per_frame), "pixels_per_gpu": px, "bvh_nodes": int(scene.info.n_nodes),
                   "nodes_per_ray": round(nodes_per_ray, 2), "tris_per_ray": round(tris_per_ray, 2),
                   "nodes_per_ray_full_walk": round(walk_nodes_per_ray, 2), "tris_per_ray_full_walk": round(walk_tris_per_ray, 2)},
        "timed_repeats": timing["timed_repeats"], "timed_region_ms": timing["timed_region_ms"], "timed_total_ms": timing["timed_total_ms"],
        "timing_note": "ms_per_step = MEDIAN over `timed_repeats` timed regions of exactly `steps` steps each (barrier + synchronize on both sides of every region; "
                       "repeated until >= 1 s is timed in total)",
        "denoised_frames_per_s": round(args.steps / elapsed, 2),                  # frames of W x H (the whole tiled frame)
        "denoised_1080p_equiv_per_s": round(W * H / (args.width * args.height) * args.steps / elapsed, 2),
        "trace_only_Mrays_per_s": round(rays_per_frame / (stages["ray_trace"]["ms"] * 1e-3) / 1e6, 2) if "ray_trace" in stages else None,
        "roofline": {"kernel": dom[0], "kernel_name": dom[1].get("kernel"), "bound": dom[1].get("bound") or "latency", "achieved": round(dom[1]["GBps"], 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": dom[1]["frac"], "traffic": dom[1].get("traffic"), "traffic_lo": dom[1].get("traffic_lo"), "dram_frac": dom[1].get("dram_frac"),
                     "valu_issue_frac": dom[1].get("valu_issue_frac"), "lane_utilisation": dom[1].get("lane_utilisation"), "valu_frac": dom[1].get("valu_frac"),
                     "binding_frac": binding_frac(dom[1]),
                     # a trace kernel's algorithmic bytes are REQUESTED bytes: every lane's 80 B node / 48 B triangle fetch, most of them L1 / L2 hits
                     # (the tree is ~17 MB) — `frac` is SURVEY 8d's figure, not an HBM share; dram_frac is (VERDICT r4 #1b)
                     "frac_is_requested_bytes": dom[0] == "ray_trace",
                     "frac_full_walk": (round(walk_bytes / (dom[1]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dom[0] == "ray_trace" and dom[1]["ms"] > 0 else None),
                     "binding_note": "fraction of the roof NAMED IN `bound`: valu -> valu_frac (issue share x lane utilisation), hbm -> dram_frac "
                                     "(counter traffic / time / 8 TB/s), latency -> the larger of the two; `frac` (contract) stays algorithmic bytes / time / 8 TB/s",
                     "profile_avg_us": dom[1].get("profile_avg_us"), "profile_state": dom[1].get("profile_state"), "live_event_us": round(dom[1]["ms"] * 1e3, 2),
                     "counters": prof["dir"] if (prof["dir"] and dom[1].get("traffic")) else None,
                     "traffic_source": ((prof["dir"] + "/pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE per launch; traffic_lo = FETCH_SIZE + WRITE_SIZE: profiles/r3_calib)"
                                         + ("" if dom[1].get("profile_state") != "stale" else " — STALE: the live kernel time is > 10 % off the profiled one")))
                                       if (prof["dir"] and dom[1].get("traffic")) else None,
                     "bound_source": (prof["dir"] + "/sq_counters.json (raw SQ_INSTS_VALU, GRBM_GUI_ACTIVE, SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU; 1080p bench frame)") if prof["dir"] else None,
                     "algorithmic_bytes": int(dom[1]["bytes"]),
                     "note": "achieved/frac divide the ALGORITHMIC bytes (SURVEY 8d: G-buffer + mask + visited BVH nodes x 80 B + tested triangles x 48 B) by the kernel time; the node / "
                             "triangle counts are those of the TIMED kernel (occluder cache on: a ray first tests the triangle that occluded its pixel last frame), frac_full_walk uses the "
                             "counts of the cache-less walk; the ~17 MB BVH is served by L2 / Infinity Cache, so dram_frac (counter traffic) is the HBM-side figure; the operative roof of "
                             "this kernel is VALU issue: valu_frac = valu_issue_frac x lane_utilisation (= binding_frac)"},
        "stages": {n: {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()} for n, s in stages.items()},
    }
    if comm_info:
        out["comm"] = comm_info

    # ---- the other arithmetic mode, same frames, shorter run (reported, never `value`)
    other = tiling.TiledShadows(ctx, W, H, rank, world, bounds=bounds)
    other.params.exact = 1 - exact
    stage("other_mode")
    PARTIAL.update(out)
    el2, _ = timed_run(other, max(20, args.steps // 4), 10, min_total_s=0.0)
    n2 = max(20, args.steps // 4)
    out["tolerance_mode" if exact else "exact_mode"] = {"ms_per_step": round(el2 / n2 * 1e3, 4), "value": round(total_rays / args.steps * n2 / el2 / 1e6, 2), "unit": "Mrays/s",
                                                        "steps": n2, "note": "hr_shadows_params.exact = %d on the same frames" % (1 - exact)}
    other.wait_exchange()
    torch.cuda.synchronize()
    other.pass_.close()

    # ---- the other BASELINE configurations (outside the timed region) ------------------------------------------------------------
    PARTIAL.update(out)
    if not args.no_passes and not args.obj and args.tier == "standard":
        stage("passes" if world == 1 else "hybrid_4k")
        try:
            if world == 1:
                out["passes"] = passes_block(torch, hr, HybridFrame, ctx, scene, sd, prof, exact)
                out["passes"]["hard_tier"] = hard_tier_block(torch, hr, synth, ctx, sob_d, sr_d, exact)
            else:
                hf = HybridFrame(ctx, scene, sd, 3840, 2160, rank, world, exact=exact)
                first_cut = list(hf.bounds)
                hf.rebalance(rounds=2, barrier=barrier)     # re-cut the bands from the measured per-rank frame times (outside the timed region)
                ms = hf.time(12, 4, barrier=barrier, repeats=2)
                rays = sum(hf.ray_counts().values())
                t = torch.tensor([ms, float(rays)], dtype=torch.float64, device="cuda")
                tm = t.clone()
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                cm = torch.tensor([hf.shadows.time_exchange(10), hf.ao.time_exchange(10), hf.refl.time_exchange(10), hf.gi.time_allgather(10)], dtype=torch.float64, device="cuda")
                dist.all_reduce(cm, op=dist.ReduceOp.MAX)
                out["hybrid_4k"] = {"workload": "ONE 3840x2160 hybrid frame (shadows + AO 4 spp + DDGI 16x8x16x256 + half-res reflections) row-tiled over the GPUs: strong scaling (BASELINE configs[4])",
                                    "n_gpus": world, "ms_per_frame": round(float(tm[0]), 4), "frames_per_s": round(1e3 / float(tm[0]), 1),
                                    "Mrays_per_s": round(float(t[1]) / float(tm[0]) / 1e3, 1), "bands": hf.bounds, "bands_before_rebalancing": first_cut, "scaling": "strong",
                                    "comm_us_per_frame": {"shadows_exchange": round(float(cm[0]), 1), "ao_exchange": round(float(cm[1]), 1), "reflections_exchange": round(float(cm[2]), 1),
                                                          "ddgi_atlas_allgather": round(float(cm[3]), 1),
                                                          "note": "stand-alone cost of each collective (posted and waited for back to back, max over ranks); the three exchanges overlap the next "
                                                                  "frame's traces, the all-gather sits on the DDGI -> reflections chain"}}
                out["hybrid_4k"]["forked_streams"] = bool(hf.forked)
                if hf.forked_error:
                    out["hybrid_4k"]["forked_error"] = hf.forked_error
                hf.close()
        except Exception as e:   # a report next to the headline, never a reason to lose the bench line
            out["passes_error"] = repr(e)[:300]

    # ---- CPU baseline on the host cores, bounded sample, rank 0 at N=1 only -----------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        stage("cpu_baseline")
        try:
            out["cpu_baseline"] = cpu_baseline(torch, np, scene, sd, gbs, ubos, seq, sob, sr, W, H, light, synth, args)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)[:300]}
        PARTIAL.update(out)
    stage("done")
    return out


def passes_block(torch, hr, HybridFrame, ctx, scene, sd, prof, exact):
    """BASELINE configs[2], [3] and [4] on ONE GPU: wall-clock per pass and per hybrid frame, per-kernel event times and roofline fractions"""
    res = {}
    hf = HybridFrame(ctx, scene, sd, 1920, 1080, exact=exact)
    hf.time(6, 2)                                                   # warm every pass (history, atlases)
    wall = {n: hf.time(12, 4, only=n, repeats=3) for n in ("shadows", "ao", "ddgi", "reflections")}
    hybrid = hf.time(12, 4, repeats=3)
    hf.concurrent_streams(True, "streams")
    hybrid3 = hf.time(12, 6, repeats=3)
    hf.concurrent_streams(True, "graph")
    hybridg = hf.time(12, 6, repeats=3)
    hf.concurrent_streams(False)
    rays = hf.ray_counts()
    st = hf.stage_times(10)
    tb = hf.trace_bytes()
    label = {"ao": "AO 4 spp + temporal + 2 blurs, 1920x1080 (configs[2])", "reflections": "reflections 1 spp at half resolution + SVGF + upsample, 1920x1080 frame (configs[3])",
             "ddgi": "DDGI 16x8x16 probes x 256 rays: trace + probe updates + per-pixel sample, 1920x1080", "shadows": "shadows 1 spp + SVGF, 1920x1080 (inside the hybrid frame)"}
    for n in ("ao", "reflections", "ddgi", "shadows"):
        res[n] = {"workload": label[n], "ms_per_frame": round(wall[n], 4), "frames_per_s": round(1e3 / wall[n], 1), "rays_per_frame": rays[n],
                  "Mrays_per_s": round(rays[n] / wall[n] / 1e3, 1), "kernels": kernel_entries(prof, n, st[n], exact, tb.get(n))}
        res[n]["roofline"] = pass_roofline(res[n]["kernels"])
    res["hybrid_1080p"] = {"workload": "whole hybrid frame 1920x1080 on one GPU: shadows + AO 4 spp + DDGI 16x8x16x256 + half-res reflections, one stream",
                           "ms_per_frame": round(hybrid, 4), "frames_per_s": round(1e3 / hybrid, 1), "Mrays_per_s": round(sum(rays.values()) / hybrid / 1e3, 1),
                           "concurrent_streams": {"ms_per_frame": round(hybrid3, 4), "frames_per_s": round(1e3 / hybrid3, 1),
                                                  "note": "hr_hybrid_frame HR_FRAME_STREAMS (C ABI; hr::HybridFrame in include/hr/passes.hpp): shadows | AO | DDGI trace + update -> reflections | "
                                                          "DDGI sample forked over four HIP streams, joined at the end of the frame (independent chains; bit-identical images)"},
                           "hip_graph": {"ms_per_frame": round(hybridg, 4), "frames_per_s": round(1e3 / hybridg, 1),
                                         "note": "HR_FRAME_GRAPH: the same forked frame captured into one hipGraph per frame; the instantiated graph is updated in place (hipGraphExecUpdate) "
                                                 "because UBO, frame counter and ping-pong parity travel in the kernel arguments"}}
    hf.close()
    # configs[3] read as FULL-resolution reflections (the reference's default, timed above, is half resolution)
    hfr = HybridFrame(ctx, scene, sd, 1920, 1080, exact=exact, refl_scale=0)
    hfr.time(4, 2)
    ms_r = hfr.time(12, 4, only="reflections", repeats=3)
    rays_r = hfr.ray_counts()["reflections"]
    res["reflections_full_res"] = {"workload": "reflections 1 spp at FULL resolution + SVGF, 1920x1080 (configs[3] without the reference's half-resolution default)",
                                   "ms_per_frame": round(ms_r, 4), "frames_per_s": round(1e3 / ms_r, 1), "rays_per_frame": rays_r, "Mrays_per_s": round(rays_r / ms_r / 1e3, 1)}
    hfr.close()
    hf4 = HybridFrame(ctx, scene, sd, 3840, 2160, exact=exact)
    ms4 = hf4.time(8, 4, repeats=3)
    r4 = hf4.ray_counts()
    hf4.concurrent_streams(True, "streams")
    ms43 = hf4.time(8, 4, repeats=3)
    hf4.concurrent_streams(True, "graph")
    ms4g = hf4.time(8, 4, repeats=3)
    hf4.concurrent_streams(False)
    st4 = hf4.stage_times(8)
    tb4 = hf4.trace_bytes()
    prof4 = load_profile("_4k")
    res["hybrid_4k_one_gpu"] = {"workload": "whole hybrid frame 3840x2160 on ONE GPU (BASELINE configs[4] names 8)", "ms_per_frame": round(ms4, 4), "frames_per_s": round(1e3 / ms4, 1),
                                "Mrays_per_s": round(sum(r4.values()) / ms4 / 1e3, 1), "concurrent_streams": {"ms_per_frame": round(ms43, 4), "frames_per_s": round(1e3 / ms43, 1)},
                                "hip_graph": {"ms_per_frame": round(ms4g, 4), "frames_per_s": round(1e3 / ms4g, 1)},
                                "kernels": {n: kernel_entries(prof4, n, st4[n], exact, tb4.get(n)) for n in ("shadows", "ao", "ddgi", "reflections")},
                                "kernels_note": "per-kernel HIP-event times of the 4K frame; counters from " + (prof4["dir"] or "(no 4K profile committed)") +
                                                " (*_4k files: the working set of a 4K pass exceeds the 256 MiB Infinity Cache, so dram_frac is an HBM figure here)"}
    res["hybrid_4k_one_gpu"]["roofline"] = {n: pass_roofline(k) for n, k in res["hybrid_4k_one_gpu"]["kernels"].items()}
    res["hybrid_4k_one_gpu"]["roofline_note"] = ("per PASS at 3840x2160: algorithmic bytes (SURVEY 8d; trace kernels: pixels x inputs + instrumented node steps x 80 B + triangle tests x 48 B) "
                                                 "summed over the pass's kernels / summed HIP-event time / 8 TB/s; `binding` = what limits the kernel the pass spends most time in")
    hf4.close()
    res["note"] = ("kernel `frac` = algorithmic bytes (SURVEY 8d) / HIP-event time / 8 TB/s; `traffic` / `dram_frac`, `valu_issue_frac`, `lane_utilisation`, `valu_frac`, `bound` from the "
                   "rocprofv3 counters committed under " + (prof["dir"] or "profiles/ (none for this configuration)") + " by EXACT kernel instance; `profile_state` = stale when the live "
                   "event time is > 10 % off the profiled duration; the ray-trace kernels' algorithmic bytes come from their instrumented builds (hr_*_trace_stats: "
                   "node steps x 80 B + triangle tests x 48 B + the per-pixel inputs / outputs); `binding_frac` = the fraction of the roof named in `bound`")
    return res


def kernel_entries(prof, pass_, stage_times, exact, trace=None):
    """{stage: {ms, kernel, alg_bytes, frac, traffic, dram_frac, valu_issue_frac, lane_utilisation, valu_frac, bound, binding_frac, profile_state}} for one
    pass.  trace: HybridFrame.trace_bytes()[pass] — the ray-trace kernel's algorithmic bytes from its instrumented build (nodes x 80 B + triangles x 48 B)"""
    out = {}
    for s, (ms, b) in stage_times.items():
        k = kernel_of(pass_, s, exact)
        e = {"ms": round(ms, 4)}
        if s == "ray_trace":
            b = trace["bytes"] if trace else 0
            if trace:
                e.update(rays=trace["rays"], nodes_per_ray=trace["nodes_per_ray"], tris_per_ray=trace["tris_per_ray"])
        if b:
            e["alg_bytes"] = int(b)
        if k:
            c = classify(prof, k, ms, b, gather=(s == "ray_trace"))
            e.update({kk: vv for kk, vv in c.items() if vv is not None})
            bf = binding_frac(c)
            if bf is not None:
                e["binding_frac"] = bf
            if (e.get("frac") or 0) > 1.0:
                e["frac_note"] = ("> 1: SURVEY 8d counts every lane's BVH node / triangle fetch (80 / 48 B each); they are L1 / L2 hits (the BVH is ~17 MB), "
                                  "not HBM traffic — the kernel is VALU-bound: read binding_frac / dram_frac")
        elif b and ms > 0:
            e["frac"] = round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out[s] = e
    return out


def hard_tier_block(torch, hr, synth, ctx, sob_d, sr_d, exact):
    """the headline pass on the harder workload tier: ~2.5 M triangles, layered fabric + foliage cards, sun 20 degrees
    above the horizon — what a deep, thin-geometry BVH costs per ray"""
    W, H = 1920, 1080
    sd = synth.sponza_like(1.0, tier="hard")
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_hard_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
    gbs = [scene.gbuffer(u, W, H) for u in ubos]
    fis = [hr.frame_inputs(gbs[k & 1], gbs[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d) for k in range(2)]
    p = hr.RayTracedShadows(ctx, W, H)
    p.params.exact = exact
    for k in range(6):
        fis[k & 1].num_frames = k
        p.render(scene, fis[k & 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    for k in range(6, 6 + n):
        fis[k & 1].num_frames = k
        p.render(scene, fis[k & 1])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    p.set_profiling(True)
    acc = {}
    for k in range(70, 80):
        fis[k & 1].num_frames = k
        p.render(scene, fis[k & 1])
        for nme, t, b in p.stage_times():
            acc[nme] = acc.get(nme, 0.0) + t / 10
    rays = p.ray_count()
    r, nn, nt = p.trace_stats(scene, fis[0])
    out = {"workload": f"1920x1080 hard tier ({sd.n_tris} tris, {int(scene.info.n_nodes)} BVH nodes, depth {int(scene.info.max_depth)}): shadows 1 spp + SVGF",
           "ms_per_frame": round(ms, 4), "rays_per_frame": int(rays), "Mrays_per_s": round(rays / ms / 1e3, 1), "nodes_per_ray": round(nn / max(r, 1), 2),
           "tris_per_ray": round(nt / max(r, 1), 2), "trace_only_Mrays_per_s": round(rays / acc.get("ray_trace", 1e9) / 1e3, 1),
           "stage_ms": {k: round(v, 4) for k, v in acc.items()}}
    p.close(); scene.close()
    return out


def cpu_baseline(torch, np, scene, sd, gbs, ubos, seq, sob, sr, W, H, light, synth, args):
    """BASELINE.md §3 / SURVEY §8d: the identical BVH (binary form) and the identical ray batch of a bench frame replayed through the
    oracle's scalar any-hit traversal (OpenMP over rays, all host threads), and the oracle's denoise chain timed on its own."""
    from oracle import pyoracle as po
    ncpu = po.set_threads()          # min(affinity, cgroup quota): the box shows 256 hardware threads and grants 16
    host = {k: {n: (t.cpu().numpy().view(np.uint16) if t.dtype == torch.float16 else t.cpu().numpy()) for n, t in gbs[k].items()} for k in seq[:3]}
    osc = po.Scene(sd)
    res = {"unit": "Mrays/s", "cores": ncpu, "kind": "port"}
    # (1) trace replay: the rays the GPU kernel fires for this frame, generated once, traversed repeatedly for ~10 s
    key = seq[1]
    rays = po.shadows_gen_rays(ubos[key], host[key]["depth"], host[key]["gb2"], sob, sr, 0.5, 1)
    rays = np.ascontiguousarray(rays[rays[:, 7] > 0.0])
    rays[:, 7] = 0.01                                                 # t_min of the shadow query (ray_query.glsl:13-27)
    _, st = osc.any_hit(rays[::16], stats=True)                       # instrumented (serial) pass on a sample: nodes / triangles per ray
    st = [float(v) * 16 for v in st]
    osc.any_hit(rays)                                                 # warm
    n_rep, t0 = 0, time.perf_counter()
    while True:
        occ = osc.any_hit(rays)
        n_rep += 1
        if time.perf_counter() - t0 > float(os.environ.get("HR_BENCH_CPU_SECONDS", "8")) or n_rep >= 400:
            break
    dt = time.perf_counter() - t0
    res["value"] = round(len(rays) * n_rep / dt / 1e6, 3)
    res["trace_replay"] = {"value": res["value"], "unit": "Mrays/s", "rays_per_batch": int(len(rays)), "batches": n_rep, "seconds": round(dt, 2),
                           "nodes_per_ray_bvh2": round(float(st[0]) / len(rays), 2), "tris_per_ray_bvh2": round(float(st[1]) / len(rays), 2),
                           "occluded_fraction": round(float(occ.mean()), 4)}
    # (1b) the same batch through the PRODUCT's tree on the host (VERDICT r4 #8 / north_star "CPU replay of the same BVH + ray batches"):
    # oracle/orc_replay8.cpp = the product's builder + the walk of csrc/traverse.h restated on the host + the oracle's triangle test
    try:
        rp = po.Replay8(sd)
        occ8, st8 = rp.any_hit(rays, stats=True)
        budget = float(os.environ.get("HR_BENCH_CPU_SECONDS", "8"))
        n8, t0 = 0, time.perf_counter()
        while True:
            occ8 = rp.any_hit(rays)
            n8 += 1
            if time.perf_counter() - t0 > budget or n8 >= 400:
                break
        dt8 = time.perf_counter() - t0
        res["trace_replay_same_tree"] = {"value": round(len(rays) * n8 / dt8 / 1e6, 3), "unit": "Mrays/s", "cores": ncpu, "batches": n8, "seconds": round(dt8, 2),
                                         "bvh_nodes": rp.num_nodes(), "nodes_per_ray": round(float(st8[0]) / len(rays), 2), "tris_per_ray": round(float(st8[1]) / len(rays), 2),
                                         "masks_equal": bool(np.array_equal(occ8, occ)),
                                         "note": "the product's 8-wide tree (csrc/bvh_build.cpp) walked on the host as csrc/traverse.h walks it on the GPU (oracle/orc_replay8.cpp)"}
    except Exception as e:
        res["trace_replay_same_tree"] = {"error": repr(e)[:200]}
    res["sample"] = (f"the {len(rays)} shadow rays of one {W}x{H} bench frame (identical origins / directions / t_max) through the oracle's scalar BVH2 any-hit "
                     f"traversal, OpenMP over rays on {ncpu} host threads (the box's cgroup CPU quota; {os.cpu_count()} hardware threads are visible), {n_rep} repetitions")
    # (2) the denoise chain of the oracle (temporal + 4 a-trous), one warm frame, for context
    op = po.ShadowsPass(W, H)
    op.render(osc, ubos[seq[0]], host[seq[0]], host[seq[0]], sob, sr, 0)
    mask, _ = po.shadows_ray_trace(osc, ubos[key], host[key]["depth"], host[key]["gb2"], sob, sr, 0.5, 1)
    t0 = time.perf_counter()
    tv, mom, tiles = po.shadows_temporal(ubos[key], mask, host[key], host[seq[0]], op.prev_image, op.moments)
    t1 = time.perf_counter()
    img = tv
    for i in range(4):
        img = po.shadows_atrous(img, host[key]["gb2"], host[key]["gb3"], tiles, 1 << i, power=1.2 if i == 3 else 0.0)
    t2 = time.perf_counter()
    res["denoise_ms"] = {"temporal": round((t1 - t0) * 1e3, 2), "atrous_x4": round((t2 - t1) * 1e3, 2), "threads": ncpu,
                         "note": "oracle restatement of shadows_denoise_reprojection.comp + 4 x shadows_denoise_atrous.comp on one 1920x1080 frame (OpenMP over rows)"}
    # (3) whole oracle frames (trace + denoise through the ctypes boundary), the figure round 1 reported
    nrays, nf = 0, max(1, args.cpu_frames)
    t0 = time.perf_counter()
    for f in range(nf):
        k1, k0 = seq[(f + 1) % 3], seq[f % 3]
        op.render(osc, ubos[k1], host[k1], host[k0], sob, sr, f + 1)
        nrays += op.stages["rays"]
    dt = time.perf_counter() - t0
    res["whole_frames"] = {"frames_per_s": round(nf / dt, 3), "Mrays_per_s": round(nrays / dt / 1e6, 3), "frames": nf}
    # (4) the reference's OWN shaders (oracle/_ref, one host thread): a small frame of the same view, trace + denoise
    try:
        from oracle import pyref, ref_harness as rh
        if pyref.available():
            rw, rhh = 240, 136
            rcams = [synth.sponza_camera(rw / rhh, frame=f, dolly=0.5) for f in range(3)]
            rubos = [synth.make_ubo(rcams[i + 1], rcams[i], light) for i in range(2)]
            rgb = [{n: (t.cpu().numpy().view(np.uint16) if t.dtype == torch.float16 else t.cpu().numpy()) for n, t in scene.gbuffer(u, rw, rhh).items()} for u in rubos]
            rp, orp = rh.RefShadowsPass(rw, rhh), po.ShadowsPass(rw, rhh)
            rp.render(osc, rubos[0], rgb[0], rgb[0], sob, sr, 0)
            orp.render(osc, rubos[0], rgb[0], rgb[0], sob, sr, 0)
            t0r = time.perf_counter()
            rp.render(osc, rubos[1], rgb[1], rgb[0], sob, sr, 1)
            dtr = time.perf_counter() - t0r
            orp.render(osc, rubos[1], rgb[1], rgb[0], sob, sr, 1)
            res["reference_shaders"] = {"value": round(orp.stages["rays"] / dtr / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                                        "sample": f"one {rw}x{rhh} frame of the same view through the reference's shaders compiled for the CPU (oracle/_ref)",
                                        "bit_identical_to_port": bool(np.array_equal(rp.stages["output"], orp.stages["output"]))}
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        res["reference_shaders"] = {"error": repr(e)[:200]}
    return res


if __name__ == "__main__":
    main()
