"""bench_report.py — the REPORTING half of bench.py (round 6 split): which kernel a stage is, what the committed rocprofv3 counters say about it,
the roofline figures derived from them, and the ONE compact JSON line.  Pure functions of numbers measured elsewhere: no GPU, no timing, nothing
from oracle/ — bench.py (the timed harness) imports everything here; tests/test_bench_robustness.py exercises compact_line() on the CPU."""
from __future__ import annotations

import json
import os
import re

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
L2_PEAK_GBS = 34000.0  # MI355X_MICROARCH.md: ~34 TB/s aggregate L2 (8 XCDs x 4 MiB) — the roof REQUESTED bytes of a BVH walk can be priced against
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
        return {"ray_trace": "k_ddgi_trace<false>", "probe_update": "k_ddgi_probe_update<true>",
                "sample_probe_grid": "k_ddgi_sample" if exact else "kf_ddgi_sample"}.get(stage)
    if pass_ == "reflections":
        if stage == "ray_trace": return "k_refl_trace<false>"   # one instance: the parity gather in both modes (round 6)
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


LINE_LIMIT = 7000        # bytes: the driver keeps an ~8 KB tail of stdout; round 4's 23 KB line was not parsed (BENCH_r04.parsed = null)
DETAIL_FILE = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


def _short(s_, n=160):
    return s_ if not isinstance(s_, str) or len(s_) <= n else s_[:n - 1] + "…"


def comm_block(backend, ranks_seen, band_bounds, history_rows, exchange_us, width):
    """the `comm` object of an N > 1 line: what the driver's SCALE record needs to tell compute from communication — how many ranks the backend
    really connected, the band cut, the stand-alone cost of one history exchange (max over ranks; in the frame it hides under the next trace)"""
    return {"backend": backend, "ranks_seen": int(ranks_seen), "band_bounds": [int(b) for b in band_bounds], "history_rows_exchanged": int(history_rows),
            "exchange_us_per_frame": round(float(exchange_us), 1), "exchange_bytes_per_boundary": int(history_rows * width * (4 + 8)),
            "note": "one grouped ncclSend/ncclRecv pair per neighbour and frame (feedback image 4 B/px + moments 8 B/px of the 40 rows next to each band boundary), "
                    "posted after the a-trous chain and waited for before the NEXT frame's temporal kernel; exchange_us_per_frame is its stand-alone cost"}


_PASS_KEYS = ("ms", "frac", "frac_kind", "l2_frac", "dram_frac", "valu_frac", "bound")


def _pass_summary(entry, roof):
    """{ms, frac, dram_frac, valu_frac, bound} of one pass from its pass_roofline() aggregate (+ wall-clock ms / Mrays/s when known)"""
    out = {}
    if isinstance(entry, dict):
        if entry.get("ms_per_frame") is not None:
            out["wall_ms"] = entry["ms_per_frame"]
        if entry.get("Mrays_per_s") is not None:
            out["Mrays_per_s"] = entry["Mrays_per_s"]
    if isinstance(roof, dict):
        out.update(ms=roof.get("ms"), frac=roof.get("frac"), frac_kind={"requested_bytes": "req", "hbm_algorithmic": "hbm"}.get(roof.get("frac_kind")), l2_frac=roof.get("l2_frac"),
                   dram_frac=roof.get("dram_frac"), valu_frac=roof.get("valu_frac"), bound=roof.get("binding"))
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
                                  "valu_frac", "binding_frac", "frac_kind", "l2_frac", "frac_is_requested_bytes", "frac_full_walk", "algorithmic_bytes", "live_event_us", "profile_avg_us",
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
        if isinstance(ps.get("instanced"), dict):
            summ["instanced"] = _pick(ps["instanced"], ("instances", "update_ms_per_frame", "top_level_rebuilds", "ms_per_frame_updated_tree", "ms_per_frame_rebuilt_tree", "masks_equal", "error"))
        if isinstance(ps.get("hard_tier"), dict):
            summ["hard_tier"] = _pick(ps["hard_tier"], ("ms_per_frame", "Mrays_per_s", "trace_only_Mrays_per_s", "nodes_per_ray", "tris_per_ray"))
        summ["keys"] = ("per pass: ms = sum of its kernels' HIP-event times; frac = SURVEY 8d bytes / ms / 8 TB/s with frac_kind hbm = compulsory HBM bytes, req = REQUESTED bytes of a "
                        "BVH walk (served by L1 / L2: may pass 1; l2_frac = the same over 34 TB/s of L2); dram_frac = counter traffic; valu_frac = issue x lanes; bound = of its longest kernel")
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
    req = sum(k.get("alg_bytes") or 0 for k in ks if k.get("frac_kind") == "requested_bytes")
    kind = "requested_bytes" if b and req > 0.5 * b else "hbm_algorithmic"   # a pass dominated by its BVH walk is priced in requested bytes: `frac` may pass 1
    return {"alg_bytes": int(b), "ms": round(ms, 4), "frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frac_kind": kind,
            "l2_frac": round(b / (ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4) if kind == "requested_bytes" else None, "binding": top.get("bound"),
            "valu_frac": round(sum(m * v for m, v in vf) / sum(m for m, _ in vf), 3) if vf else None,
            "dram_frac": round(sum(m * v for m, v in df) / sum(m for m, _ in df), 3) if df else None,
            "kernels_missing_bytes": [n for n, k in kernels.items() if k.get("ms") and not k.get("alg_bytes")]}


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
            # what `frac` divides: a stencil's compulsory HBM bytes (each input / output once), or — for a kernel that walks the BVH — SURVEY 8d's
            # REQUESTED bytes (nodes x 80 B + triangles x 48 B per lane), which L1 / L2 / Infinity Cache serve: those can exceed the HBM roof, so
            # they also get priced against the aggregate L2 bandwidth
            e["frac_kind"] = "requested_bytes" if s == "ray_trace" else "hbm_algorithmic"
            if s == "ray_trace" and ms > 0:
                e["l2_frac"] = round(b / (ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4)
        if k:
            c = classify(prof, k, ms, b, gather=(s == "ray_trace"))
            e.update({kk: vv for kk, vv in c.items() if vv is not None})
            bf = binding_frac(c)
            if bf is not None:
                e["binding_frac"] = bf
        elif b and ms > 0:
            e["frac"] = round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out[s] = e
    return out
