"""In-tree build of the HIP library (gfx950).  `python -m hybrid_rendering_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhybrid_rendering_amd.so")
SOURCES = ["api.hip", "shadows.hip", "bvh_build.cpp"]
OPTIONAL = ["denoise_fast.hip", "ao.hip", "reflections.hip", "ddgi.hip", "deferred.hip", "ground_truth.hip", "taa.hip", "frame.hip", "instances.hip"]
# -ffp-contract=off: every fp32 op is individually rounded; FMAs are explicit.
# -fno-slp-vectorize: the SLP vectoriser pairs independent fp32 operations into v_pk_mul / v_pk_add / v_pk_fma_f32.  On gfx950 a packed
# fp32 instruction issues at half rate, and pairing costs v_mov / v_pk_mov shuffles and hazard s_nops on top (kf_ddgi_sample: 305 packed
# ops + 82 moves + 84 nops of 1501 instructions).  Measured at 1080p: DDGI probe-grid sample 84.7 -> 73.1 us, depth probe update
# 92.4 -> 83.4, AO trace 400 -> 390, reflections trace 221 -> 214; nothing slower; same bits (same operations, scalar encodings).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, s) for s in OPTIONAL if os.path.exists(os.path.join(CSRC, s))]


COMM_LIB = os.path.join(HERE, "libhr_comm.so")   # include/hr_comm.h: RCCL / loopback transport of the row-tiled frame (host code only)


def build_comm(force: bool = False, verbose: bool = False) -> str:
    src = os.path.join(CSRC, "comm.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "hr_comm.h"), os.path.join(HERE, "..", "include", "hr_api.h"), LIB]
    if not force and os.path.exists(COMM_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(COMM_LIB) for d in deps):
        return COMM_LIB
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-x", "hip", "--offload-arch=gfx950", src, "-o", COMM_LIB + ".tmp",
           "-L", HERE, "-lhybrid_rendering_amd", "-ldl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(COMM_LIB + ".tmp", COMM_LIB)
    return COMM_LIB


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "hr_api.h"), os.path.join(HERE, "..", "include", "hr_api_stages.h"),
                                                           os.path.join(HERE, "..", "include", "hr_api_post.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, variant: str = None) -> str:
    """variant: a developer A/B build (HR_CFLAGS="-DFT_SHADOWS_EU=5" python -m hybrid_rendering_amd.build --variant eu5) written next to
    the product library as variants/libhybrid_rendering_amd.<variant>.so; HR_LIBRARY=<that path> selects it (api.py)."""
    out = LIB
    if variant:
        os.makedirs(os.path.join(HERE, "variants"), exist_ok=True)
        out = os.path.join(HERE, "variants", f"libhybrid_rendering_amd.{variant}.so")
    elif not force and not needs_build():
        build_comm(False, verbose)
        return LIB
    extra = os.environ.get("HR_CFLAGS", "").split()
    # one object per translation unit, compiled in parallel and cached by (source, headers, flags): a one-file change rebuilds in seconds
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(HERE, "_obj", variant or "product")
    os.makedirs(obj_dir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + extra
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(HERE, "..", "include", h) for h in ("hr_api.h", "hr_api_stages.h", "hr_api_post.h")] + [__file__]
    hdr_t = max(os.path.getmtime(h) for h in headers)
    tag = hashlib.sha1(" ".join(cflags).encode()).hexdigest()[:10]

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + "." + tag + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            return obj
        cmd = [hipcc()] + cflags + ["-x", "hip", "-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out + ".tmp", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    if not variant:
        build_comm(True, verbose)
    return out


if __name__ == "__main__":
    v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose=True, variant=v))
