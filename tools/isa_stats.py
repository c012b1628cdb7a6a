"""Static per-kernel ISA statistics of one HIP source (no GPU needed): registers, LDS, scratch and instruction mix from the
gfx950 assembly hipcc emits.  The denoise kernels are VALU-bound, so the static VALU count along the hot path is the offline
proxy for their run time.   python tools/isa_stats.py hybrid_rendering_amd/csrc/denoise_fast.hip [kernel-name-filter]"""
import os, re, subprocess, sys, tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybrid_rendering_amd import build as hb

with tempfile.TemporaryDirectory() as td:
    flags = [f for f in hb.FLAGS if f not in ("-shared",)] + os.environ.get("HR_CFLAGS", "").split()
    subprocess.check_call([hb.hipcc()] + flags + ["-c", "-x", "hip", src, "-o", os.path.join(td, "o.o"), "-save-temps=obj"], stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(td) if f.endswith(".s") and "amdgcn" in f][0]
    text = open(os.path.join(td, asm)).read()

kernels = re.split(r"\n(?=\s*\.globl\s)", text)
for blk in kernels:
    m = re.search(r"\.globl\s+(\S+)", blk)
    if not m or ".amdhsa_kernel" not in blk:
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    body = blk.split(".amdhsa_kernel")[0]
    ins = [l.strip().split()[0] for l in body.splitlines() if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(".")]
    cnt = lambda p: sum(1 for i in ins if re.match(p, i))
    g = lambda k: (re.search(r"\." + k + r"\s+(\d+)", blk) or [None, "?"])[1]
    trans = cnt(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0]
    print(f"{short[-48:]:48s} vgpr {g('amdhsa_next_free_vgpr'):>3} sgpr {g('amdhsa_next_free_sgpr'):>3} lds {g('amdhsa_group_segment_fixed_size'):>6} "
          f"scratch {g('amdhsa_private_segment_fixed_size'):>4} | valu {cnt(r'v_'):5d} (trans {trans:3d}) salu {cnt(r's_'):5d} vmem {cnt(r'(global|buffer|flat|scratch)_'):4d} lds {cnt(r'ds_'):4d}")
