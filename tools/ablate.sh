#!/bin/bash
# GPU developer tool: rebuild the library with ablation macros and time the traced passes (what does each part of a hit-shading
# kernel cost?).  usage (on the GPU box): bash tools/ablate.sh "ddgi,reflections" "" "-DHR_ABL_NO_SECONDARY" "-DHR_ABL_NO_SECONDARY -DHR_ABL_DDGI_NO_IRRADIANCE" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
PASSES=$1; shift
for flags in "$@"; do
    HR_CFLAGS="$flags" python -m hybrid_rendering_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
    echo "== flags: '$flags'"
    python $R/tools/passbench.py --exact 0 --passes $PASSES --frames 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', d['pass_'], d['ms_per_frame'], {k: v['ms'] for k, v in d['stages'].items() if 'trace' in k})
"
done
python -m hybrid_rendering_amd.build --force > /dev/null 2>&1
