#!/bin/bash
# GPU developer tool: time passes with variant libraries built beforehand (no compiler time on the GPU box):
#   HR_CFLAGS="-DHR_COOP_PUSH=3" python -m hybrid_rendering_amd.build --variant push3     (here, any number of variants)
#   gpurun -- 'bash tools/ab.sh ddgi,reflections base push3'                              ("base" = the product library)
R=${GRAFT_REPO_ROOT:-/root/repo}
PASSES=$1; shift
for v in "$@"; do
    if [ "$v" = base ]; then unset HR_LIBRARY; else export HR_LIBRARY=$R/hybrid_rendering_amd/variants/libhybrid_rendering_amd.$v.so; fi
    echo "== $v"
    timeout 300 python $R/tools/passbench.py --exact ${EXACT:-0} --passes $PASSES --frames ${FRAMES:-20} ${PB_ARGS:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', d['pass_'], d['ms_per_frame'], {k: v['ms'] for k, v in d['stages'].items()})
"
done
