R=${GRAFT_REPO_ROOT:-/root/repo}
for res in "1920 1080" "3840 2160"; do set -- $res
for v in ${VARIANTS:-base}; do
    if [ "$v" = base ]; then unset HR_LIBRARY; else export HR_LIBRARY=$R/hybrid_rendering_amd/variants/libhybrid_rendering_amd.$v.so; fi
    echo "== $v $1x$2"
    timeout 300 python $R/tools/passbench.py --width $1 --height $2 --passes ${PASSES:-shadows,ao,reflections,ddgi} --frames 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', d['pass_'], d['ms_per_frame'], {k: round(v['ms']*1e3,1) for k, v in d['stages'].items()})
"
done; done
