"""Developer tool: percentiles of the per-tile wave lifetimes a trace launch recorded (csrc/tile_order.h; HR_DEBUG_TILE_COSTS=<prefix> makes
every pass write <prefix>.<pass> after each trace launch).  python tools/tile_cost_stats.py <prefix>.shadows [...]"""
import sys
import numpy as np

for path in sys.argv[1:]:
    c = np.fromfile(path, dtype=np.uint16).astype(np.float64) * 0.01   # us
    q = np.percentile(c, [10, 50, 75, 90, 99, 100])
    print('%-40s tiles %6d  us: p10 %6.2f  p50 %6.2f  p75 %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f   p90/p50 %.2f  p99/p50 %.2f  sum %.0f' % (
        path.split('/')[-1], len(c), q[0], q[1], q[2], q[3], q[4], q[5], q[3] / max(q[1], 0.01), q[4] / max(q[1], 0.01), c.sum()))
