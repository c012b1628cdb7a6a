"""Quick per-stage timing of one pass at bench resolution (developer tool)."""
import sys, json
sys.argv = [sys.argv[0]] + sys.argv[1:]
import subprocess
out = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--warmup", "10", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
try:
    j = json.loads(out.stdout.strip().splitlines()[-1])
    print("value %.1f Mrays/s  ms/step %.4f  trace-only %.1f" % (j["value"], j["ms_per_step"], j["trace_only_Mrays_per_s"]))
    for k, v in j["stages"].items():
        print("  %-24s %8.4f ms  %7.1f GB/s  %5.1f%%" % (k, v["ms"], v["GBps"], 100 * v["frac"]))
except Exception as e:
    print(out.stdout[-2000:], out.stderr[-3000:], e)
