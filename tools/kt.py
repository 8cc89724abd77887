"""Print the per-kernel table of a bench.py JSON line (stdin or file)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f lines/s  %.3f ms/step  gemm %.1f TF (%.3f)" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"]))
tot = 0
for k, v in d["kernels"].items():
    tot += v["ms"]
    print("  %-12s %8.4f ms  %s TF  %s GB/s" % (k, v["ms"], v["tflops"], v["gbs_algorithmic"]))
print("  sum %.3f ms" % tot)
