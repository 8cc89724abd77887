#!/bin/bash
# F(4x4) Winograd experiment: bench across masks / shapes (same box)
set -e
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/f4_$name.json 2> gpurun_out/f4_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.load(open("gpurun_out/f4_%s.json" % sys.argv[1]))
k = d["kernels"]
print("%-10s %8.1f lines/s %6.3f ms | exec %6.1f TF | cls %.3f conv6 %.3f | f4/conv: %s" % (sys.argv[1], d["value"], d["ms_per_step"],
      d["pipeline_executed_mfma_tflops"], k["classifier"]["ms"], k["conv6"]["ms"],
      " ".join("%s=%.3f" % (n, v["ms"]) for n, v in k.items() if n.endswith("_f4") or n in ("conv1_pool1", "conv2", "conv3", "conv4", "conv5"))), flush=True)
PY
}
run base   CRNN_WINOGRAD_F4=0
run f4_3e  CRNN_WINOGRAD_F4=0x3E
run f4_3c  CRNN_WINOGRAD_F4=0x3C
run f4_38  CRNN_WINOGRAD_F4=0x38
run f4_3e2 CRNN_WINOGRAD_F4=0x3E CRNN_WINO4_SHAPE=2
run base2  CRNN_WINOGRAD_F4=0
run f4_3eb CRNN_WINOGRAD_F4=0x3E
