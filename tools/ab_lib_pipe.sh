#!/bin/bash
# Pipelined-only A/B of two library builds inside one box (tools/libcrnn_old.so, tools/libcrnn_new.so): bench.py --steps 100,
# alternating, REPS times each (default 4; STEPS=20 for the driver's form).  Prints lines/s per run.
set -e
R=${1:-4}
mkdir -p gpurun_out/r3x
for rep in $(seq 1 $R); do
  for w in old new; do
    cp tools/libcrnn_$w.so ocr_amd/libcrnn_hip.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-100} > gpurun_out/r3x/lp_${w}_$rep.json 2>>gpurun_out/r3x/err.log
    python - <<PY
import json
d=json.loads(open("gpurun_out/r3x/lp_${w}_$rep.json").read().strip().splitlines()[-1])
print("$w", $rep, d["value"], d["ms_per_step"])
PY
  done
done
cp tools/libcrnn_new.so ocr_amd/libcrnn_hip.so
