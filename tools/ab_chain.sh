#!/bin/bash
# A/B of the chained transform kernels inside one box: bench.py with CRNN_WINO4_CHAIN off / on / one layer at a time
set -e
mkdir -p gpurun_out/r3x
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --no-cpu-baseline > gpurun_out/r3x/c_$name.json 2>>gpurun_out/r3x/err.log
}
for rep in 1 2; do
run off_$rep CRNN_WINO4_CHAIN=0
run on_$rep CRNN_X=1
run m30_$rep CRNN_WINO4_CHAIN=0x30
done
