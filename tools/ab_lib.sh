#!/bin/bash
# A/B of two builds of the library inside one box: tools/libcrnn_old.so and tools/libcrnn_new.so are copied over
# ocr_amd/libcrnn_hip.so in turn; bench.py pipelined (--steps 40) and --serial for each, twice.
set -e
mkdir -p gpurun_out/r3x
run() { # name, bench args...
  local name=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3x/l_$name.json 2>>gpurun_out/r3x/err.log
}
cp tools/libcrnn_new.so ocr_amd/libcrnn_hip.so
timeout -k 10 300 python -u -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3x/l_tests.log 2>&1
for rep in 1 2; do
cp tools/libcrnn_old.so ocr_amd/libcrnn_hip.so
run old_p$rep --steps 40
run old_s$rep --steps 20 --serial
cp tools/libcrnn_new.so ocr_amd/libcrnn_hip.so
run new_p$rep --steps 40
run new_s$rep --steps 20 --serial
done
