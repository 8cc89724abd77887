#!/bin/bash
# A/B of one environment switch inside one box: tools/ab_env.sh NAME=VALUE  (bench.py pipelined and --serial, twice each)
set -e
mkdir -p gpurun_out/r3x
run() { # name, env, bench args...
  local name=$1; shift
  local e=$1; shift
  env "$e" timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3x/e_$name.json 2>>gpurun_out/r3x/err.log
}
for rep in 1 2; do
run off_p$rep "$1" --steps 40
run off_s$rep "$1" --steps 20 --serial
run on_p$rep CRNN_X=1 --steps 40
run on_s$rep CRNN_X=1 --steps 20 --serial
done
