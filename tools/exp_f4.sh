#!/bin/bash
set -e
O=gpurun_out
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > $O/h_$name.json 2> $O/h_$name.err
  python -c "
import json; d = json.load(open('$O/h_$name.json')); print('$name', d['value'], d['ms_per_step'])"
}
for r in 1 2 3 4; do
run dflt CRNN_WINOGRAD_F4=0x3C
run half CRNN_WINOGRAD_F4=0x3E
run full CRNN_WINOGRAD_F4=0x3E CRNN_WINO4_HALF=0
done
