#!/bin/bash
set -e
O=gpurun_out
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export CRNN_DBG_SKIP_REC=1; else unset CRNN_DBG_SKIP_REC; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline > $O/x_$v.json 2> $O/x_$v.err
  python -c "
import json; d = json.load(open('$O/x_$v.json')); print('skiprec $v', d['value'], d['ms_per_step'])"
done
