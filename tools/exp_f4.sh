#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 180 python tools/exp_logit_error.py
timeout -k 10 400 python -u -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
for v in 0 1 0 1; do
  CRNN_LIN_GEMM=$v timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/lin_$v.json 2> gpurun_out/lin_$v.err
  python -c "
import json; d = json.load(open('gpurun_out/lin_$v.json')); k = d['kernels']
print('lin $v', d['value'], d['ms_per_step'], d['pipeline_executed_mfma_tflops'], ' '.join('%s=%.3f' % (n, k[n]['ms']) for n in ('lstm0_ih','lstm1_ih','classifier','argmax','emb0','conv6')))"
done
