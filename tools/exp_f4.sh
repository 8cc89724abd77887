#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for v in 0 -1 0 -1; do
  if [ $v = -1 ]; then unset CRNN_LIN_GN; else export CRNN_LIN_GN=$v; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline > $O/gn_$v.json 2> $O/gn_$v.err
  python -c "
import json; d = json.load(open('$O/gn_$v.json')); k = d['kernels']
print('gn $v', d['value'], d['ms_per_step'], d['pipeline_executed_mfma_tflops'], ' '.join('%s=%.3f' % (n, k[n]['ms']) for n in ('lstm0_ih','lstm1_ih','classifier')))"
done
unset CRNN_LIN_GN
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/traffic_fp32.json f4 | grep -E "ih|classifier|_f4"
