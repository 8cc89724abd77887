#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace --stats, bench.py --serial) with one environment switch off and on, one box:
# tools/ab_env_prof.sh NAME=VALUE
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x
mkdir -p $O
env "$1" timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/p_off -o s --output-format csv -- python3 bench.py --serial --steps 20 --warmup 3 --no-cpu-baseline > $O/p_off.json 2> $O/p_off.err
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/p_on -o s --output-format csv -- python3 bench.py --serial --steps 20 --warmup 3 --no-cpu-baseline > $O/p_on.json 2> $O/p_on.err
for w in off on; do echo "== $w"; grep -E "cls_select|gemm16_dma" $O/p_$w/*kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)"/"/'; done
