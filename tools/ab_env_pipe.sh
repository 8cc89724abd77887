#!/bin/bash
# Pipelined-only A/B of environment switches inside one box: tools/ab_env_pipe.sh REPS "A=1" "B=2 C=3" ...
# (an argument "ARGS:--lanes 3" passes bench.py arguments instead)
# bench.py --steps 100 with no switch ("base") and with each argument's switches, alternating, REPS times.
set -e
R=$1; shift
mkdir -p gpurun_out/r3x
for rep in $(seq 1 $R); do
  i=0
  for e in "CRNN_X=1" "$@"; do
    a=""
    case "$e" in ARGS:*) a="${e#ARGS:}"; e="CRNN_X=1";; esac     # "ARGS:--lanes 3": bench.py arguments instead of a switch
    timeout -k 10 200 env $e python bench.py --no-cpu-baseline --steps 100 $a > gpurun_out/r3x/ep_${i}_$rep.json 2>>gpurun_out/r3x/err.log
    python - "$e" $rep gpurun_out/r3x/ep_${i}_$rep.json "$a" <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[4] if len(sys.argv) > 4 else "", sys.argv[2], d["value"], d["ms_per_step"], flush=True)
PY
    i=$((i+1))
  done
done
