cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3
for st in 20 100; do
for v in "192 2" "224 2" "256 2" "160 2" "192 3" "256 3" "192 2" "256 2"; do
set -- $v
CRNN_PIPELINE_BLOCKS=$1 timeout -k 10 200 python bench.py --steps $st --lanes $2 --no-cpu-baseline > gpurun_out/r3/pb_$1_$2.json 2>/dev/null || exit 1
python - <<PY
import json
j=json.load(open("gpurun_out/r3/pb_$1_$2.json")); print("steps $st blocks=$1 lanes=$2",j["value"],j["ms_per_step"], j["config"].get("split_timeouts"))
PY
done
done
