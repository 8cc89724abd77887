# Issue-side counters of every kernel of a serial step (one pass): tools/pmc_valu.sh <outdir>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r3x/valu}
mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $O -o v --output-format csv -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $O/run.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    if "out_in" in k or "wino4_in_h3" in k or "wino4_out" in k:
        print(k, {a: round(b / 1e6, 2) for a, b in c.items()})
PY
