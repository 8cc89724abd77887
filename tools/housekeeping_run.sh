set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/st_l1 -o s --output-format csv -- python3 bench.py --lanes 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/st_l1.json 2> $O/st_l1.err
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/st_l3 -o s --output-format csv -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/st_l3.json 2> $O/st_l3.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w profiles/r1_pmc_traffic_fp32.json fused > $O/traffic.txt
cp profiles/r1_pmc_traffic_fp32.json $O/traffic_fp32.json
timeout -k 10 300 python bench.py > $O/final_fp32.json 2> $O/final_fp32.err
timeout -k 10 300 python bench.py --precision bf16 > $O/final_bf16.json 2> $O/final_bf16.err
timeout -k 10 300 python tools/bench_configs.py --precision fp32 > $O/cfg_fp32.json 2> $O/cfg_fp32.err
timeout -k 10 300 python tools/bench_configs.py --precision bf16 > $O/cfg_bf16.json 2> $O/cfg_bf16.err
echo done
