set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/st_l1 -o s --output-format csv -- python3 bench.py --serial --steps 20 --warmup 3 --no-cpu-baseline > $O/st_l1.json 2> $O/st_l1.err
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/st_l3 -o s --output-format csv -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/st_l3.json 2> $O/st_l3.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w profiles/r1_pmc_traffic_fp32.json f4 > $O/traffic.txt
cp profiles/r1_pmc_traffic_fp32.json $O/traffic_fp32.json
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_m32 -o m --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_m32.log 2>&1
python tools/pmc_mfma.py $O/pmc_m32 profiles/r1_pmc_mfma_fp32.json f32 > $O/mfma32.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_m16 -o m --output-format csv -- python3 bench.py --precision bf16 --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_m16.log 2>&1
python tools/pmc_mfma.py $O/pmc_m16 profiles/r1_pmc_mfma_bf16.json bf16 > $O/mfma16.txt
cp profiles/r1_pmc_mfma_fp32.json $O/mfma_fp32.json; cp profiles/r1_pmc_mfma_bf16.json $O/mfma_bf16.json
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/st_b16 -o s --output-format csv -- python3 bench.py --precision bf16 --serial --steps 20 --warmup 3 --no-cpu-baseline > $O/st_b16.json 2> $O/st_b16.err
timeout -k 10 300 python bench.py > $O/final_fp32.json 2> $O/final_fp32.err
timeout -k 10 300 python bench.py --precision bf16 > $O/final_bf16.json 2> $O/final_bf16.err
timeout -k 10 300 python tools/bench_configs.py --precision fp32 > $O/cfg_fp32.json 2> $O/cfg_fp32.err
timeout -k 10 300 python tools/bench_configs.py --precision bf16 > $O/cfg_bf16.json 2> $O/cfg_bf16.err
echo done
