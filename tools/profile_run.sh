# One gpurun call that produces everything under profiles/ for a round: tools/profile_run.sh r3
# (rocprofv3 kernel-trace stats, PMC passes in runs of their own, the bench lines, configs 3/4/5).
set -e
R=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
timeout -k 10 300 python bench.py > $O/final_fp32.json 2> $O/final_fp32.err
timeout -k 10 300 python bench.py --steps 20 > $O/final_fp32_steps20.json 2> $O/final_fp32_steps20.err
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/st_l1 -o s --output-format csv -- python3 bench.py --serial --steps 20 --warmup 3 --no-cpu-baseline > $O/st_l1.json 2> $O/st_l1.err
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/st_pipe -o s --output-format csv -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/st_pipe.json 2> $O/st_pipe.err
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/traffic_fp32.json $O/st_l1.json > $O/traffic.txt
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_m32 -o m --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_m32.log 2>&1
python tools/pmc_mfma.py $O/pmc_m32 $O/mfma_fp32.json f16 > $O/mfma32.txt
timeout -k 10 300 python bench.py --precision bf16 > $O/final_bf16.json 2> $O/final_bf16.err
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/st_b16 -o s --output-format csv -- python3 bench.py --precision bf16 --serial --steps 20 --warmup 3 --no-cpu-baseline > $O/st_b16.json 2> $O/st_b16.err
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fb -o f --output-format csv -- python3 bench.py --precision bf16 --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fb.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_wb -o w --output-format csv -- python3 bench.py --precision bf16 --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_wb.log 2>&1
python tools/pmc_traffic.py $O/pmc_fb $O/pmc_wb $O/traffic_bf16.json $O/st_b16.json > $O/traffic_bf16.txt
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_m16 -o m --output-format csv -- python3 bench.py --precision bf16 --serial --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_m16.log 2>&1
python tools/pmc_mfma.py $O/pmc_m16 $O/mfma_bf16.json bf16 > $O/mfma16.txt
timeout -k 10 300 python bench.py --lines 1000000 > $O/lines1m_fp32.json 2> $O/lines1m_fp32.err
timeout -k 10 300 python bench.py --lines 1000000 --precision bf16 > $O/lines1m_bf16.json 2> $O/lines1m_bf16.err
timeout -k 10 300 python tools/bench_configs.py --precision fp32 > $O/cfg_fp32.json 2> $O/cfg_fp32.err
echo done
