# copies what tools/profile_run.sh left under gpurun_out/<round>/ into profiles/ (tracked): tools/collect_profiles.sh r3
R=${1:-r3}; O=gpurun_out/$R; P=profiles
cp $O/final_fp32.json $P/${R}_bench_fp32.json
cp $O/final_fp32_steps20.json $P/${R}_bench_fp32_steps20.json
cp $O/final_bf16.json $P/${R}_bench_bf16.json
cp $O/lines1m_fp32.json $P/${R}_lines1m_fp32.json
cp $O/lines1m_bf16.json $P/${R}_lines1m_bf16.json
cp $O/st_l1/s_kernel_stats.csv $P/${R}_kernel_stats_fp32_serial.csv
cp $O/st_l1.json $P/${R}_bench_fp32_serial_under_rocprof.json
cp $O/st_pipe/s_kernel_stats.csv $P/${R}_kernel_stats_fp32_pipelined.csv
cp $O/traffic_fp32.json $P/${R}_pmc_traffic_fp32.json
cp $O/mfma_fp32.json $P/${R}_pmc_mfma_fp32.json
cp $O/mfma32.txt $P/${R}_pmc_mfma_fp32.txt
cp $O/cfg_fp32.json $P/${R}_configs_3_5_fp32.json
cp $O/st_b16/s_kernel_stats.csv $P/${R}_kernel_stats_bf16_serial.csv
cp $O/traffic_bf16.json $P/${R}_pmc_traffic_bf16.json
cp $O/mfma_bf16.json $P/${R}_pmc_mfma_bf16.json
cp $O/mfma16.txt $P/${R}_pmc_mfma_bf16.txt
