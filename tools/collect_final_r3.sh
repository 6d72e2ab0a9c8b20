#!/bin/bash
# tools/collect_final_r3.sh: copies what tools/final_profiles_r3.sh left in gpurun_out/final3/ into profiles/ under the
# names profiles/README.md lists (run here, after the gpurun call has merged its output back)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/final3; P=$R/profiles
for n in default cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do cp $O/bench_$n.json $P/r03_bench_$n.json; done
cp $O/bench_kernel_stats.csv $P/r03_bench_kernel_stats.csv
cp $O/kernel_stats_single_image.csv $P/r03_kernel_stats_single_image.csv
cp $O/kernel_stats_lfc_block_s_10000.csv $P/r03_kernel_stats_lfc_block_s_10000_final.csv
cp $O/latency.txt $P/r03_latency.txt
cp $O/path_rates.txt $P/r03_path_rates.txt
cat $O/batch_sweep_lfcW1A1.txt $O/batch_sweep_cnvW1A1.txt > $P/r03_batch_sweep.txt
cp $O/pmc_traffic_cnvW1A1.txt $P/r03_pmc_traffic_cnvW1A1.txt
cp $O/pmc_fetch_counter_collection.csv $P/r03_pmc_fetch_counter_collection.csv
cp $O/pmc_write_counter_collection.csv $P/r03_pmc_write_counter_collection.csv
cp $O/sq_bench/sq_summary.json $P/r03_sq_summary.json
cp $O/sq_lfc10k/sq_summary.json $P/r03_sq_lfc_block_s_10000_final.json
cp $O/rccl_world1.json $P/r03_rccl_world1.json
cp $O/rehearse_gloo_2_torchrun.json $P/r03_rehearse_gloo_2_torchrun.json
cp $O/rehearse_gloo_4_plain_start.json $P/r03_rehearse_gloo_4_plain_start.json
cp $O/bench_gpus2_on_one_gpu.json $P/r03_bench_gpus2_on_one_gpu.json
echo collected
