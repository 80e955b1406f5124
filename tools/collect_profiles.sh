#!/bin/bash
# Turn the output of tools/gpu/r03_profiles.sh (merged into gpurun_out/r03p/) into the tracked files under profiles/.
set -e
O=gpurun_out/r03p; R=${1:-r03}
PMC_KERNEL_SOURCE_HASH=$(cat $O/kernel_source_hash.txt) python3 tools/pmc_traffic.py profiles/${R}_pmc_traffic $O/pmc_10002434 $O/pmc_49826818 > /dev/null
python3 tools/kernel_stats_summary.py $O/prof_bench10m/b_kernel_stats.csv profiles/${R}_bench10m_kernel_stats.csv
python3 tools/kernel_stats_summary.py $O/prof_bench50m_sgld/b_kernel_stats.csv profiles/${R}_bench50m_sgld_kernel_stats.csv 16 60
python3 tools/kernel_stats_summary.py $O/prof_bench50m_rsghmc/b_kernel_stats.csv profiles/${R}_bench50m_rsghmc_kernel_stats.csv 16 60
python3 tools/kernel_stats_summary.py $O/probe_10002434_stats/s_kernel_stats.csv profiles/${R}_probe_10m_cold_kernel_stats.csv
python3 tools/kernel_stats_summary.py $O/probe_49826818_stats/s_kernel_stats.csv profiles/${R}_probe_50m_cold_kernel_stats.csv
cp $O/bench_driver_cmd_a.json profiles/${R}_bench_driver_cmd_a.json
cp $O/bench_driver_cmd_b.json profiles/${R}_bench_driver_cmd_b.json
cp $O/bench_default.json profiles/${R}_bench_n1.json
cp $O/bench_2000.json profiles/${R}_bench_2000.json
cp $O/bench_2chains_per_gpu.json profiles/${R}_bench_2chains_per_gpu.json
cp $O/bench_50m_sgld.json profiles/${R}_bench_50m_sgld.json
cp $O/bench_50m_rsghmc.json profiles/${R}_bench_50m_rsghmc.json
cp $O/bench_selflaunch_n2_gloo.json profiles/${R}_bench_selflaunch_n2_gloo.json
cp $O/bench_selflaunch_n8_gloo.json profiles/${R}_bench_selflaunch_n8_gloo.json
cp $O/prof_bench10m.json profiles/${R}_bench_n1_under_rocprof.json
cp $O/stats_variant_cost.txt profiles/${R}_stats_variant_cost.txt
cp $O/adapt_sweep.txt profiles/${R}_adapt_sweep.txt
cp $O/pytest_gpu.txt profiles/${R}_pytest_gpu.txt
cp $O/examples.txt profiles/${R}_examples.txt
