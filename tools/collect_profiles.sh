#!/bin/bash
# Turn the output of tools/gpu/profiles.sh (merged into gpurun_out/<round>p/) into the tracked files under profiles/:
#   bash tools/collect_profiles.sh r06
set -e
R=${1:-r06}; O=gpurun_out/${R}p
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $O/pmc_traffic.md profiles/${R}_pmc_traffic.md
python3 tools/kernel_stats_summary.py $O/prof_bench10m/b_kernel_stats.csv profiles/${R}_bench10m_kernel_stats.csv
python3 tools/kernel_stats_summary.py $O/prof_bench50m_rsghmc/b_kernel_stats.csv profiles/${R}_bench50m_rsghmc_kernel_stats.csv 16 60
python3 tools/kernel_stats_summary.py $O/prof_bench50m_sgld/b_kernel_stats.csv profiles/${R}_bench50m_sgld_kernel_stats.csv 16 60
python3 tools/kernel_stats_summary.py $O/prof_sinc/b_kernel_stats.csv profiles/${R}_sinc_bnn_kernel_stats.csv 8 20
python3 tools/kernel_stats_summary.py $O/probe_10002434_stats/s_kernel_stats.csv profiles/${R}_probe_10m_cold_kernel_stats.csv
python3 tools/kernel_stats_summary.py $O/probe_49826818_stats/s_kernel_stats.csv profiles/${R}_probe_50m_cold_kernel_stats.csv
for f in driver_cmd_a driver_cmd_b 2000 2chains_per_gpu 50m_sgld 50m_rsghmc sinc_bnn 10m_f64 selflaunch_n2_gloo selflaunch_n8_gloo; do
  cp $O/bench_$f.json profiles/${R}_bench_$f.json
done
cp $O/bench_default.json profiles/${R}_bench_n1.json
cp $O/prof_bench10m.json profiles/${R}_bench_n1_under_rocprof.json
cp $O/pytest_gpu.txt profiles/${R}_pytest_gpu.txt
cp $O/examples.txt profiles/${R}_examples.txt
cp $O/step_timeline.txt profiles/${R}_step_timeline.txt
cp $O/step_timeline_f64.txt profiles/${R}_step_timeline_f64.txt
