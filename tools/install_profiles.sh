#!/bin/bash
# Copies a tools/collect_profiles.sh output directory into profiles/<round>/ under stable names and refreshes profiles/pmc_latest.json.
#   tools/install_profiles.sh gpurun_out/<tag> profiles/r03 <prefix>
set -e
src=$1; dst=$2; p=$3
mkdir -p $dst
cp $src/bench_default.json $dst/${p}_bench_default.json
cp $src/stats50/kernel_stats.csv $dst/${p}_kernel_stats_bench_steps50.csv
cp $src/stats50/bench_under_rocprof.json $dst/${p}_bench_under_rocprof.json
cp $src/stats_views2/kernel_stats.csv $dst/${p}_kernel_stats_views2_steps300.csv
cp $src/stats_dense/kernel_stats.csv $dst/${p}_kernel_stats_config5views8_steps20.csv
cp $src/pmc/summary.json $dst/${p}_pmc_per_kernel.json
cp $src/pmc_dense/summary.json $dst/${p}_pmc_per_kernel_config5views8.json
for f in $src/diag_*.json $src/lanes_*.json; do [ -f "$f" ] && tail -1 $f > $dst/${p}_$(basename $f); done
[ -f $src/local_step_at_world.json ] && cp $src/local_step_at_world.json $dst/${p}_local_step_at_world.json
[ -f $src/envelope_diag.json ] && cp $src/envelope_diag.json $dst/${p}_envelope_diag.json
for f in roctx_marker_stats.csv roctx_marker_trace_head.csv; do [ -f $src/$f ] && cp $src/$f $dst/${p}_$f; done
python3 tools/pmc_to_latest.py $src/pmc/summary.json profiles/pmc_latest.json > /dev/null
echo installed into $dst
