#!/bin/bash
# Per-kernel time table of a bench run: rocprofv3 --kernel-trace --stats (no counters in the same run).
#   tools/prof_stats.sh <out_dir> [bench args...]      -> <out_dir>/kernel_stats.csv, <out_dir>/bench_under_rocprof.json
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$out/raw" -o p --output-format csv -- python3 bench.py "$@" > "$out/bench_under_rocprof.json" 2> "$out/rocprof.err" || { tail -5 "$out/rocprof.err"; exit 1; }
f=$(find "$out/raw" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/raw"
cat "$out/kernel_stats.csv"
