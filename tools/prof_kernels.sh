#!/bin/bash
# Steady-state duration of kernels matching a pattern under a variant library: rocprofv3 --kernel-trace of a short bench run, the
# median of every kernel's last 20 launches.
#   gpurun -- 'tools/prof_kernels.sh <tag> <variant|base> <pattern> [bench args ...]'
tag=$1; v=$2; pat=$3; shift 3
if [ "$v" != base ]; then export GSPLAT_MI355_LIB=$PWD/tools/_variants/lib_$v.so; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/${tag}_$v -o p --output-format csv -- python3 bench.py --no-cpu-baseline --long-steps 0 "$@" > gpurun_out/${tag}_$v.log 2>&1
python3 - "$tag" "$v" "$pat" <<'PY'
import csv, glob, sys, statistics, collections
tag, v, pat = sys.argv[1:4]
f = glob.glob(f"gpurun_out/{tag}_{v}/**/*kernel_trace.csv", recursive=True)[0]
per = collections.defaultdict(list)
for x in csv.DictReader(open(f)):
    k = x["Kernel_Name"].split("(")[0]
    if pat in k:
        per[k].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1000.0)
for k, d in per.items():
    print(v, k, "calls", len(d), "median of last 20: %.1f us" % statistics.median(d[-20:]))
PY
