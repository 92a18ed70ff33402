#!/bin/bash
# Collect hardware counters for the bench's kernels, one `rocprofv3 --pmc` pass per counter group (never combined
# with a trace domain), and print per-kernel sums of the LARGEST dispatch of each kernel (the 16-view launch).
#   tools/pmc_pass.sh <out_dir> "<counter group 1>" "<counter group 2>" ...   [env BENCH_ARGS="--steps 2 --warmup 1"]
# Run on the GPU box:  gpurun -- 'tools/pmc_pass.sh gpurun_out/pmc "SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"'
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args=${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --long-steps 0 --prewarm-seconds 0}
i=0
for grp in "$@"; do
    i=$((i + 1))
    timeout -k 10 400 rocprofv3 --pmc $grp -d "$out/p$i" -o p --output-format csv -- python3 bench.py $args > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/p$i.log"; }
done
python3 - "$out" <<'EOF'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> value
    grid = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        grid[(k, r["Dispatch_Id"])] = int(r.get("Grid_Size", 0) or 0)
    best = {}
    for (k, d), g in grid.items():
        if k not in best or g >= grid[(k, best[k])]:   # ties: the LAST such dispatch (an arena-overflow replay exits early)
            best[k] = d
    for k, d in best.items():
        res[k].update(per[(k, d)])
        res[k]["grid"] = grid[(k, d)]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k, {a: (int(b) if b == int(b) else b) for a, b in v.items()})
EOF
