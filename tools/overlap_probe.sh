#!/bin/bash
# How much of a step's latency-bound work can hide under another stream's render kernels?  An upper bound without writing the
# two-stream trainer: TWO independent processes on the one GPU, each running the cfg3 step on half of the cameras (--views 8),
# against one process on all sixteen views.  Aggregate views/s of the pair vs the single process = the ceiling of any
# camera-group pipelining inside one trainer (which would still have to join for the reduce and the update).
#   gpurun -- 'tools/overlap_probe.sh <tag>'
tag=$1
python bench.py --steps 400 --no-cpu-baseline --long-steps 0 > gpurun_out/${tag}_single16.json 2>/dev/null
python bench.py --views 8 --steps 800 --no-cpu-baseline --long-steps 0 > gpurun_out/${tag}_single8.json 2>/dev/null
python bench.py --views 8 --steps 800 --no-cpu-baseline --long-steps 0 > gpurun_out/${tag}_pairA.json 2>/dev/null &
pa=$!
python bench.py --views 8 --steps 800 --no-cpu-baseline --long-steps 0 > gpurun_out/${tag}_pairB.json 2>/dev/null &
pb=$!
wait $pa; wait $pb
python - <<PY
import json
g = lambda n: json.load(open("gpurun_out/${tag}_%s.json" % n))
s16, s8, a, b = g("single16"), g("single8"), g("pairA"), g("pairB")
print("one process, 16 views: %.1f steps/s = %.0f views/s" % (s16["value"], 16 * s16["value"]))
print("one process,  8 views: %.1f steps/s = %.0f views/s" % (s8["value"], 8 * s8["value"]))
print("two processes, 8 views each, side by side: %.1f + %.1f steps/s = %.0f views/s (%.3f x the 16-view process)" % (a["value"], b["value"], 8 * (a["value"] + b["value"]), 8 * (a["value"] + b["value"]) / (16 * s16["value"])))
PY
