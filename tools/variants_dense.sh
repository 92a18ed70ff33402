#!/bin/bash
# The cfg5-size 8-pass diagnostic bench under variant libraries (tools/build_variant.sh).
#   gpurun -- 'tools/variants_dense.sh <tag> <name> ...'   ("base" = the in-tree library)
tag=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset GSPLAT_MI355_LIB; else export GSPLAT_MI355_LIB=$PWD/tools/_variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --config 5 --views 8 --steps 20 --warmup 3 --no-cpu-baseline --long-steps 0 > gpurun_out/${tag}_$v.json 2> gpurun_out/${tag}_$v.err || tail -c 500 gpurun_out/${tag}_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$v.json"))
print("$v", round(d["value"], 1), "steps/s", round(d["ms_per_step"], 4), "ms", d["stages_ms_per_launch"])
PY
done
