#!/bin/bash
# Times bench.py against variant libraries built by tools/build_variant.sh (timing experiments; results of a variant need not be right).
#   gpurun -- 'tools/variants.sh <tag> <name> [<name> ...]'   ("base" = the in-tree library)
tag=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset GSPLAT_MI355_LIB; else export GSPLAT_MI355_LIB=$PWD/tools/_variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline ${BENCH_EXTRA} > gpurun_out/${tag}_$v.json 2> gpurun_out/${tag}_$v.err || tail -c 500 gpurun_out/${tag}_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$v.json"))
print("$v", round(d["value"], 1), "steps/s", round(d["ms_per_step"], 4), "ms", d["stages_ms_per_launch"])
PY
done
