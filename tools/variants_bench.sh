#!/bin/bash
# bench the base library and every tools/_variants/lib_<name>.so given, same box, same arguments; prints steps/s and the render stages
#   gpurun -- 'tools/variants_bench.sh <tag> name1 name2 ...'
tag=$1; shift
run() { # label, lib or empty
  if [ -n "$2" ]; then export GSPLAT_MI355_LIB=$2; else unset GSPLAT_MI355_LIB; fi
  python bench.py --steps 200 --no-cpu-baseline --long-steps 0 > gpurun_out/${tag}_$1.json 2> gpurun_out/${tag}_$1.err || { echo "$1 FAILED"; tail -3 gpurun_out/${tag}_$1.err; return; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$1.json")); s = d["stages_ms_per_launch"]
print("%-10s %.1f steps/s  fwd %.4f bwd %.4f (timed-region bwd %.4f)" % ("$1", d["value"], s["render_forward"], s["render_backward"], d["roofline"]["ms_per_launch"]))
PY
}
run base ""
for n in "$@"; do run $n tools/_variants/lib_$n.so; done
run base2 ""
