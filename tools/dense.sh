#!/bin/bash
# Dense-scene iteration: list-exactness tests at cfg5 size + the cfg5-size 8-pass diagnostic bench.
#   gpurun -- 'tools/dense.sh <tag>'
tag=$1
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_raster.py -x -q -m gpu -k "lists_are_the_stable or long_tile or forward_parity" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/${tag}_tests.log
timeout -k 10 300 python bench.py --config 5 --views 8 --steps 20 --warmup 3 --no-cpu-baseline --long-steps 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -c 800 gpurun_out/${tag}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print(round(d["value"], 1), "steps/s", round(d["ms_per_step"], 4), "ms", d["stages_ms_per_launch"])
PY
