#!/bin/bash
# Kernel iteration loop on the GPU box: backward parity tests, a bench line, optionally counters of the dominant kernel.
#   gpurun -- 'tools/iter.sh <tag> [pmc]'      -> gpurun_out/<tag>_{tests.log,bench.json,pmc/}
tag=$1
timeout -k 10 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_trainer.py -x -q -m gpu -k "backward or reduce or step_sgd or fused or reproduc" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${tag}_tests.log
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --long-steps 300 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -c 800 gpurun_out/${tag}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print(round(d["value"], 1), "steps/s", round(d["ms_per_step"], 4), "ms", d["stages_ms_per_launch"])
PY
if [ "$2" = "pmc" ]; then
  BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --long-steps 0 --prewarm-seconds 0" tools/pmc_pass.sh gpurun_out/${tag}_pmc \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" > gpurun_out/${tag}_pmc.txt 2>&1
  grep -E "k_render_bwd_pair|k_render_fwd" gpurun_out/${tag}_pmc.txt
fi
