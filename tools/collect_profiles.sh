#!/bin/bash
# Round profile collection on the GPU box (everything the DESIGN / bench numbers cite).  Output: gpurun_out/<tag>/...
#   gpurun --timeout 1200 -- 'tools/collect_profiles.sh <tag>'
set -x
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tools/prof_stats.sh $out/stats50 --steps 50 --warmup 5 --no-cpu-baseline --long-steps 0 > $out/stats50.txt 2>&1
tools/prof_stats.sh $out/stats_views2 --views 2 --steps 300 --warmup 5 --no-cpu-baseline --long-steps 0 > $out/stats_views2.txt 2>&1
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --long-steps 0 --prewarm-seconds 0" tools/pmc_pass.sh $out/pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" > $out/pmc.txt 2>&1
BENCH_ARGS="--config 5 --views 8 --steps 2 --warmup 1 --no-cpu-baseline --long-steps 0 --prewarm-seconds 0" tools/pmc_pass.sh $out/pmc_dense "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" > $out/pmc_dense.txt 2>&1
tools/prof_stats.sh $out/stats_dense --config 5 --views 8 --steps 20 --warmup 3 --no-cpu-baseline --long-steps 0 > $out/stats_dense.txt 2>&1
for c in 1 2 4; do python bench.py --config $c --steps $([ $c = 4 ] && echo 30 || echo 500) --no-cpu-baseline > $out/diag_config$c.json 2>> $out/diag.err; done
python bench.py --config 5 --views 8 --steps 20 --warmup 3 --no-cpu-baseline > $out/diag_config5views8_fp32.json 2>> $out/diag.err
python bench.py --config 5 --views 8 --steps 20 --warmup 3 --no-cpu-baseline --sh-fp16 > $out/diag_config5views8_shfp16.json 2>> $out/diag.err
for v in 2 4 8; do python bench.py --views $v --steps $((600 / v)) --no-cpu-baseline > $out/diag_views$v.json 2>> $out/diag.err; done
python bench.py --steps 30 --no-cpu-baseline --force-dist --collective torch-sharded --long-steps 0 > $out/diag_forcedist_torch_sharded.json 2>> $out/diag.err
ls -la $out
