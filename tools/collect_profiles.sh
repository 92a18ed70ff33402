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
python bench.py --steps 30 --no-cpu-baseline --force-dist --collective torch-compact --long-steps 0 > $out/diag_forcedist_torch_compact.json 2>> $out/diag.err
# host side of a data-parallel step at the per-rank load of a 4- / 8-GPU run (round-4 review, item 2a): the same one-rank step through the torch.distributed
# hooks (a ctypes -> Python callback per collective and step) and through the library's own RCCL communicators (no Python in the step): `host` object of the line
for v in 2 4; do for c in torch rccl torch-compact rccl-compact; do python bench.py --views $v --steps 300 --no-cpu-baseline --force-dist --collective $c --long-steps 0 > $out/diag_host_views${v}_$c.json 2>> $out/diag.err; done; done
python bench.py --steps 30 --no-cpu-baseline --force-dist --collective auto --long-steps 0 > $out/diag_forcedist_auto.json 2>> $out/diag.err
python bench.py --config 5 --views 8 --steps 20 --warmup 3 --no-cpu-baseline --sh-fp16 --list-cut 0 > $out/diag_config5views8_shfp16_nocut.json 2>> $out/diag.err
python bench.py --no-cpu-baseline --fuse-update 0 --long-steps 0 > $out/diag_config3_update_launch.json 2>> $out/diag.err
python tools/local_step_at_world.py > $out/local_step_at_world.json 2>> $out/diag.err
python tools/diag_envelope.py 1 s1 s2 2 3 > $out/envelope_diag.txt 2>&1; cp gpurun_out/r5_envelope_diag.json $out/envelope_diag.json
# roctx stage ranges (trainer option "roctx" via GS_ROCTX=1): marker + kernel trace of a short run, no counters in the same run
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && GS_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats -d $out/roctx_raw -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --long-steps 0 --prewarm-seconds 0 > $out/roctx_bench.json 2> $out/roctx.err; f=$(find $out/roctx_raw -name "*marker*stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $out/roctx_marker_stats.csv; find $out/roctx_raw -name "*marker_api_trace.csv" | head -1 | xargs -r head -40 > $out/roctx_marker_trace_head.csv; rm -rf $out/roctx_raw )
# useful-lane fraction of a hit (needs tools/_variants/lib_lanes.so: tools/build_variant.sh lanes k_render.hip -DGS_DIAG_COUNT_ACTIVE)
if [ -f tools/_variants/lib_lanes.so ]; then for c in 2 3; do GSPLAT_MI355_LIB=$PWD/tools/_variants/lib_lanes.so python tools/diag_lanes.py --config $c > $out/lanes_config$c.json 2>> $out/diag.err; done; GSPLAT_MI355_LIB=$PWD/tools/_variants/lib_lanes.so python tools/diag_lanes.py --config 5 --views 8 --steps 2 > $out/lanes_config5views8.json 2>> $out/diag.err; fi
ls -la $out
