#!/bin/bash
# FETCH_SIZE / TCC hit+miss of the render kernels under variant libraries (tools/build_variant.sh).
#   gpurun -- 'tools/variants_pmc.sh <tag> <name> ...'
tag=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset GSPLAT_MI355_LIB; else export GSPLAT_MI355_LIB=$PWD/tools/_variants/lib_$v.so; fi
  BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --long-steps 0" tools/pmc_pass.sh gpurun_out/${tag}_${v}_pmc "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" > gpurun_out/${tag}_${v}_pmc.txt 2>&1
  echo "== $v"; grep -E "k_render_fwd|k_render_bwd_pair" gpurun_out/${tag}_${v}_pmc.txt
done
