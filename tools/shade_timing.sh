#!/bin/bash
# where a k_shade_pool wave's life goes (classify / entry / push), per bounce (two-kernel schedule: PTX_FUSED=0 PTX_FUSED_GLOBAL=0): needs build_variants/libptx_shtime.so
# (tools/build_variant.sh shtime "-DPT_SHADE_TIMING=1")
mkdir -p gpurun_out
PTX_FUSED=0 PTX_FUSED_GLOBAL=0 PTX_LIB=$PWD/build_variants/libptx_shtime.so PTX_STREAMS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-workloads ${BENCH_ARGS} 2> gpurun_out/shade_timing.err > gpurun_out/shade_timing.json
grep shade_timing gpurun_out/shade_timing.err | sort -u | head -40
