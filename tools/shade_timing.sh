#!/bin/bash
# where a k_shade wave's life goes (sort / entry / append / store), per bounce: needs build_variants/libptx_shtime.so
# (tools/build_variant.sh shtime "-DPT_SHADE_TIMING=1")
mkdir -p gpurun_out
PTX_LIB=$PWD/build_variants/libptx_shtime.so PTX_STREAMS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-workloads ${BENCH_ARGS} 2> gpurun_out/shade_timing.err > gpurun_out/shade_timing.json
grep shade_timing gpurun_out/shade_timing.err | sort -u | head -40
