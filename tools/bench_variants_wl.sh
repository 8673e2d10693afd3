#!/bin/bash
# usage: tools/bench_variants_wl.sh <workload> a b c ...   (build_variants/libptx_<x>.so); env STREAMS="2 1" picks the schedules
wl=$1; shift
BENCH_ARGS="--workload $wl ${BENCH_ARGS}" bash "$(dirname "$0")/bench_variants.sh" "$@"
