#!/bin/bash
# usage: tools/bench_variants.sh a b c ...   (libraries build_variants/libptx_<x>.so); env STREAMS="2 1" picks the schedules
for v in "$@"; do
  for ns in ${STREAMS:-2 1}; do
    echo -n "variant $v streams $ns: "
    PTX_STREAMS=$ns PTX_LIB=$PWD/build_variants/libptx_$v.so timeout -k 10 300 python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-workloads ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s  %.2f ms/step '%(d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
  done
done
