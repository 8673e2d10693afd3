#!/bin/bash
# usage (GPU box): FUSED="1 2" STREAMS="1 2" tools/fused_variants.sh <workload> a b c ...  (build_variants/libptx_<x>.so)
wl=$1; shift
for v in "$@"; do
  for f in ${FUSED:-1}; do
    for ns in ${STREAMS:-1 2}; do
      echo -n "variant $v fused=$f streams=$ns: "
      PTX_LIB=$PWD/build_variants/libptx_$v.so PTX_FUSED=$f PTX_STREAMS=$ns timeout -k 10 300 python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-workloads --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s  %.2f ms/step '%(d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
    done
  done
done
