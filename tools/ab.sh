#!/bin/bash
# usage: tools/ab.sh "<variants>" "<workloads>" [rounds]  -- interleaved A/B of build_variants/libptx_<v>.so, two-stream schedule, one line per run
vs=$1; wls=$2; rounds=${3:-2}
for r in $(seq $rounds); do
  for wl in $wls; do
    for v in $vs; do
      echo -n "$wl $v: "
      PTX_LIB=$PWD/build_variants/libptx_$v.so timeout -k 10 300 python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-workloads --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step '%(d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
    done
  done
done
