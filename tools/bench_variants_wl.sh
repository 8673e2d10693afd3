#!/bin/bash
# usage: tools/bench_variants_wl.sh <workload> a b c ...
wl=$1; shift
for v in "$@"; do
  echo -n "variant $v: "
  PTX_LIB=$PWD/build_variants/libptx_$v.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s  %.2f ms/step '%(d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
done
