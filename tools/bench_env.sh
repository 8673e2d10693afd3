#!/bin/bash
# usage: tools/bench_env.sh <workload> "ENV1=a ENV2=b" "ENV1=c" ...   -> one bench line per environment (default library)
wl=$1; shift
for e in "$@"; do
  for ns in ${STREAMS:-2 1}; do
    echo -n "env [$e] streams $ns: "
    env $e PTX_STREAMS=$ns timeout -k 10 300 python bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-workloads --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s  %.2f ms/step '%(d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
  done
done
