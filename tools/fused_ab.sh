#!/bin/bash
# usage (GPU box): FUSED="0 1 2" STREAMS="1 2" tools/fused_ab.sh <workload> [steps]   -> PTX_FUSED x PTX_STREAMS, interleaved twice
wl=${1:-shirley_1080p_spp64_d8}; steps=${2:-5}
for rep in 1 2; do
  for f in ${FUSED:-0 1 2}; do
    for ns in ${STREAMS:-1 2}; do
      echo -n "fused=$f streams=$ns: "
      PTX_FUSED=$f PTX_STREAMS=$ns timeout -k 10 300 python bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-workloads --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s  %.2f ms/step '%(d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
    done
  done
done
