#!/bin/bash
# streams x passes-per-batch sweep on the headline workload (current build)
for ns in 1 2 3 4; do for ppb in 32 16 8; do
  echo -n "streams $ns ppb $ppb: "
  PTX_STREAMS=$ns timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-workloads --passes-per-batch $ppb 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s  %.2f ms/step '%(d['value'], d['ms_per_step']))"
done; done
