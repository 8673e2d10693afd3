#!/bin/bash
# every BASELINE workload on the current build (two-stream default), one line each
for wl in shirley_600x300_spp32_d8 shirley_1080p_spp64_d8 cornell_1024_spp256_d16 ganesha_1080p_spp64_d8 shirley_4k_spp256_d8; do
  echo -n "$wl: "
  timeout -k 10 600 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-workloads --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s  %.2f ms/step '%(d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
done
