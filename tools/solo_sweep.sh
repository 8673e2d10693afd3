#!/bin/bash
# usage: tools/solo_sweep.sh "<PTX_SOLO_ENTRIES values>"  -> per value: config 1, the 1/8 share of the headline frame (queued steps), the headline,
# cornell and the ganesha-like frame -- what running a batch's last bounces in one launch (k_bounce's PtSolo) is worth, and where it costs
for e in $1; do
  echo "== PTX_SOLO_ENTRIES=$e"
  for wl in shirley_600x300_spp32_d8 shirley_1080p_spp64_d8 cornell_1024_spp256_d16 ganesha_1080p_spp64_d8; do
    echo -n "$wl: "
    PTX_SOLO_ENTRIES=$e timeout -k 10 300 python bench.py --steps ${STEPS:-6} --warmup 1 --no-cpu-baseline --no-workloads --workload $wl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step '%(d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})"
  done
  PTX_SOLO_ENTRIES=$e timeout -k 10 200 python tools/share_step_rate.py 8 60 2>/dev/null | tail -2
done
