#!/bin/bash
# usage (on an MI355X box): tools/round_record.sh <tag>      e.g. r05_final
# The end-of-round record: full GPU suite log, smoke, the driver's bench command (with its `workloads` block), every workload as its
# own bench line, the share timings -> gpurun_out/<tag>_*  (copy what should be judged into profiles/).
tag=${1:-r05_final}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/${tag}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err; echo "bench exit $?"
python - "$tag" <<'PY'
import json, sys
d=json.loads(open('gpurun_out/%s_bench_n1.json' % sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print('headline', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms; roofline', r['kernel'], r['bound'], 'frac', round(r['frac'],3),
      'incl. est. shade', round((r.get('incl_estimated_shade') or {}).get('frac', 0),3), 'frame valu', round(r['frame']['frac'],3),
      'hbm_frame', round(r['hbm_frame']['frac'],3), 'calibrated', r['hbm_frame'].get('calibrated'), 'cpu', d['cpu_baseline']['value'],
      'parity', d['parity']['rel_linf_vs_cpu_ref'], 'host', d['host_api']['ptx_render_ms'])
for k, v in d.get('workloads', {}).items():
    print(' ', k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a in ('ms_per_step', 'value', 'dominant_kernel', 'dominant_kernel_ms_one_stream', 'error')},
          'parity', (v.get('parity') or {}).get('rel_linf_vs_cpu_ref'), 'hbm_frame', round(((v.get('hbm_frame') or {}).get('frac') or 0), 3))
PY
# the multi-rank line on this one GPU: two ranks sharing the device over gloo (host-staged exchange) -- timed-frame parity over both
# ranks' bands, the CPU leg, config 5 across the ranks, the collective block
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 > gpurun_out/${tag}_bench_gloo2.json 2> gpurun_out/${tag}_bench_gloo2.err; echo "gloo-2 bench exit $?"
python - "$tag" <<'PY'
import json, sys
d=json.loads([l for l in open('gpurun_out/%s_bench_gloo2.json' % sys.argv[1]) if l.startswith('{')][-1])
w=d['workloads']['shirley_4k_spp256_d8']
print('gloo-2 on one GPU: headline', round(d['ms_per_step'],2), 'ms, timed-frame parity', d['parity']['timed_frame']['rel_linf_vs_cpu_ref'], 'ranks covered', d['parity']['timed_frame']['ranks_covered'],
      '; world', d['collective']['world'], 'render ms per rank', [round(x,2) for x in d['collective']['render_ms_per_rank']], 'gather', round(d['collective']['rank0_gather_ms'],2),
      '; config 5:', round(w['ms_per_step'],1), 'ms, parity', w['parity']['rel_linf_vs_cpu_ref'], '; cpu', round(d['cpu_baseline']['value'],1))
PY
bash tools/bench_all.sh
python tools/band_share_timing.py ${tag%_final} 2>/dev/null | tail -8
python tools/share_step_rate.py 8 60 2>/dev/null | tail -2
# soak: fresh seeds beyond the suite's fixed ones (random render configurations bit for bit against the oracle; random soups of the fuzz module)
if [ "${SOAK:-1}" != "0" ]; then
  PTX_TEST_SEEDS=${SOAK_RENDER_SEEDS:-5000} PTX_FUZZ_SEEDS=${SOAK_FUZZ_SEEDS:-500} timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q \
    -k "random_configs or soup" > gpurun_out/${tag}_soak.log 2>&1; echo "soak exit $?"; tail -1 gpurun_out/${tag}_soak.log
fi
