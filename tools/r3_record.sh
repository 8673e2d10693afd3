#!/bin/bash
# the end-of-round record: full GPU suite log, smoke, the driver's bench command -> gpurun_out/r03_final_*
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03_final_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r03_final_bench_n1.json 2> gpurun_out/r03_final_bench_n1.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_final_bench_n1.json').read().strip().splitlines()[-1])
r=d['roofline']
print('headline', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms; roofline', r['kernel'], r['bound'], round(r['frac'],3), 'frame valu', round(r['frame']['frac'],3), 'hbm_frame', round(r['hbm_frame']['frac'],3), 'cpu', d['cpu_baseline']['value'], 'parity', d['parity']['rel_linf_vs_cpu_ref'], 'host', d['host_api']['ptx_render_ms'])
PY
bash tools/bench_all.sh
