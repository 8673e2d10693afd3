#!/bin/bash
# end-of-round record: full GPU suite, smoke, the driver's bench command, every workload, cornell + ganesha profiles
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r02_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_n1.json').read().strip().splitlines()[-1])
print('headline', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms; roofline', d['roofline']['bound'], round(d['roofline']['frac'],3), 'hbm frac', d['roofline']['hbm']['frac'], 'cpu', d['cpu_baseline']['value'], 'parity', d['parity']['rel_linf_vs_cpu_ref'], 'host', d['host_api'])
PY
bash tools/bench_all.sh
bash tools/collect_profile.sh shirley_1080p_spp64_d8 gpurun_out/r02_shirley > /dev/null 2>&1
bash tools/collect_profile.sh cornell_1024_spp256_d16 gpurun_out/r02_cornell > /dev/null 2>&1
bash tools/collect_profile.sh ganesha_1080p_spp64_d8 gpurun_out/r02_ganesha > /dev/null 2>&1
echo profiles done
