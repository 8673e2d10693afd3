#!/bin/bash
# round-2 first GPU pass: the whole GPU suite, the headline bench, and the bounce-packet experiment
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r2_pytest_gpu.log
tail -3 gpurun_out/r2_pytest_gpu.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_base.json 2> gpurun_out/r2_bench_base.err
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r2_bench_base.json').read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
PY
for bp in 1 2 8; do
  PTX_BOUNCE_PACKET=$bp python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_bp$bp.json 2> gpurun_out/r2_bench_bp$bp.err
  python - $bp <<'PY'
import json,sys; d=json.loads(open('gpurun_out/r2_bench_bp%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); print('bounce_packet', sys.argv[1], d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
PY
done
for bp in 0 1 8; do
  PTX_STREAMS=1 PTX_BOUNCE_PACKET=$bp python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_1s_bp$bp.json 2> gpurun_out/r2_bench_1s_bp$bp.err
  python - $bp <<'PY'
import json,sys; d=json.loads(open('gpurun_out/r2_bench_1s_bp%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); print('one stream, bounce_packet', sys.argv[1], d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
PY
done
