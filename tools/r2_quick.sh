#!/bin/bash
# quick GPU check: parity tests that cover the render path + bench (two streams and one stream)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/q_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/q_pytest.log
for v in "A=1" "PTX_STREAMS=1"; do
  env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err || tail -5 gpurun_out/q_bench.err
  python - "$v" <<'PY'
import json,sys; d=json.loads(open('gpurun_out/q_bench.json').read().strip().splitlines()[-1]); print(sys.argv[1], '%.1f Msamples/s %.2f ms'%(d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})
PY
done
