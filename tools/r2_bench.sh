#!/bin/bash
mkdir -p gpurun_out
python bench.py > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err; echo "bench exit $?"; tail -3 gpurun_out/r2_bench_n1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_n1.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline','kernel_ms_per_step','host_api','cpu_baseline','parity')}, indent=1)[:4000])
PY
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo > gpurun_out/r2_bench_gloo2.json 2> gpurun_out/r2_bench_gloo2.err; echo "gloo2 exit $?"; tail -2 gpurun_out/r2_bench_gloo2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_gloo2.json').read().strip().splitlines()[-1])
print('gloo 2 ranks on one GPU:', d['value'], d['ms_per_step'], d['config']['sharding'])
PY
