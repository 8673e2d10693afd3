#!/bin/bash
# usage: tools/kstats.sh <outdir> [env assignments...]   per-kernel durations of one-stream bench steps
export TMPDIR=/tmp
out=$1; shift
for kv in "$@"; do export "$kv"; done
export PTX_STREAMS=${PTX_STREAMS:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads ${BENCH_ARGS} > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('%-70s calls %5s avg %9.1f us total %8.2f ms  %4.1f%%'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
