#!/bin/bash
# usage: tools/pmc_variant.sh <variant> <outdir> "<counters>" [workload]   (one-stream schedule; per-kernel sums)
export TMPDIR=/tmp
export PTX_STREAMS=1
export PTX_LIB=$PWD/build_variants/libptx_$1.so
mkdir -p "$(dirname "$2")"
rocprofv3 --pmc $3 --output-format csv -d $2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-workloads ${4:+--workload $4} > $2.log 2>&1
python3 - "$2" <<'PY'
import csv, collections, glob, sys
f=sorted(glob.glob(sys.argv[1]+'/*/*counter_collection.csv'))[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    if 'k_trace' not in k and 'k_shade' not in k and 'k_bounce' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,c in sorted(agg.items()):
    print(k, {n:'%.4g'%v for n,v in sorted(c.items())})
PY
