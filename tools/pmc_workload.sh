#!/bin/bash
# usage: tools/pmc_workload.sh <workload> <outdir> "<counters>"   (one rocprofv3 --pmc pass; prints per-kernel sums)
export TMPDIR=/tmp
rocprofv3 --pmc $3 --output-format csv -d $2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-workloads --workload $1 > $2.log 2>&1
python3 - "$2" <<'PY'
import csv, collections, glob, sys
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if 'k_trace' not in k and 'k_shade' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    n[k]+=1
for k,c in sorted(agg.items()):
    print(k, n[k]//max(len(c),1), 'launches', {m:'%.4g'%v for m,v in sorted(c.items())})
    if c.get('SQ_ACTIVE_INST_VALU') and c.get('SQ_WAVE_CYCLES'): print('    lane util %.3f  valu/wavecycles %.3f wait_inst_any %.3f wait_any %.3f'%(c.get('SQ_THREAD_CYCLES_VALU',0)/(c['SQ_ACTIVE_INST_VALU']*64), c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVE_CYCLES'], c.get('SQ_WAIT_INST_ANY',0)/c['SQ_WAVE_CYCLES'], c.get('SQ_WAIT_ANY',0)/c['SQ_WAVE_CYCLES']))
PY
