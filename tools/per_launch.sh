#!/bin/bash
# usage: tools/per_launch.sh <outdir> [env...]  -> the last frame's trace / shade launches in order with durations (one stream)
export TMPDIR=/tmp
out=$1; shift
for kv in "$@"; do export "$kv"; done
export PTX_STREAMS=${PTX_STREAMS:-1}
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-workloads ${BENCH_ARGS} > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith(('void k_trace','void k_shade','void k_bounce','void k_classify','k_accum'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the timed frame = the 2nd of 3 renders (warmup, timed, count): pick launches by splitting on k_accum groups
frames=[]; cur=[]
n_acc=0
for r in rows:
    cur.append(r)
    if r['Kernel_Name'].startswith('k_accum'):
        n_acc+=1
        if n_acc%2==0: frames.append(cur); cur=[]
fr=frames[1] if len(frames)>1 else frames[0]
t0=int(fr[0]['Start_Timestamp'])
for r in fr:
    name=r['Kernel_Name'].split('(')[0].replace('void ','')
    print('%9.1f us  +%8.1f  %s'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, (int(r['Start_Timestamp'])-t0)/1e3, name))
print('frame span %.2f ms'%((int(fr[-1]['End_Timestamp'])-t0)/1e6))
PY
