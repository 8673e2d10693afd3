#!/bin/bash
# usage: tools/share_timeline.sh <outdir> [world=8]  -> the launch-by-launch timeline of the LAST of 6 renders of rank 0's share of the
# headline frame (two-stream default schedule), from rocprofv3 --kernel-trace of tools/one_band_share.py
export TMPDIR=/tmp
out=$1; world=${2:-8}
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/one_band_share.py $world 6 > $out.log 2>&1
tail -3 $out.log
python3 - "$out" <<'PY'
import csv, glob, sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# split into renders at gaps > 300 us
frames=[]; cur=[]; last_end=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if last_end is not None and s-last_end>300000: frames.append(cur); cur=[]
    cur.append(r); last_end=max(last_end or 0,e)
frames.append(cur)
fr=frames[-1]
t0=int(fr[0]['Start_Timestamp'])
for r in fr:
    name=r['Kernel_Name'].split('(')[0].replace('void ','')
    print('%9.1f us  +%8.1f  q%s %s'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, (int(r['Start_Timestamp'])-t0)/1e3, r.get('Queue_Id','?'), name[:60]))
print('render span %.3f ms, %d launches'%((max(int(r['End_Timestamp']) for r in fr)-t0)/1e6, len(fr)))
PY
