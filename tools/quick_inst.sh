#!/bin/bash
# usage: tools/quick_inst.sh "<explicit instantiation line(s)>" [extra hipcc flags]
#   -> VGPRs / spills / scratch / occupancy / LDS and instruction counts of exactly those kernels, compiled alone in seconds
#   (kernels.hip with PT_KERNELS_ONLY); the assembly stays in ${TMPDIR:-/tmp}/quick_inst.s
# e.g. tools/quick_inst.sh 'template __global__ void k_trace<1, false, false, false, true>(PtSceneDev, PtQueue, PtHits, int, PtCounters*, PtGenParams, const double*, uint32_t, uint32_t*, uint4*, int);' -DPT_TRACE_GLOBAL_WAVES=6
cd "$(dirname "$0")/../path_tracer_ocaml_amd/csrc"
T=${TMPDIR:-/tmp}/quick_inst.hip
inst=$1; shift
cat > $T <<EOT
#define PT_KERNELS_ONLY 1
#include "$PWD/kernels.hip"
$inst
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-math-errno --cuda-device-only -S -o ${T%.hip}.s $T \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re,subprocess
rows=[];cur={}
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass-analysis',line)
    if not m:
        if 'error' in line or 'warning' in line: print(line.rstrip())
        continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur: rows.append(cur)
        cur={'name':t.split(':',1)[1].strip()}
    else:
        k,_,v=t.partition(':'); cur[k.strip()]=v.strip()
if cur: rows.append(cur)
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    n=re.sub(r'\(.*$','',n).replace('void ','')
    if not re.match(r'k_(trace|bounce|shade)', n): continue
    print('%4s vgpr %3s vspill %4s scratch %3s sspill occ %s lds %6s  %s'%(r.get('VGPRs'),r.get('VGPRs Spill'),r.get('ScratchSize [bytes/lane]'),r.get('SGPRs Spill'),r.get('Occupancy [waves/SIMD]'),r.get('LDS Size [bytes/block]'),n))
"
python3 ../../tools/isa_count.py ${T%.hip}.s 'k_(trace|bounce|shade)' | c++filt | sed 's/(PtSceneDev.*)//' | cut -c1-160
