#!/bin/bash
# usage: tools/quick_kernel.sh [extra hipcc flags]  -> registers / spills / ISA counts of a few representative kernels in seconds
# (kernels.hip with PT_KERNELS_ONLY: the device code without the API layer, explicit instantiations below)
cd "$(dirname "$0")/../path_tracer_ocaml_amd/csrc"
T=${TMPDIR:-/tmp}/quick_kernel.hip
cat > $T <<EOT
#define PT_KERNELS_ONLY 1
#include "$PWD/kernels.hip"
template __global__ void k_shade_pool<false, false>(PtSceneDev, PtQueue, PtHits, PtQueue, PtContrib, const double*, int, int, PtGenParams, uint32_t, uint32_t*);
template __global__ void k_bounce<0, false, false, false, true>(PtSceneDev, PtQueue, PtHits, PtQueue, PtContrib, const double*, int, int, PtGenParams, uint32_t, int, uint32_t, PtCounters*, int, PtSolo);
template __global__ void k_bounce<1, false, true, false, true>(PtSceneDev, PtQueue, PtHits, PtQueue, PtContrib, const double*, int, int, PtGenParams, uint32_t, int, uint32_t, PtCounters*, int, PtSolo);
template __global__ void k_bounce<1, false, false, false, false>(PtSceneDev, PtQueue, PtHits, PtQueue, PtContrib, const double*, int, int, PtGenParams, uint32_t, int, uint32_t, PtCounters*, int, PtSolo);
${QK_EXTRA}
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-math-errno --cuda-device-only -S -o ${T%.hip}.s $T \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re,subprocess
rows=[];cur={}
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass-analysis',line)
    if not m:
        if 'error' in line: print(line.rstrip())
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
    print('%4s vgpr %3s vspill %4s scratch %3s sspill  %s'%(r.get('VGPRs'),r.get('VGPRs Spill'),r.get('ScratchSize [bytes/lane]'),r.get('SGPRs Spill'),n))
"
python3 ../../tools/isa_count.py ${T%.hip}.s | c++filt | sed 's/(PtSceneDev.*)//' | cut -c1-140
