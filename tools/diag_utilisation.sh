#!/bin/bash
# usage (GPU box): tools/diag_utilisation.sh <workload>   -- needs build_variants/libptx_diag{1,2,3,4}.so
# (make -C path_tracer_ocaml_amd/csrc -B OUT=../../build_variants/libptx_diagN.so EXTRA=-DPT_DIAG=N)
wl=${1:-shirley_1080p_spp64_d8}
for v in ${DIAGS:-1 2 3 4}; do
  PTX_LIB=$PWD/build_variants/libptx_diag$v.so timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-workloads --workload $wl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['work']
names={1:'node walk',2:'node walk, tail loss only',3:'packet scan',4:'packet heavy'}
print('PT_DIAG=$v %-28s lane steps %.4g  lane slots %.4g  utilisation %.3f' % (names[$v], d['nodes_tested'], d['floor_tested'], d['nodes_tested']/max(d['floor_tested'],1)))"
done
