#!/bin/bash
# usage (on the GPU box): tools/collect_profile.sh <workload> <outdir>
# Three separate rocprofv3 passes of the same bench command: kernel trace + stats, then one PMC pass per HBM counter
# (MI355X_MICROARCH.md: never combine --pmc with trace domains; FETCH_SIZE and WRITE_SIZE in their own passes).
set -e
export TMPDIR=/tmp
wl=$1; out=$2
mkdir -p $out
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -- $cmd > $out/bench_ktrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $cmd > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $cmd > $out/bench_write.log 2>&1
# keep what travels back small: the per-dispatch traces are not needed, the stats and counter tables are
find $out -name '*kernel_trace.csv' -delete
grep '^{' $out/bench_ktrace.log | tail -1
