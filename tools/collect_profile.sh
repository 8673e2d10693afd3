#!/bin/bash
# usage (on the GPU box): tools/collect_profile.sh <workload> <outdir>
# Separate rocprofv3 passes of the same bench command (MI355X_MICROARCH.md: never combine --pmc with trace domains;
# FETCH_SIZE and WRITE_SIZE in their own passes; 8 SQ counters per pass):
#   ktrace   kernel trace + stats, default two-stream schedule        -> the frame as the driver times it
#   ktrace1  kernel trace + stats, PTX_STREAMS=1                      -> per-kernel durations that add up to the step
#   sq_a     SQ issue counters (VALU busy, lane utilisation, LDS instructions)
#   sq_b     SQ wait / LDS bank-conflict counters
#   fetch / write   HBM traffic (FETCH_SIZE, WRITE_SIZE)
#   rdreq           the read requests behind FETCH_SIZE by size (32 / 64 / 128 B): FETCH_SIZE tallies every request at 64 B on
#                   gfx950 (TCC_BUBBLE reads 0), the sized counters give the bytes (profiles/r04_fetch_calibration.json)
# The program goes directly after `--` (no env/bash wrapper: the profiler's preload has already initialised the GPU).
set -e
export TMPDIR=/tmp
wl=$1; out=$2
mkdir -p $out
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --workload $wl"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -- $cmd > $out/bench_ktrace.log 2>&1
export PTX_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace1 -- $cmd > $out/bench_ktrace1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
  --output-format csv -d $out/sq_a -- $cmd > $out/bench_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU \
  --output-format csv -d $out/sq_b -- $cmd > $out/bench_sq_b.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE \
  --output-format csv -d $out/sq_c -- $cmd > $out/bench_sq_c.log 2>&1 || echo "sq_c pass failed (counter names differ on this ROCm?)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $cmd > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $cmd > $out/bench_write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/pmc_rdreq -- $cmd > $out/bench_rdreq.log 2>&1 || echo "rdreq pass failed"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc_l2 -- $cmd > $out/bench_l2.log 2>&1 || echo "l2 pass failed"
unset PTX_STREAMS
# keep what travels back small: per-dispatch kernel traces are not needed, the stats and counter tables are
find $out -name '*kernel_trace.csv' -delete
grep '^{' $out/bench_ktrace.log | tail -1
