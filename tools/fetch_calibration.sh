#!/bin/bash
# usage: tools/fetch_calibration.sh <outdir> [tag]   -> profiles/<tag>_fetch_calibration.json  (run on an MI355X box)
# Three separate counter passes of build/fetch_calib (the program goes directly after `--`): FETCH_SIZE; the raw request
# counters it derives from; WRITE_SIZE as a control (the program writes nothing).
set -e
out=$1; tag=${2:-r04}
export TMPDIR=/tmp
mkdir -p $out
make -C "$(dirname "$0")/fetch_calib" > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- build/fetch_calib $out/fetch_calib.json > $out/run_fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/pmc_rdreq -- build/fetch_calib $out/fetch_calib_b.json > $out/run_rdreq.log 2>&1 || echo "TCC_EA0_RDREQ pass failed (counter names not exposed?)"
rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_READ_SECTORS_sum --output-format csv -d $out/pmc_sizes -- build/fetch_calib $out/fetch_calib_d.json > $out/run_sizes.log 2>&1 || echo "TCC_EA0_RDREQ_128B pass failed"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- build/fetch_calib $out/fetch_calib_c.json > $out/run_tcc.log 2>&1 || echo "TCC_REQ pass failed"
python3 "$(dirname "$0")/fetch_calibration_summary.py" $out $tag
