#!/bin/bash
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1 || true
bash tools/collect_profile.sh shirley_1080p_spp64_d8 gpurun_out/r02a_shirley
ls gpurun_out/r02a_shirley
