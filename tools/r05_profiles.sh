#!/bin/bash
# usage (on an MI355X box): tools/r05_profiles.sh  -> the rocprofv3 passes behind profiles/r05_* for the three single-GPU frames, the
# band-share timings and the unfused / fused A/B of the mesh walked from HBM / L2 (gpurun_out/r5p/...; tools/summarize_sq.py turns
# each directory into profiles/<tag>_sq.json + kernel-stats CSVs here or there)
mkdir -p gpurun_out/r5p
for wl in shirley_1080p_spp64_d8 cornell_1024_spp256_d16 ganesha_1080p_spp64_d8; do
  bash tools/collect_profile.sh $wl gpurun_out/r5p/$wl > gpurun_out/r5p/$wl.collect.log 2>&1
  echo "collected $wl: $(tail -c 300 gpurun_out/r5p/$wl.collect.log | head -c 200)"
done
python tools/summarize_sq.py gpurun_out/r5p/shirley_1080p_spp64_d8 r05 shirley_1080p_spp64_d8 > gpurun_out/r5p/sum_shirley.txt 2>&1
python tools/summarize_sq.py gpurun_out/r5p/cornell_1024_spp256_d16 r05_cornell cornell_1024_spp256_d16 > gpurun_out/r5p/sum_cornell.txt 2>&1
python tools/summarize_sq.py gpurun_out/r5p/ganesha_1080p_spp64_d8 r05_ganesha ganesha_1080p_spp64_d8 > gpurun_out/r5p/sum_ganesha.txt 2>&1
mkdir -p gpurun_out/r5p/profiles && cp profiles/r05* profiles/roofline_inputs.json gpurun_out/r5p/profiles/ 2>/dev/null
python tools/band_share_timing.py r05 2>/dev/null | tail -12; cp profiles/r05_band_share.json gpurun_out/r5p/profiles/ 2>/dev/null
python tools/share_step_rate.py 8 60 2>/dev/null | tail -2
bash tools/bench_env.sh ganesha_1080p_spp64_d8 "PTX_FUSED_GLOBAL=0" "PTX_FUSED_GLOBAL=1" 2>&1 | tee gpurun_out/r5p/ganesha_fused_ab.txt
