#!/usr/bin/env python3
"""Turns a gpurun_out/<run>/ profile directory (rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE
passes of `bench.py`) into the small files kept under profiles/ and into profiles/traffic.json (read by bench.py
for roofline.traffic).  usage: tools/summarize_profile.py gpurun_out/r01c r01 shirley_1080p_spp64_d8"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun merges every call's files into the same local directory: take the latest run's file"""
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(os.path.join(src, "ktrace", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_ktrace.log")) if l.startswith("{")]
if line:
    open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(line[-1])
pmc = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = newest(os.path.join(src, sub, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"):
            continue
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    pmc[name] = {k: {"launches": n, "KB_per_launch": v / n, "KB_total": v} for k, (n, v) in sorted(agg.items())}
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_hbm.json"), "w"), indent=1)
# HBM bytes per k_trace launch of the timed (non-counting) instantiations.  FETCH_SIZE / WRITE_SIZE are in KB.
# MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read -> double it.
def per_launch(counter, scale):
    n = tot = 0
    for k, v in pmc[counter].items():
        targs = k[k.index("<") + 1:].split(",") if k.startswith(("k_trace<", "k_trace_stream<")) else []
        if len(targs) >= 2 and targs[1].strip() == "false":  # COUNT == false: the instantiations bench.py times
            n += v["launches"]
            tot += v["KB_total"]
    return tot / max(n, 1) * 1024.0 * scale
fetch = per_launch("FETCH_SIZE", 2.0)
write = per_launch("WRITE_SIZE", 1.0)
tpath = os.path.join(dst, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
traffic[workload] = {"trace_hbm_bytes_per_launch": fetch + write, "fetch_bytes_per_launch_x2_corrected": fetch,
                     "write_bytes_per_launch": write, "source": f"profiles/{tag}_pmc_hbm.json"}
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps(traffic[workload], indent=1))
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:1500])
