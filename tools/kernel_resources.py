#!/usr/bin/env python3
"""Register / scratch usage of every kernel in kernels.hip as hipcc reports it (-Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py [extra hipcc flags]  -> table: VGPRs, VGPR spills, scratch B/lane, SGPR spills, occupancy, LDS"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "path_tracer_ocaml_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fno-math-errno", "--cuda-device-only", "-c", "kernels.hip", "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    else:
        k, _, v = t.partition(":")
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%5s %6s %7s %6s %3s %6s  kernel" % ("VGPR", "vspill", "scratch", "sspill", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    print("%5s %6s %7s %6s %3s %6s  %s" % (r.get("VGPRs", "?"), r.get("VGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                          r.get("SGPRs Spill", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"), n))
