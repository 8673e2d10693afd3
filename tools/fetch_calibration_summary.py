#!/usr/bin/env python3
"""Joins build/fetch_calib's own record (bytes requested, duration per access shape) with the rocprofv3 counter CSVs of the
same runs -> profiles/<tag>_fetch_calibration.json: for every shape, what FETCH_SIZE (KB, as rocprofv3 reports it) is worth in
requested bytes, i.e. the factor tools/summarize_sq.py has to apply instead of the blanket x2 of wide streaming reads.
usage: tools/fetch_calibration_summary.py <outdir> [tag]"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def counters(sub):
    files = sorted(glob.glob(os.path.join(out, sub, "*", "*counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    if not files:
        return agg, n
    for r in csv.DictReader(open(files[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    return agg, n


prog = json.load(open(os.path.join(out, "fetch_calib.json")))
fetch, nf = counters("pmc_fetch")
rd, nr = counters("pmc_rdreq")
tcc, nt = counters("pmc_tcc")
sz, ns = counters("pmc_sizes")
rows = []
for r in prog["rows"]:
    k = r["kernel"]
    e = dict(r)
    if k in fetch and "FETCH_SIZE" in fetch[k]:
        per = fetch[k]["FETCH_SIZE"] * 1024.0 / nf[k]["FETCH_SIZE"]  # bytes per dispatch as the counter reports them
        e["fetch_size_bytes"] = per
        e["fetch_size_per_requested"] = per / r["requested_bytes"]
        e["requested_per_fetch_size"] = r["requested_bytes"] / per if per else None
        e["fetch_size_per_record"] = per / r["records"]
    if k in rd:
        for c in rd[k]:
            e[c + "_per_record"] = rd[k][c] / nr[k][c] / r["records"]
    if k in sz:
        for c in sz[k]:
            e[c + "_per_record"] = sz[k][c] / ns[k][c] / r["records"]
        if "TCC_EA0_RDREQ_128B_sum" in sz[k] and "TCC_EA0_RDREQ_64B_sum" in sz[k]:
            b128 = sz[k]["TCC_EA0_RDREQ_128B_sum"] / ns[k]["TCC_EA0_RDREQ_128B_sum"]
            b64 = sz[k]["TCC_EA0_RDREQ_64B_sum"] / ns[k]["TCC_EA0_RDREQ_64B_sum"]
            e["sized_request_bytes"] = 128.0 * b128 + 64.0 * b64  # + 32 B requests: none on any shape here
            e["sized_request_bytes_per_requested"] = e["sized_request_bytes"] / r["requested_bytes"]
    if k in tcc:
        for c in tcc[k]:
            e[c + "_per_record"] = tcc[k][c] / nt[k][c] / r["records"]
    rows.append(e)
res = {"source": out, "buffer_bytes": prog["buffer_bytes"],
       "note": "fetch_size_per_requested = FETCH_SIZE (rocprofv3, KB x 1024) / bytes the kernel asked for; 0.5 is the guide's wide-streaming rule",
       "rows": rows}
path = os.path.join(root, "profiles", f"{tag}_fetch_calibration.json")
json.dump(res, open(path, "w"), indent=1)
for e in rows:
    print("%-22s %-7s rec %3d: requested %7.1f MB  %7.1f GB/s  FETCH_SIZE/requested %s  sized/requested %s  per record %s B  %s" % (
        e["kernel"], e["pattern"], e["record_bytes"], e["requested_bytes"] * 1e-6, e["requested_gbs"],
        "%.3f" % e["fetch_size_per_requested"] if "fetch_size_per_requested" in e else "-",
        "%.3f" % e["sized_request_bytes_per_requested"] if "sized_request_bytes_per_requested" in e else "-",
        "%.1f" % e["fetch_size_per_record"] if "fetch_size_per_record" in e else "-",
        {c: round(v, 3) for c, v in e.items() if c.endswith("_per_record") and c != "fetch_size_per_record"}))
print("->", path)
