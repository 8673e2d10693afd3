#!/usr/bin/env python3
"""Per-kernel summary of the SQ / HBM counter passes of tools/collect_profile.sh -> profiles/<tag>_sq.json.
usage: tools/summarize_sq.py gpurun_out/r02a_shirley r02 [workload]

Derived figures (MI355X_MICROARCH.md, rocprofv3 PMC section; SQ_*_CYCLES and SQ_ACTIVE_INST_* count quad-cycles):
  lane_util      = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)   active lanes per issued VALU instruction
  valu_busy      = SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CYCLES_per_SIMD)   share of time the vector pipes issue (per SIMD)
  valu_of_wave   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES                 share of a resident wave's life spent issuing VALU
  lds_of_wave    = SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES
  bank_conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE             LDS-array cycles that are conflict replays
  wait_any / wait_inst_any = SQ_WAIT_* / SQ_WAVE_CYCLES                 parked on s_waitcnt / stalled at issue
  hbm_bytes      = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; the gfx950 x2 correction for wide coalesced reads)
"""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "shirley_1080p_spp64_d8"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    c = glob.glob(pattern)
    return max(c, key=os.path.getmtime) if c else None


def short(name):
    k = name.split("(")[0].replace("void ", "").strip()
    return k


agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.Counter())
for sub in ("sq_a", "sq_b", "sq_c", "pmc_fetch", "pmc_write"):
    f = newest(os.path.join(src, sub, "*", "*counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1

dur = {}
f = newest(os.path.join(src, "ktrace1", "*", "*kernel_stats.csv"))
if f:
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}

out = {"workload": workload, "source": src, "note": "bench.py --steps 2 --warmup 1 under rocprofv3, PTX_STREAMS=1 for the counter and ktrace1 passes", "kernels": {}}
for k, c in sorted(agg.items()):
    n = max(launches[k].values())
    e = {"launches": n, "counters_sum": {m: v for m, v in sorted(c.items())}}
    g = c.get
    if g("SQ_ACTIVE_INST_VALU"):
        e["lane_util"] = g("SQ_THREAD_CYCLES_VALU", 0) / (g("SQ_ACTIVE_INST_VALU") * 64)
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        # sq_a and sq_b both carry SQ_WAVE_CYCLES (summed twice): normalise by the number of passes that had it
        passes = launches[k]["SQ_WAVE_CYCLES"] / max(launches[k].get("SQ_ACTIVE_INST_VALU", n), 1)
        wc1 = wc / max(passes, 1)
        e["valu_of_wave"] = g("SQ_ACTIVE_INST_VALU", 0) / wc1
        e["lds_of_wave"] = g("SQ_ACTIVE_INST_LDS", 0) / wc1
        e["wait_any"] = g("SQ_WAIT_ANY", 0) / wc1
        e["wait_inst_any"] = g("SQ_WAIT_INST_ANY", 0) / wc1
        e["active_inst_any"] = g("SQ_ACTIVE_INST_ANY", 0) / wc1
    if g("SQ_BUSY_CYCLES"):
        e["valu_busy"] = g("SQ_ACTIVE_INST_VALU", 0) * 4 / g("SQ_BUSY_CYCLES") if g("SQ_BUSY_CYCLES") else None
    if g("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        e["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = (2.0 * g("FETCH_SIZE", 0) + g("WRITE_SIZE", 0)) * 1024.0 / n
    if k in dur:
        e["one_stream"] = dur[k]
        if "hbm_bytes_per_launch" in e:
            e["hbm_gbs"] = e["hbm_bytes_per_launch"] / (dur[k]["avg_us"] * 1e-6) * 1e-9
            e["hbm_frac_of_8tbs"] = e["hbm_gbs"] / 8000.0
    out["kernels"][k] = e
dst = os.path.join(root, "profiles", f"{tag}_sq.json")
json.dump(out, open(dst, "w"), indent=1)
for k, e in out["kernels"].items():
    print(k)
    print("   ", {m: (round(v, 4) if isinstance(v, float) else v) for m, v in e.items() if m not in ("counters_sum",)})
print("wrote", dst)
