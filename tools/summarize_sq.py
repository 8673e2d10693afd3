#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 passes of tools/collect_profile.sh -> profiles/<tag>_sq.json,
profiles/<tag>_kernel_stats.csv (two-stream frame), profiles/<tag>_kernel_stats_one_stream.csv, and the entry of
profiles/roofline_inputs.json that bench.py reads for its `roofline.issue` / `roofline.hbm` blocks.
usage: tools/summarize_sq.py gpurun_out/r02_shirley r02 shirley_1080p_spp64_d8

Derived figures (MI355X_MICROARCH.md, "rocprofv3 PMC slots": SQ_*_CYCLES and SQ_ACTIVE_INST_* count quad-cycles;
SQ_BUSY_CYCLES counts cycles once per shader engine, 32 of them):
  cycles          = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs by the profiler) of the kernel's dispatches (sq_c pass, scaled to the
                    sq_a pass by SQ_WAVE_CYCLES): the
                    kernel's duration in shader cycles, summed over launches.  SQ_BUSY_CYCLES / 32 (one count per shader
                    engine) is kept as cycles_sq: it UNDER-counts the duration by >= 10 % on kernels that keep every engine
                    busy (round 2 got valu_busy 1.115 from it and clamped); no figure below is clamped any more
  valu_busy       = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles)      share of SIMD time the vector pipe issues
  valu_issue_from_insts = 4 SQ_INSTS_VALU / (1024 SIMDs x cycles)      the same from the instruction COUNT (4 cycles per wave64 instruction)
  lane_util       = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)   active lanes per issued vector instruction
  useful_issue_frac = valu_busy x lane_util                            useful lane-issue slots / all lane-issue slots
  lds_busy        = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)   (LDS-array cycles; scaled between passes by SQ_WAVE_CYCLES)
  lds_bank_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_any / wait_inst_any / active_inst_any = SQ_* / SQ_WAVE_CYCLES   parked on s_waitcnt or a barrier / stalled at issue / issuing
  wave_residency  = 4 SQ_WAVE_CYCLES / (SQ_WAVES x cycles)             share of the kernel's duration an average wave is alive
  hbm_bytes_per_launch = read bytes + WRITE_SIZE KB.  Read bytes: 128 n128 + 64 n64 + 32 n32 from the sized request counters
                    (TCC_EA0_RDREQ_{128B,64B,32B}) when the rdreq pass exists -- `"calibrated": true` -- else 2 FETCH_SIZE KB.
                    profiles/r04_fetch_calibration.json (tools/fetch_calib): on gfx950 every read request of every access shape
                    tried -- streams, isolated gathers of 16 ... 80-byte records, pooled gathers -- is a 128-byte one, FETCH_SIZE
                    tallies it at 64 B (its 128-B term, TCC_BUBBLE, reads 0), so 2 x FETCH_SIZE is the fabric read traffic
                    EXACTLY, not a ceiling; what differs per shape is the over-fetch (a 32-byte record gathered alone costs a line).
                    These are requests of the L2 to the fabric: Infinity-Cache hits are included (MI355X_MICROARCH.md).
  l2_read_hit_rate = TCC_HIT / TCC_REQ of the kernel (l2 pass)
"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "shirley_1080p_spp64_d8"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def newest(pattern):
    c = glob.glob(pattern)
    return max(c, key=os.path.getmtime) if c else None


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


passes = {}
for sub in ("sq_a", "sq_b", "sq_c", "pmc_fetch", "pmc_write", "pmc_rdreq", "pmc_l2"):
    f = newest(os.path.join(src, sub, "*", "*counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
    passes[sub] = (agg, n)

dur = {}
for sub, dst in (("ktrace1", f"{tag}_kernel_stats_one_stream.csv"), ("ktrace", f"{tag}_kernel_stats.csv")):
    f = newest(os.path.join(src, sub, "*", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(prof, dst))
        if sub == "ktrace1":
            for r in csv.DictReader(open(f)):
                dur[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
line = [l for l in open(os.path.join(src, "bench_ktrace.log")) if l.startswith("{")] if os.path.exists(os.path.join(src, "bench_ktrace.log")) else []
if line:
    open(os.path.join(prof, f"{tag}_bench_under_rocprof.json"), "w").write(line[-1])


def val(sub, k, name):
    if sub not in passes:
        return None, 0
    agg, n = passes[sub]
    return (agg[k].get(name), n[k].get(name, 0)) if k in agg else (None, 0)


kernels = sorted({k for agg, _ in passes.values() for k in agg})
out = {"workload": workload, "source": src,
       "note": "bench.py --steps 2 --warmup 1 under rocprofv3; PTX_STREAMS=1 for the counter passes and ktrace1", "kernels": {}}
for k in kernels:
    e = {}
    g = lambda name, sub: val(sub, k, name)[0]
    busy, nl = val("sq_a", k, "SQ_BUSY_CYCLES")
    e["launches"] = nl
    if busy:
        cycles_sq = busy / 32.0
        av, tc, waves, wca = g("SQ_ACTIVE_INST_VALU", "sq_a"), g("SQ_THREAD_CYCLES_VALU", "sq_a"), g("SQ_WAVES", "sq_a"), g("SQ_WAVE_CYCLES", "sq_a")
        grbm, wcc = g("GRBM_GUI_ACTIVE", "sq_c"), g("SQ_WAVE_CYCLES", "sq_c")
        # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs of the MI355X (each XCD's GRBM counts its own busy cycles): / 8
        cycles = grbm / 8.0 * (wca / wcc) if (grbm and wcc) else cycles_sq
        e["duration_cycles_source"] = "GRBM_GUI_ACTIVE" if (grbm and wcc) else "SQ_BUSY_CYCLES/32 (sq_c pass missing)"
        e["cycles_per_launch"] = cycles / nl
        e["cycles_sq_per_launch"] = cycles_sq / nl
        e["valu_busy_sq_busy_cycles"] = 4.0 * av / (1024.0 * cycles_sq)  # round 2's normalisation, for comparison
        e["valu_busy"] = 4.0 * av / (1024.0 * cycles)
        e["valu_issue_from_insts"] = 4.0 * g("SQ_INSTS_VALU", "sq_a") / (1024.0 * cycles)
        e["lane_util"] = tc / (64.0 * av)
        e["useful_issue_frac"] = e["valu_busy"] * e["lane_util"]
        e["valu_insts_per_launch"] = g("SQ_INSTS_VALU", "sq_a") / nl
        e["lds_insts_per_launch"] = g("SQ_INSTS_LDS", "sq_a") / nl
        e["wave_residency"] = 4.0 * wca / (waves * cycles / nl) if waves else None
        e["waves_per_launch"] = waves / nl
        wcb = g("SQ_WAVE_CYCLES", "sq_b")
        if wcb:
            scale = wca / wcb
            e["lds_busy"] = g("SQ_LDS_IDX_ACTIVE", "sq_b") * scale / (256.0 * cycles)
            idx = g("SQ_LDS_IDX_ACTIVE", "sq_b")
            e["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT", "sq_b") / idx if idx else 0.0
            e["wait_any"] = g("SQ_WAIT_ANY", "sq_b") / wcb
            e["wait_inst_any"] = g("SQ_WAIT_INST_ANY", "sq_b") / wcb
            e["active_inst_any"] = g("SQ_ACTIVE_INST_ANY", "sq_b") / wcb
            e["salu_insts_per_launch"] = g("SQ_INSTS_SALU", "sq_b") / val("sq_b", k, "SQ_INSTS_SALU")[1]
    f, nf = val("pmc_fetch", k, "FETCH_SIZE")
    w, nw = val("pmc_write", k, "WRITE_SIZE")
    if f is not None and w is not None and nf and nw:
        e["hbm_fetch_bytes_per_launch_x2_corrected"] = 2.0 * f * 1024.0 / nf
        e["hbm_write_bytes_per_launch"] = w * 1024.0 / nw
        read_bytes = e["hbm_fetch_bytes_per_launch_x2_corrected"]
        n128, c128 = val("pmc_rdreq", k, "TCC_EA0_RDREQ_128B_sum")
        n64, c64 = val("pmc_rdreq", k, "TCC_EA0_RDREQ_64B_sum")
        n32, c32 = val("pmc_rdreq", k, "TCC_EA0_RDREQ_32B_sum")
        nall, call = val("pmc_rdreq", k, "TCC_EA0_RDREQ_sum")
        if n128 is not None and c128:
            sized = (128.0 * n128 + 64.0 * (n64 or 0.0) + 32.0 * (n32 or 0.0)) / c128
            e["hbm_read_bytes_per_launch_sized_requests"] = sized
            e["read_requests_per_launch"] = {"all": (nall or 0.0) / max(call, 1), "128B": n128 / c128, "64B": (n64 or 0.0) / max(c64, 1), "32B": (n32 or 0.0) / max(c32, 1)}
            e["hbm_calibrated"] = True
            read_bytes = sized
        e["hbm_bytes_per_launch"] = read_bytes + e["hbm_write_bytes_per_launch"]
    req, creq = val("pmc_l2", k, "TCC_REQ_sum")
    hit, _ = val("pmc_l2", k, "TCC_HIT_sum")
    if req:
        e["l2_hit_rate"] = (hit or 0.0) / req
        e["l2_requests_per_launch"] = req / max(creq, 1)
    if k in dur:
        e["one_stream"] = dur[k]
        if "hbm_bytes_per_launch" in e:
            e["hbm_gbs"] = e["hbm_bytes_per_launch"] / (dur[k]["avg_us"] * 1e-6) * 1e-9
            e["hbm_frac_of_8tbs"] = e["hbm_gbs"] / 8000.0
    out["kernels"][k] = e
json.dump(out, open(os.path.join(prof, f"{tag}_sq.json"), "w"), indent=1)


# ---- the dominant kernel's entry for bench.py: k_bounce (one launch per bounce, LDS-resident scenes) where it ran, else the
# k_trace / k_trace_stream instantiations; COUNT = false (the counting renders are untimed)
def targs_of(k):
    return [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]


def is_timed_trace(k):
    if not k.startswith(("k_trace<", "k_trace_stream<", "k_bounce<")):
        return False
    return targs_of(k)[1] == "false"


def is_primary(k):
    if k.startswith("k_bounce<"):
        return targs_of(k)[3] == "true"
    if k.startswith("k_trace<"):
        return targs_of(k)[2] == "true"
    return False


tr = {k: e for k, e in out["kernels"].items() if is_timed_trace(k) and e.get("launches")}
if any(k.startswith("k_bounce<") for k in tr):
    tr = {k: e for k, e in tr.items() if k.startswith("k_bounce<")}
sec = {k: e for k, e in tr.items() if not is_primary(k)}
ri_path = os.path.join(prof, "roofline_inputs.json")
ri = json.load(open(ri_path)) if os.path.exists(ri_path) else {}
if tr:
    nl = sum(e["launches"] for e in tr.values())
    entry = {"source": f"profiles/{tag}_sq.json", "calibrated": all(e.get("hbm_calibrated", False) for e in tr.values()),
             "trace_hbm_bytes_per_launch": sum(e.get("hbm_bytes_per_launch", 0.0) * e["launches"] for e in tr.values()) / nl,
             "trace_avg_launch_us_one_stream": sum(e["one_stream"]["avg_us"] * e["one_stream"]["calls"] for e in tr.values() if "one_stream" in e)
                                               / max(sum(e["one_stream"]["calls"] for e in tr.values() if "one_stream" in e), 1)}
    dom = max(sec.values(), key=lambda e: e["launches"]) if sec else max(tr.values(), key=lambda e: e["launches"])
    for key in ("valu_busy", "valu_issue_from_insts", "valu_busy_sq_busy_cycles", "duration_cycles_source", "lane_util", "useful_issue_frac", "lds_busy",
                "lds_bank_conflict_share", "wait_any", "wait_inst_any", "wave_residency"):
        entry[key] = dom.get(key)
    entry["counters_of"] = [k for k, e in out["kernels"].items() if e is dom][0]
    # time-weighted share of SIMD time the vector pipe issues, per stage (timed instantiations only): what bench.py turns
    # into the frame's vector-issue time (sum over stages of one-stream kernel time x this share)
    for stage, pref in (("trace", ("k_trace<", "k_trace_stream<")), ("shade", ("k_shade<", "k_shade_pool<", "k_shade_cat<")), ("bounce", ("k_bounce<",))):
        ks = [e for k, e in out["kernels"].items() if k.startswith(pref) and "one_stream" in e and e.get("valu_busy") is not None
              and (stage == "shade" or is_timed_trace(k))]
        tot = sum(e["one_stream"]["total_ms"] for e in ks)
        if tot > 0:
            entry[f"valu_busy_{stage}_time_weighted"] = sum(e["valu_busy"] * e["one_stream"]["total_ms"] for e in ks) / tot
    # frame-level HBM traffic: every timed (non-COUNT) kernel's counter bytes x launches, per rendered frame.
    # Frames in the profiled command are counted by the walk kernels themselves: launches of the timed camera-ray (PRIMARY)
    # instantiations / batches per frame (bench line: the walk kernel's launches per step / max_bounces).  Until round 5 the
    # count was "k_accum launches / accumulate launches of one step" -- which stopped being a frame count when ptx_render began
    # to issue a frame's last accumulate in four row slabs (5 launches instead of 2): 34 launches read as 17 frames where the
    # bounce kernels had rendered 10, and the bounce stage's bytes per step came out 1.7 x too low (hbm_frame 0.23 for 0.38).
    # k_accum and k_film run for every frame, counting render included: their totals are divided by ALL camera-ray launches.
    try:
        bl = json.loads(line[-1]) if line else {}
        lps = bl.get("kernel_launches_per_step", {})
        walk_per_step = lps.get("bounce") or lps.get("trace")
        depth = (bl.get("config") or {}).get("max_bounces")
        def walk(k):
            return k.startswith(("k_trace<", "k_trace_stream<", "k_bounce<"))
        prim_timed = sum(e["launches"] for k, e in out["kernels"].items() if walk(k) and is_primary(k) and is_timed_trace(k) and e.get("launches"))
        prim_all = sum(e["launches"] for k, e in out["kernels"].items() if walk(k) and is_primary(k) and e.get("launches"))
        if walk_per_step and depth and prim_timed:
            batches = walk_per_step / depth
            frames = prim_timed / batches
            frames_all = prim_all / batches
            def timed(k):
                if walk(k):
                    return is_timed_trace(k)
                return k.startswith(("k_shade", "k_classify"))
            def every_frame(k):
                return k.startswith(("k_accum", "k_film"))
            def per_step(k, e):
                b = e.get("hbm_bytes_per_launch", 0.0) * e["launches"]
                return b / frames if timed(k) else (b / frames_all if every_frame(k) else 0.0)
            entry["frames_profiled"] = frames
            entry["frames_profiled_incl_counting"] = frames_all
            # the same count a second way -- the timed walk launches that are NOT camera-ray launches, (depth - 1) x batches per frame
            # in the one-launch-per-bounce schedule -- so that a change of the launch pattern cannot skew the bytes per step unnoticed
            # again (a solo run, PTX_SOLO_ENTRIES, returns early from later launches but still issues them)
            sec_timed = sum(e["launches"] for k, e in out["kernels"].items() if walk(k) and not is_primary(k) and is_timed_trace(k) and e.get("launches"))
            if depth > 1 and sec_timed:
                frames_check = sec_timed / (batches * (depth - 1))
                entry["frames_profiled_check"] = frames_check
                if abs(frames_check - frames) > 0.01 * frames:
                    entry["frames_inconsistent"] = True
                    print(f"WARNING: frames of the profiled run: {frames} by camera-ray launches, {frames_check} by the other walk launches -- "
                          f"hbm_bytes_per_step is not to be trusted", file=sys.stderr)
            entry["frames_from"] = "camera-ray launches of the timed walk kernels / batches per frame"
            entry["hbm_bytes_per_step"] = sum(per_step(k, e) for k, e in out["kernels"].items() if e.get("launches"))
            entry["hbm_bytes_per_step_by_stage"] = {
                st: sum(per_step(k, e) for k, e in out["kernels"].items() if k.startswith(pref) and e.get("launches"))
                for st, pref in (("bounce", ("k_bounce",)), ("trace", ("k_trace",)), ("shade", ("k_shade", "k_classify")), ("accum", ("k_accum",)), ("film", ("k_film",)))}
    except Exception as ex:  # a profile without the bench line still summarises
        entry["hbm_bytes_per_step_error"] = str(ex)
    ri[workload] = entry
    json.dump(ri, open(ri_path, "w"), indent=1)
    print(json.dumps(entry, indent=1))
for k, e in out["kernels"].items():
    print(k, {m: (round(v, 4) if isinstance(v, float) else v) for m, v in e.items() if m != "one_stream"}, e.get("one_stream"))
