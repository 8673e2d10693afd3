#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s of the per-pixel sampling integrator on MI355X.

Workload (BASELINE.json configs[1]): shirley_spheres 1920x1080, spp=64, max bounces 8, fixed seeds.
A "step" = one complete render of that image: generate -> (trace, shade) x 8 -> accumulate for every
sample, gather of the raw sums to rank 0 (N > 1) and the film filter + gamma.  The scene (BVH, packets,
materials) is resident in HBM before the timed region; the framebuffer stays on the device.

N > 1: the SAME image, rows dealt to ranks in interleaved 8-row bands (strong scaling), one group of
point-to-point sends (RCCL over xGMI) of the raw sums into rank 0 per step; the film reads the bands in place.
Started as the driver does (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, WORLD_SIZE set) each
process is one rank; started plainly (`python bench.py --gpus N`, no WORLD_SIZE) it launches that command itself -- N fresh
child processes, before this process imports torch or touches a GPU -- and relays rank 0's line and the exit code
(launch_ranks).  `--rehearse-launch` runs the same launch + band exchange on the CPU over gloo (no render, `value` null).

Prints ONE JSON line on rank 0 (contract in the task statement) with "roofline" and "cpu_baseline".

Roofline.  The dominant kernel is k_trace on queued (bounce >= 1) rays.  On the headline scene the
whole tree lives in LDS, so the kernel is bound by vector-instruction issue, not by HBM: `bound` names that pipe,
`achieved` is the reference algorithm's floating-point work (27 flop per Bbox.is_hit, 24 per packet slot scanned, counted
by the kernels themselves and equal to the oracle's counts) over the kernel's launch time measured live on ONE stream (so
the per-kernel times add up to the step), `frac` <= 1 against the binary64 vector peak.  `hbm` carries the counter-measured
traffic of the same launches (profiles/, FETCH_SIZE x 2 + WRITE_SIZE) and `issue` the tracked SQ counters (VALU busy,
lane utilisation).  Scenes traversed from HBM/L2 (ganesha-like) report `bound: "hbm"` on algorithmic bytes instead.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, width, height, spp, max_bounces)
    "shirley_1080p_spp64_d8": ("shirley", 1920, 1080, 64, 8),       # BASELINE configs[1] (headline)
    "shirley_600x300_spp32_d8": ("shirley", 600, 300, 32, 8),       # configs[0]
    "cornell_1024_spp256_d16": ("cornell", 1024, 1024, 256, 16),    # configs[2]
    "ganesha_1080p_spp64_d8": ("ganesha", 1920, 1080, 64, 8),       # configs[3]
    "shirley_4k_spp256_d8": ("shirley", 3840, 2160, 256, 8),        # configs[4]
}

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def build_scene(H, name, w, h):
    if name == "shirley":
        return H.shirley_spheres(w, h)
    if name == "cornell":
        return H.cornell_box(w, h, float(os.environ.get("PTX_BENCH_CORNELL_EMIT", "12.0")))  # 0: no emitter -- same paths, no emission records (a bandwidth A/B)
    if name == "ganesha":
        return H.ganesha_like(w, h, 150000, 7)
    raise ValueError(name)


def algorithmic_bytes(stats, spp, triangles):
    """B = 24/spp + sum_segments [N_node*64 + N_prim*P + 192] per sample."""
    P = 84 if triangles else 32
    per_prim = (stats["prims_tested"] + stats["floor_tested"]) * P
    total = stats["samples"] * 24.0 / spp + stats["nodes_tested"] * 64.0 + per_prim + stats["segments"] * 192.0
    trace_only = stats["nodes_tested"] * 64.0 + per_prim + stats["segments"] * 64.0  # ray in 48 B + hit out 16 B
    return total, trace_only


FLOP_PER_NODE_TEST = 27.0   # Bbox.is_hit: 6 sub, 6 mul, 12 min/max, 2 clamps, 1 compare (bbox.ml:40-56)
FLOP_PER_SLOT_SCAN = 24.0   # spheres_intersect_aux up to the discriminant (lib.rs:115-160); triangles: Moller-Trumbore ~ the same
# k_bounce also shades what it traced: the reference's arithmetic per segment outside Scene.intersect, counted like the two above
# (add / mul / fma-as-2 / div / sqrt / compare = 1 each, a transcendental = 1): a surface hit ~135 (Ray.point_at 6, normal 12 + facing 5,
# Shader_space rotation ~20, two quaternion transforms 2 x 33, Material.scatter + Pdf sample ~15, new ray + attenuation ~14),
# a miss ~25 (unit direction + background lerp); 100 = the headline scene's 0.61 : 0.39 mix.  An approximation, stated as such.
FLOP_PER_SEGMENT_SHADE = 100.0
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X binary64 vector peak = half the 157.3 TFLOP/s binary32 figure (MI355X_MICROARCH.md)


def tracked_counters(workload):
    """SQ / HBM counter summary of the dominant kernel, from the tracked rocprofv3 passes (tools/collect_profile.sh ->
    tools/summarize_sq.py -> profiles/roofline_inputs.json).  None if this workload was never profiled."""
    path = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    try:
        return json.load(open(path)).get(workload)
    except Exception:
        return None


def measured_hbm_copy_gbs(torch, dev):
    """Measured device copy bandwidth (read + write bytes / time) of a 1 GiB f32 tensor: the 'measured HBM
    roofline' BASELINE.md asks for next to the 8 TB/s datasheet figure."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * a.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) * 1e-9


def effective_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU box shows 256 logical
    CPUs but grants 16 CPUs' worth of time; more runnable threads than that only add throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(workload, seconds_budget=20.0):
    """The oracle (C restatement of the reference CPU path, libm math like the OCaml runtime, tile-parallel
    over all host threads like integrator.ml:138-146) timed on a bounded sample of the same workload."""
    from oracle import oracle as O
    flags = O.use_native_build()  # -O3 -march=native on the box that runs it
    scene, w, h, spp, depth = WORKLOADS[workload]
    if scene == "shirley":
        d = O.desc_shirley(w, h)
    elif scene == "cornell":
        d = O.desc_cornell(w, h, 12.0)
    else:
        d = O.desc_ganesha_like(w, h, 150000, 7)
    s = O.Scene(d.ptr, d)
    cores = effective_cpus()
    O.set_math(1)
    try:
        # calibrate on 1 pass, then size the sample to the budget
        r = s.render(w, h, 1, depth, threads=cores)
        rate = w * h / max(r["ms"], 1e-3) * 1e3
        n_pass = int(max(1, min(spp, seconds_budget * rate / (w * h))))
        r = s.render(w, h, n_pass, depth, threads=cores)
    finally:
        O.set_math(0)
    samples = w * h * n_pass
    return {
        "value": samples / r["ms"] * 1e-3, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"{w}x{h} spp={n_pass} of {spp} depth={depth}, same scene/camera, {cores} threads, "
                  f"{r['ms'] / 1e3:.1f} s (C restatement of the reference OCaml/Rust path, libm math, gcc {flags} -ffp-contract=off)",
        "real_reference": real_reference(workload),
    }, r["rgb"], n_pass


def real_reference(workload):
    """when dune and cargo are on PATH and a checkout of the reference is at hand, also time the
    README command (`dune exec --release shirley_spheres ...`, /root/reference/README.md:7) and call it "real
    reference".  Neither toolchain exists in the build image or on the GPU box, so this normally reports why not."""
    import shutil
    import subprocess
    scene, w, h, spp, depth = WORKLOADS[workload]
    ref = os.environ.get("PT_REFERENCE_DIR", "/root/reference")
    missing = [t for t in ("dune", "cargo") if shutil.which(t) is None]
    if missing:
        return {"value": None, "why": "not on PATH: " + ", ".join(missing)}
    if scene != "shirley" or not os.path.isdir(ref):
        return {"value": None, "why": "no reference checkout" if scene == "shirley" else "the reference has no path-integrator binary for this scene"}
    cmd = ["dune", "exec", "--release", "shirley_spheres", "--", f"--dimension={w},{h}", f"--samples-per-pixel={spp}",
           f"--max-ray-bounces={depth}", "--no-progress", "-o", "/tmp/real_reference.png"]
    try:
        subprocess.run(["dune", "build", "--release"], cwd=ref, check=True, capture_output=True, timeout=1800)
        t0 = time.perf_counter()
        out = subprocess.run(cmd, cwd=ref, check=True, capture_output=True, text=True, timeout=3600).stdout
        wall = time.perf_counter() - t0
        ms = [float(l.split(":")[1].split()[0]) for l in out.splitlines() if l.startswith("rendered in:")]
        sec = ms[0] * 1e-3 if ms else wall
        return {"value": w * h * spp / sec * 1e-6, "unit": "Msamples/s", "kind": "reference", "seconds": sec, "command": " ".join(cmd)}
    except Exception as e:
        return {"value": None, "why": f"{type(e).__name__}: {e}"}


# The other single-GPU BASELINE configurations, timed by the driver's own command (`workloads` in the JSON line): name -> (workload,
# band_first, band_step).  Config 5 needs 8 GPUs as a whole; what one GPU can time of it is the busiest rank's share
# (rank 0 of the 8-rank band deal: 272 of 2160 rows, tools/band_share_timing.py).  At N > 1 the ranks time config 5 itself.
EXTRA_WORKLOADS = {
    "shirley_600x300_spp32_d8": ("shirley_600x300_spp32_d8", 0, 0),
    "cornell_1024_spp256_d16": ("cornell_1024_spp256_d16", 0, 0),
    "ganesha_1080p_spp64_d8": ("ganesha_1080p_spp64_d8", 0, 0),
    "shirley_4k_spp256_d8_share_1_of_8": ("shirley_4k_spp256_d8", 0, 8),
}
SCALING_WORKLOAD = "shirley_4k_spp256_d8"  # BASELINE.json configs[4]: what the >= 6x target at 8 GPUs is quoted on
METRIC = "Msamples/s (WxHxspp) + achieved HBM GB/s vs roofline; per-pixel Linf vs CPU ref"


class Mods:
    """The modules a rank needs.  In a launch rehearsal (no GPU) the product library is not loaded: there is no CPU renderer."""

    def __init__(self, rehearse):
        import torch
        import torch.distributed as dist
        from path_tracer_ocaml_amd import distributed as D
        self.torch, self.dist, self.D = torch, dist, D
        self.P = self.H = None
        if not rehearse:
            import path_tracer_ocaml_amd as P
            from path_tracer_ocaml_amd import host as H
            self.P, self.H = P, H


def oracle_desc(O, scene_name, w, h):
    return {"shirley": lambda: O.desc_shirley(w, h), "cornell": lambda: O.desc_cornell(w, h, 12.0),
            "ganesha": lambda: O.desc_ganesha_like(w, h, 150000, 7)}[scene_name]()


class Job:
    """One configuration on this rank of `world`: scene resident in HBM, every buffer of a step allocated once, steps QUEUED
    (PTX_RENDER_ASYNC render of this rank's interleaved bands -> one group of sends into rank 0 -> banded film on rank 0, the
    framebuffer left on the device).  band_step / band_first override the deal (a single GPU timing one rank's share of an
    N-rank job).  rehearse: the same buffers, exchange and bookkeeping on the CPU over gloo with a pattern in place of the render."""

    def __init__(self, m, workload, rank, world, dev, local_dev, backend, passes_per_batch=0, rehearse=False, band_first=None, band_step=None):
        self.m, self.workload, self.rank, self.world, self.dev, self.local_dev = m, workload, rank, world, dev, local_dev
        self.backend, self.rehearse, self.ppb = backend, rehearse, passes_per_batch
        self.scene_name, self.w, self.h, self.spp, self.depth = WORKLOADS[workload]
        self.band_first = rank if band_first is None else band_first
        self.band_step = world if band_step is None else band_step
        self.deal = max(self.band_step, 1)  # ranks of the band deal this job's layout belongs to
        D = m.D
        self.bg = D.BandGather(self.h, self.w, self.band_first, self.deal, dev) if self.deal != world or world == 1 else D.BandGather(self.h, self.w, rank, world, dev)
        self.rows = len(D.band_layout(self.h, self.deal)[self.band_first])
        self.n_frames = 0
        self.scene = self.params = self.rgb = None
        self.stream = None
        if not rehearse:
            P, H, torch = m.P, m.H, m.torch
            hs = build_scene(H, self.scene_name, self.w, self.h)
            self.scene = P.Scene(hs.ptr, local_dev, keepalive=hs)
            self.params = self._params(asynchronous=True)
            self.rgb = torch.zeros((self.h, self.w, 3), dtype=torch.float64, device=dev) if rank == 0 else None
            self.stream = torch.cuda.current_stream().cuda_stream

    def _params(self, **kw):
        return self.m.P.render_params(self.w, self.h, self.spp, self.depth, band_rows=self.m.D.BAND_ROWS, band_first=self.band_first,
                                      band_step=self.band_step, passes_per_batch=self.ppb, **kw)

    # -- the pieces of a step
    def render(self, params=None):
        if self.rehearse:  # a pattern that names (global row, frame): what rank 0 checks after the exchange
            import numpy as np
            rows = self.m.D.band_layout(self.h, self.deal)[self.band_first]
            self.bg.part[:len(rows)] = self.m.torch.from_numpy(rows.astype(np.float64) + 0.25 * self.n_frames)[:, None, None]
            return None
        return self.scene.render_raw_device(params or self.params, self.bg.part.data_ptr(), self.stream)

    def exchange(self):
        return self.bg.gather() if self.deal == self.world else self.bg.gathered

    def film(self, wait=False):
        if self.rank == 0 and not self.rehearse:
            self.m.P.film_resolve_banded_device(self.local_dev, self.w, self.h, self.spp, self.bg.gathered.data_ptr(), self.deal, self.m.D.BAND_ROWS,
                                                self.bg.pad_rows, self.rgb.data_ptr(), self.stream, wait=wait)

    def step(self):
        st = self.render()
        self.exchange()
        self.film()
        self.n_frames += 1
        return st

    def sync(self):
        if not self.rehearse:
            self.m.torch.cuda.synchronize()

    def fence(self):
        self.sync()
        if self.world > 1:
            self.m.dist.barrier()
            self.sync()

    def timed(self, steps, warmup):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks (seconds)."""
        for _ in range(warmup):
            self.step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        return self.max_over_ranks(time.perf_counter() - t0)

    # -- small collectives on control values
    def _cdev(self):
        return self.dev if (self.backend == "nccl" and not self.rehearse) else self.m.torch.device("cpu")

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        t = self.m.torch.tensor([x], dtype=self.m.torch.float64, device=self._cdev())
        self.m.dist.all_reduce(t, op=self.m.dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank"""
        if self.world == 1:
            return [float(x)]
        torch = self.m.torch
        mine = torch.tensor([x], dtype=torch.float64, device=self._cdev())
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.m.dist.all_gather(out, mine)
        return [float(o.item()) for o in out]

    def breakdown(self):
        """One more step taken apart, untimed: every rank's render of its own bands alone (waited for), then the exchange into
        rank 0, then rank 0's film -- where a scaling loss comes from, in one line."""
        self.fence()
        t0 = time.perf_counter()
        self.render()
        self.sync()
        render_ms = (time.perf_counter() - t0) * 1e3
        per_rank = self.per_rank(render_ms)
        self.fence()
        t1 = time.perf_counter()
        self.exchange()
        self.sync()
        gather_ms = (time.perf_counter() - t1) * 1e3
        t2 = time.perf_counter()
        self.film(wait=True)
        self.sync()
        film_ms = (time.perf_counter() - t2) * 1e3
        self.n_frames += 1
        self.fence()
        return {"render_ms_per_rank": per_rank, "rank0_gather_ms": gather_ms if self.world > 1 else 0.0,
                "rank0_film_ms": None if self.rehearse else film_ms}

    def frame_parity(self, pixels=64):
        """Rank 0, untimed, the oracle as the checker: `pixels` whole pixels of the LAST frame's raw sums -- read out of the
        gathered banded layout, i.e. what every rank rendered and the exchange delivered -- against the CPU oracle in libm math
        (the reference's), relative L-inf of the per-pixel mean radiance."""
        if self.rank != 0:
            return None
        if self.rehearse:
            return {"rel_linf_vs_cpu_ref": None, "tolerance": 1e-5, "pixels": pixels, "note": "launch rehearsal: nothing was rendered"}
        import numpy as np
        D = self.m.D
        try:
            from oracle import oracle as O
            od = oracle_desc(O, self.scene_name, self.w, self.h)
            rng = np.random.default_rng(20)
            own = D.band_layout(self.h, self.deal)
            rows = np.concatenate(own) if self.deal == self.world else own[self.band_first]  # image rows this job rendered
            py = rows[rng.integers(0, len(rows), pixels)]
            px = rng.integers(0, self.w, pixels)
            xs, ys, ps = np.repeat(px, self.spp), np.repeat(py, self.spp), np.tile(np.arange(self.spp), pixels)
            O.set_math(1)
            try:
                o_rgb, _ = O.Scene(od.ptr, od).trace_samples(self.w, self.h, self.spp, self.depth, xs, ys, ps)
            finally:
                O.set_math(0)
            want = o_rgb.reshape(pixels, self.spp, 3).sum(axis=1) / self.spp
            self.sync()
            flat = self.bg.gathered.reshape(self.deal * self.bg.pad_rows, self.w, 3)
            idx = D.band_row_index(py, self.deal, D.BAND_ROWS, self.bg.pad_rows)
            got = flat[self.m.torch.as_tensor(idx, device=flat.device), self.m.torch.as_tensor(px, device=flat.device)].cpu().numpy() / self.spp
            return {"rel_linf_vs_cpu_ref": float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-3))), "tolerance": 1e-5, "pixels": pixels,
                    "ranks_covered": sorted({int(b) for b in (py // D.BAND_ROWS) % self.deal}) if self.deal == self.world else [self.band_first],
                    "note": "mean radiance of whole pixels of the timed (gathered) frame vs the CPU oracle (libm math), untimed"}
        except Exception as e:
            return {"rel_linf_vs_cpu_ref": None, "note": f"unavailable: {type(e).__name__}: {e}"}

    def one_stream_kernel_ms(self):
        """Untimed: this rank's render on ONE stream with every launch bracketed by HIP events on the launch stream, so that the
        per-kernel durations are not inflated by a co-scheduled batch and add up to (at most) the step."""
        tparams = self._params(time_kernels=True)
        prev = os.environ.get("PTX_STREAMS")
        os.environ["PTX_STREAMS"] = "1"
        try:
            self.render(tparams)  # warm
            t1 = time.perf_counter()
            st1 = self.render(tparams)
            ms = (time.perf_counter() - t1) * 1e3
        finally:
            if prev is None:
                del os.environ["PTX_STREAMS"]
            else:
                os.environ["PTX_STREAMS"] = prev
        return st1, ms

    def close(self):
        if self.scene is not None:
            self.scene.close()
            self.scene = None
        self.bg = self.rgb = None


def collective_block(m, job, backend, bd):
    """What the exchange library saw: enough to attribute a scaling loss from the one line and to show it ran over N ranks."""
    dist, torch = m.dist, m.torch
    lib = None
    if backend == "nccl" and not job.rehearse:
        try:
            lib = "rccl/nccl " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            lib = f"nccl (version unavailable: {e})"
    elif job.world > 1:
        lib = "gloo (host memory; a rehearsal backend, never the product path)"
    per_peer = job.bg.pad_rows * job.w * 3 * 8 if job.world > 1 else 0
    return {"backend": backend if job.world > 1 else None, "world": dist.get_world_size() if job.world > 1 else 1, "library": lib,
            "exchange": "one group of point-to-point sends of raw sums into rank 0 per step (BandGather)" if job.world > 1 else "none (one rank)",
            "bytes_per_peer_per_step": per_peer, "bytes_into_rank0_per_step": per_peer * (job.world - 1), **bd,
            "note": "untimed extra step taken apart: each rank's render of its bands alone, then the exchange, then rank 0's film"}


def measure_job(m, job, steps, warmup, parity_pixels=64):
    """warm-up + timed steps of `job`, the breakdown, the timed frame's parity; rank 0 gets the dictionary."""
    elapsed = job.timed(steps, warmup)
    parity = job.frame_parity(parity_pixels)
    bd = job.breakdown()
    if job.rank != 0:
        return None
    ms = elapsed / steps * 1e3
    samples = job.rows * job.w * job.spp if job.deal != job.world else job.w * job.h * job.spp
    return {"workload": job.workload, "n_gpus": job.world, "rows": job.rows if job.deal != job.world else job.h, "of_rows": job.h,
            "band_step": job.band_step, "samples_per_step": samples, "steps": steps, "warmup": warmup,
            "ms_per_step": None if job.rehearse else ms, "value": None if job.rehearse else samples / ms * 1e-3, "unit": "Msamples/s",
            "parity": parity, "collective": collective_block(m, job, job.backend, bd)}


def measure_extra_workload(m, dev, local_dev, name, steps=3, warmup=1, parity_pixels=64):
    """One more BASELINE configuration on ONE GPU: `warmup` + `steps` queued frames exactly like the headline's (render of the
    band share + banded film, framebuffer left on the device), one untimed one-stream pass with every launch bracketed by HIP
    events (the dominant kernel's time), the tracked counter bytes of a step against the step as timed, and the timed frame's
    parity against the oracle."""
    workload, band_first, band_step = EXTRA_WORKLOADS[name]
    job = Job(m, workload, 0, 1, dev, local_dev, "nccl", band_first=band_first, band_step=band_step)
    try:
        res = measure_job(m, job, steps, warmup, parity_pixels)
        st1, _ = job.one_stream_kernel_ms()
        kms = {k: v for k, v in st1["kernel_ms"].items() if v}
        dom = max(kms, key=kms.get) if kms else None
        tc = tracked_counters(workload) or {}
        hb = tc.get("hbm_bytes_per_step") if band_step <= 1 else None  # the tracked profile is of the whole frame
        ms = res["ms_per_step"]
        res.update({"dominant_kernel": {"bounce": "k_bounce", "trace": "k_trace", "shade": "k_shade_pool"}.get(dom, dom),
                    "dominant_kernel_ms_one_stream": kms.get(dom), "kernel_ms_one_stream": kms,
                    "hbm_frame": {"traffic_per_step": hb, "achieved": hb / (ms * 1e-3) * 1e-9, "unit": "GB/s", "frac": hb / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBS,
                                  "source": tc.get("source"), "calibrated": bool(tc.get("calibrated"))} if hb else None})
        res.pop("collective", None)
        return res
    finally:
        job.close()


def extras_child(args):
    """`bench.py --extras-child`: the other single-GPU configurations in a process of their own -- one JSON line per workload as
    soon as it is done, so that a hang, an abort or a timeout here costs the parent that workload only, never its headline."""
    m = Mods(False)
    torch = m.torch
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: this benchmark has no CPU fallback")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    for name in (args.extras_child.split(",") if args.extras_child != "all" else list(EXTRA_WORKLOADS)):
        try:
            res = measure_extra_workload(m, dev, 0, name)
        except Exception as e:  # noqa: BLE001
            res = {"value": None, "error": f"{type(e).__name__}: {e}"}
        print(json.dumps({"extra_workload": name, "result": res}), flush=True)
        torch.cuda.empty_cache()


def run_extras_in_child(timeout_s):
    """Starts `bench.py --extras-child all` as a CHILD process (never an exec of this one: it holds the GPU), collects its
    per-workload lines until it exits or the deadline passes, then ends exactly that process group."""
    import signal
    import subprocess
    import threading
    out = {}
    cmd = [sys.executable, os.path.abspath(__file__), "--extras-child", "all"]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, start_new_session=True)

    def reader():
        for line in proc.stdout:
            line = line.strip()
            if line.startswith("{"):
                try:
                    d = json.loads(line)
                    out[d["extra_workload"]] = d["result"]
                except (ValueError, KeyError):
                    pass

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    try:
        proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)  # the exact process group this function started
        except ProcessLookupError:
            pass
        proc.wait()
    th.join(timeout=5)
    for name in EXTRA_WORKLOADS:
        if name not in out:
            out[name] = {"value": None, "error": f"child process ended (rc {proc.returncode}) or ran past {timeout_s:.0f} s before this workload reported"}
    return out


def launch_ranks(n, argv):
    """One process per GPU, exactly as the driver launches them: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py <argv>` (here with `--standalone --local-addr 127.0.0.1` instead of a fixed master port),
    as a child of this process.  The
    parent imports neither torch nor the library, so no GPU state exists here; it relays the child's output (rank 0
    prints the one JSON line) and returns its exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: the launcher picks (and holds) a free rendezvous port itself -- choosing one here by bind-then-close could
    # lose it to another job on a shared node between the close and the launcher's bind
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", f"--nproc-per-node={n}", "--local-addr", "127.0.0.1",
           os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def roofline_block(args, job, world, counts, kernel_ms, launches, trace_ms_total, trace_launches, ms_per_step, elapsed, copy_gbs):
    """`roofline` of the headline line; rank 0."""
    scene_name, spp = job.scene_name, job.spp
    samples = job.w * job.h * spp
    sstats = job.sstats
    fused = launches["bounce"] > 0
    b_total, b_trace = algorithmic_bytes(counts, spp, scene_name != "shirley")
    # dominant kernel: algorithmic work of one step's launches / their summed HIP-event time (one stream)
    trace_ms_step = trace_ms_total
    n_launch = max(trace_launches, 1.0)
    avg_launch_s = trace_ms_step * 1e-3 / n_launch
    in_lds = bool(sstats["traversal_in_lds"])
    tc = tracked_counters(args.workload) or {}
    traffic = tc.get("trace_hbm_bytes_per_launch")
    walk_flop = counts["nodes_tested"] * FLOP_PER_NODE_TEST + (counts["prims_tested"] + counts["floor_tested"]) * FLOP_PER_SLOT_SCAN
    shade_flop = counts["segments"] * FLOP_PER_SEGMENT_SHADE if fused else 0.0  # k_trace does not shade
    alg_flop = walk_flop + shade_flop
    flops_achieved = alg_flop / (trace_ms_step * 1e-3) * 1e-12 if trace_ms_step > 0 else 0.0
    bytes_achieved = b_trace / (trace_ms_step * 1e-3) * 1e-9 if trace_ms_step > 0 else 0.0
    hbm_block = {"traffic": traffic, "achieved": (traffic / avg_launch_s * 1e-9) if (traffic and avg_launch_s > 0) else None,
                 "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "peak_measured_copy": copy_gbs,
                 "source": tc.get("source", "not profiled"), "calibrated": bool(tc.get("calibrated")),
                 "note": "fabric bytes per launch of the dominant kernel from rocprofv3 counters -- reads by request size (128 n128 + 64 n64 + 32 n32; on gfx950 every "
                         "request is 128 B and FETCH_SIZE tallies it at 64: profiles/r04_fetch_calibration.json) + WRITE_SIZE -- over the live launch duration; "
                         "Infinity-Cache hits are requests too"}
    hbm_block["frac"] = (hbm_block["achieved"] / hbm_block["peak"]) if hbm_block["achieved"] else None
    if in_lds:
        # tree + packets are LDS-resident: node / slot reads never reach HBM, the binding pipe is vector issue
        # `achieved` / `frac` price COUNTED work only: the walk's flops (node tests and packet slots counted by the kernels and
        # equal to the oracle's).  k_bounce also shades in the same launch; that arithmetic is an estimate per segment and is
        # reported beside it (incl_estimated_shade), not inside the figure people compare across rounds.
        walk_achieved = walk_flop / (trace_ms_step * 1e-3) * 1e-12 if trace_ms_step > 0 else 0.0
        roofline = {"bound": "valu_f64", "kernel": "k_bounce (walk + shade of a bounce in one launch)" if fused else "k_trace", "achieved": walk_achieved, "peak": F64_VECTOR_PEAK_TFLOPS * world,
                    "unit": "TFLOP/s", "frac": walk_achieved / (F64_VECTOR_PEAK_TFLOPS * world), "traffic": traffic,
                    "incl_estimated_shade": {"achieved": flops_achieved, "frac": flops_achieved / (F64_VECTOR_PEAK_TFLOPS * world),
                                             "note": f"+ {FLOP_PER_SEGMENT_SHADE:g} flop per segment shaded, a hand estimate of the reference's arithmetic outside Scene.intersect"} if fused else None,
                    "algorithmic_flop_per_launch": walk_flop / n_launch,
                    "flop_model": f"{FLOP_PER_NODE_TEST:g} per Bbox.is_hit + {FLOP_PER_SLOT_SCAN:g} per packet slot scanned (reference arithmetic, binary64)",
                    "algorithmic_flop_per_step": {"walk": walk_flop, "shade": shade_flop},
                    "issue": {k: tc.get(k) for k in ("valu_busy", "valu_issue_from_insts", "lane_util", "useful_issue_frac", "lds_busy", "lds_bank_conflict_share", "duration_cycles_source", "source")},
                    "hbm": hbm_block}
    else:
        roofline = {"bound": "hbm", "kernel": "k_bounce (walk from HBM / L2 + shade of a bounce in one launch)" if fused else "k_trace", "achieved": bytes_achieved, "peak": HBM_PEAK_GBS * world,
                    "unit": "GB/s", "frac": bytes_achieved / (HBM_PEAK_GBS * world), "traffic": traffic,
                    "algorithmic_bytes_per_launch": b_trace / n_launch,
                    "served_from": "l1/l2/infinity cache/hbm (the tree is larger than LDS)",
                    "issue": {k: tc.get(k) for k in ("valu_busy", "lane_util", "useful_issue_frac", "source")},
                    "hbm": hbm_block}
        if roofline["frac"] > 1.0:  # algorithmic bytes served from cache: say so instead of claiming > 100 % of HBM
            roofline["note"] = "algorithmic bytes exceed what HBM could deliver: node re-reads are served by L2 / Infinity Cache; see hbm.traffic"
    roofline.update({"launches_per_step": n_launch, "avg_launch_ms": trace_ms_step / n_launch,
                     "timing": "HIP events on the launch stream, one-stream pass (PTX_STREAMS=1)",
                     "pipeline": {"bytes_per_sample": b_total / samples, "achieved": b_total * args.steps / elapsed * 1e-9,
                                  "unit": "GB/s (algorithmic, SURVEY section 8 D)"}})
    # second kernel: the shade stage moves the bytes.  Algorithmic HBM bytes of one step: every segment reads
    # its queue entry (ray 48 B + path state 32 B [+ carried emission 32 B]), its hit slot (4 B; not in a fused launch, where it
    # stays in the wave) and what the shade step needs of the hit: t (8 B) on sphere scenes; on triangle scenes nothing is stored
    # -- the step re-reads the 80-byte triangle and recomputes (t, u, v) (PT_RECOMPUTE_HIT) -- ; a segment that survives writes the
    # next entry (80 B [+ 32 B]), a path that ends writes one 32-byte contribution.  survivors = segments - samples.
    emit_b = 32 if scene_name == "cornell" else 0
    hit_b = 12 if scene_name == "shirley" else 4 + 80  # slot + t; triangle scenes: slot + the triangle's record again
    segs = counts["segments"]
    shade_bytes = segs * (80 + emit_b + hit_b) + max(segs - counts["samples"], 0) * (80 + emit_b) + counts["samples"] * 32
    shade_s = kernel_ms["shade"] * 1e-3  # rank 0's share; the ranks run side by side, so job bytes / this = aggregate rate
    shade_gbs = shade_bytes / shade_s * 1e-9 if shade_s > 0 else None
    if shade_s > 0 and not fused:
        roofline["shade"] = {"bound": "hbm", "kernel": "k_shade_pool", "achieved": shade_gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                             "frac": (shade_gbs / (HBM_PEAK_GBS * world)) if shade_gbs else None,
                             "algorithmic_bytes_per_step": shade_bytes, "ms_per_step_one_stream": kernel_ms["shade"],
                             "timing": "HIP events on the launch stream, one-stream pass (PTX_STREAMS=1)"}
    if fused:
        # the same kernel against HBM: what a bounce must move -- the queue entry in, the survivor's entry out or the
        # path's contribution; the hit record stays inside the wave (its 4-byte slot never reaches memory, t goes through L2)
        bounce_bytes = shade_bytes - segs * 4
        bs = kernel_ms["bounce"] * 1e-3
        roofline["bytes"] = {"bound": "hbm", "kernel": "k_bounce", "achieved": bounce_bytes / bs * 1e-9 if bs > 0 else None,
                             "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": bounce_bytes / bs * 1e-9 / (HBM_PEAK_GBS * world) if bs > 0 else None,
                             "algorithmic_bytes_per_step": bounce_bytes, "ms_per_step_one_stream": kernel_ms["bounce"]}
    # how much of the step the vector pipe is busy: sum over stages of (one-stream kernel time x the stage's tracked
    # SQ_ACTIVE_INST_VALU share) against the step as timed.  Trace and shade of two batches run side by side on every
    # CU, so this -- not either kernel's own roofline -- is what the frame converges to.
    vb = {st_: tc.get(f"valu_busy_{st_}_time_weighted") for st_ in ("trace", "shade", "bounce")}
    if all(vb[st_] is not None for st_ in vb if kernel_ms[st_] > 0) and any(kernel_ms[st_] > 0 for st_ in vb):
        valu_ms = sum(kernel_ms[st_] * vb[st_] for st_ in vb if kernel_ms[st_] > 0)
        roofline["frame"] = {"bound": "valu_issue", "valu_busy_ms_per_step": valu_ms, "ms_per_step": ms_per_step,
                             "frac": valu_ms / ms_per_step if ms_per_step > 0 else None,
                             "valu_busy_share": {st_: vb[st_] for st_ in vb if kernel_ms[st_] > 0}, "source": tc.get("source"),
                             "note": "vector-pipe busy time of the step's kernels (one-stream durations x tracked counter shares) / the step"}
    # frame-level HBM traffic: counter-measured bytes of every kernel of a step (tracked profile) over the step as timed here
    hb = tc.get("hbm_bytes_per_step")
    if hb:
        roofline["hbm_frame"] = {"bound": "hbm", "traffic_per_step": hb, "achieved": hb / (ms_per_step * 1e-3) * 1e-9, "peak": HBM_PEAK_GBS * world,
                                 "unit": "GB/s", "frac": hb / (ms_per_step * 1e-3) * 1e-9 / (HBM_PEAK_GBS * world), "peak_measured_copy": copy_gbs,
                                 "peak_guide_copy": 6290.0,
                                 "by_stage": tc.get("hbm_bytes_per_step_by_stage"), "source": tc.get("source"), "calibrated": bool(tc.get("calibrated")),
                                 "note": "sum over the step's kernels of rocprofv3 counter bytes (sized read requests + WRITE_SIZE) x launches, from the tracked profile; "
                                         "peak_measured_copy = a torch f32 copy_ on this box, peak_guide_copy = the guide's float4 copy (MI355X_MICROARCH.md)"}
    return roofline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="shirley_1080p_spp64_d8", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the `workloads` block (N = 1: the other single-GPU BASELINE configurations in a child process; N > 1: config 5 across the ranks)")
    ap.add_argument("--workloads-timeout", type=float, default=240.0, help="N = 1: seconds the child process that times the other configurations may take")
    ap.add_argument("--extras-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=None)
    ap.add_argument("--passes-per-batch", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank path on fewer GPUs than ranks (ranks share devices)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU needed and nothing rendered: start the ranks and walk the multi-rank code path (band exchange over gloo on "
                         "synthetic bands, breakdown, CPU baseline, config-5 block, the JSON line) with value null -- checks that "
                         "`bench.py --gpus N` can start its own ranks and what its line will carry")
    args = ap.parse_args()

    if args.extras_child:
        return extras_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, before anything in
        # this process has touched the GPU (it never does), and pass rank 0's JSON line and the exit code through
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    rehearse = args.rehearse_launch
    m = Mods(rehearse)
    torch, dist, D, P = m.torch, m.dist, m.D, m.P
    import datetime
    if rehearse:
        backend, dev, local_dev = "gloo", torch.device("cpu"), 0
        if world > 1:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
    else:
        if not torch.cuda.is_available():
            raise SystemExit("no GPU visible: this benchmark has no CPU fallback")
        backend = args.backend
        n_dev = torch.cuda.device_count()
        if backend == "nccl" and local_rank >= n_dev:
            raise SystemExit(f"rank {rank}: local rank {local_rank} but only {n_dev} GPU(s) visible")
        local_dev = local_rank % n_dev  # gloo rehearsal: ranks may share a device
        torch.cuda.set_device(local_dev)
        dev = torch.device("cuda", local_dev)
        if world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))
            else:
                dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
    cpu_seconds = args.cpu_seconds if args.cpu_seconds is not None else (1.0 if rehearse else 15.0)

    # the timed steps are QUEUED (PTX_RENDER_ASYNC, ptx_film_resolve_banded_queue): frame k + 1 is being launched while frame k's
    # bands travel and its film runs; the fence around the K steps waits for all of it.  No event timing in the timed region.
    job = Job(m, args.workload, rank, world, dev, local_dev, backend, passes_per_batch=args.passes_per_batch, rehearse=rehearse)
    scene_name, w, h, spp, depth = job.scene_name, job.w, job.h, job.spp, job.depth
    elapsed = job.timed(args.steps, args.warmup)
    timed_parity = job.frame_parity()       # rank 0: whole pixels of the timed, gathered frame against the oracle
    rehearsal_ok = None
    if rehearse and rank == 0:
        import numpy as np
        img = D.ungather(job.bg.gathered.numpy(), h, world)
        rehearsal_ok = bool((img[:, 0, 0] == np.arange(h) + 0.25 * (job.n_frames - 1)).all())
    bd = job.breakdown()

    counts = kernel_ms = launches = None
    trace_ms_total = trace_launches = one_stream_ms = 0.0
    if not rehearse:
        job.sstats = job.scene.stats()
        st1, one_stream_ms = job.one_stream_kernel_ms()
        kernel_ms = {k: st1["kernel_ms"][k] for k in ("generate", "trace", "shade", "bounce", "accum", "film")}
        launches = {k: st1["kernel_launches"][k] for k in kernel_ms}
        # the dominant kernel: k_bounce (walk + shade of one bounce in one launch) where it runs, else k_trace
        dom = "bounce" if launches["bounce"] > 0 else "trace"
        # untimed: work counters for the algorithmic-bytes figure (whole job = sum over ranks)
        cst = job.render(job._params(count_work=True))
        keys = ("samples", "segments", "nodes_tested", "prims_tested", "floor_tested")
        cvec = torch.tensor([float(cst[k]) for k in keys] + [kernel_ms[dom], launches[dom]], dtype=torch.float64, device=job._cdev())
        if world > 1:
            # counters: sum over ranks; trace time: the slowest rank bounds the job, launches: per rank
            summed = cvec.clone()
            dist.all_reduce(summed, op=dist.ReduceOp.SUM)
            mx = cvec.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            cvec = torch.cat([summed[:5], mx[5:]])
        counts = dict(zip(keys, [int(v) for v in cvec[:5].tolist()]))
        trace_ms_total, trace_launches = float(cvec[5]), float(cvec[6])  # one step, one stream

    out = None
    if rank == 0:
        samples = w * h * spp
        ms_per_step = elapsed / args.steps * 1e3
        value = samples * args.steps / elapsed * 1e-6
        out = {
            "metric": METRIC, "value": None if rehearse else value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None if rehearse else ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "none (launch rehearsal over gloo: no GPU used, nothing rendered, nothing measured)" if rehearse else "synthetic",
            "config": {"workload": args.workload, "scene": scene_name, "width": w, "height": h, "spp": spp,
                       "max_bounces": depth, "samples_per_step": samples,
                       "sharding": f"{world} rank(s), interleaved {D.BAND_ROWS}-row bands, 1 gather/step" if world > 1 else "1 rank"},
            "collective": collective_block(m, job, backend, bd),
        }
        if rehearse:
            out["rehearsal"] = {"bands_arrived_in_place": rehearsal_ok}
            out["roofline"] = None
        else:
            sstats = job.sstats
            out["config"].update({"tree_nodes": sstats["tree_nodes"], "tree_depth": sstats["tree_depth"], "leaf_slots": sstats["leaf_slots"]})
            try:
                copy_gbs = measured_hbm_copy_gbs(torch, dev)
            except Exception:
                copy_gbs = None
            out["roofline"] = roofline_block(args, job, world, counts, kernel_ms, launches, trace_ms_total, trace_launches, ms_per_step, elapsed, copy_gbs)
            out["kernel_ms_per_step"] = dict(kernel_ms, one_stream_step_ms=one_stream_ms)
            out["kernel_launches_per_step"] = launches
            # the host-framebuffer entry point the CLI and the OCaml stub call (one 24 B/pixel device-to-host copy more)
            host_api = None
            if world == 1:
                try:
                    import numpy as np
                    fb = np.zeros((h, w, 3))  # the caller's image, allocated and touched once like the reference's Bimage
                    job.scene.pin_image(fb)   # ... and pinned for its lifetime, as the CLI and the OCaml stub do (ptx_image_pin)
                    job.scene.render(w, h, spp, depth, out=fb)
                    t2 = time.perf_counter()
                    for _ in range(3):
                        job.scene.render(w, h, spp, depth, out=fb)
                    host_ms = (time.perf_counter() - t2) * 1e3 / 3
                    job.scene.unpin_image()
                    host_api = {"ptx_render_ms": host_ms, "msamples_per_s": samples / host_ms * 1e-3,
                                "note": "ptx_render: whole frame on this GPU, post-gamma framebuffer copied into the caller's host image, pinned by the caller for its lifetime like the CLI's (ptx_image_pin; PCIe inclusive: never `value`)"}
                except Exception as e:
                    host_api = {"ptx_render_ms": None, "note": f"unavailable: {e}"}
            out["host_api"] = host_api
            out["work"] = {**counts, "segments_per_sample": counts["segments"] / samples,
                           "nodes_per_segment": counts["nodes_tested"] / max(counts["segments"], 1),
                           "prims_per_segment": counts["prims_tested"] / max(counts["segments"], 1)}
        out["parity"] = {"timed_frame": timed_parity}
        if not args.no_cpu_baseline:
            # rank 0 alone, after the timed region; the other ranks wait at the barrier below
            try:
                out["cpu_baseline"], cpu_rgb, cpu_spp = cpu_baseline(args.workload, cpu_seconds)
                if not rehearse:
                    # the metric's third part: per-pixel relative L-inf of the post-gamma framebuffer against the CPU reference, at
                    # the sample count the CPU leg rendered (untimed, the whole frame on rank 0's GPU; the oracle is only the checker)
                    import numpy as np
                    pp = P.render_params(w, h, cpu_spp, depth)
                    g_raw = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
                    g_rgb = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
                    job.scene.render_raw_device(pp, g_raw.data_ptr(), job.stream)
                    P.film_resolve_device(local_dev, w, h, cpu_spp, g_raw.data_ptr(), g_rgb.data_ptr(), job.stream)
                    torch.cuda.synchronize()
                    g = g_rgb.cpu().numpy()
                    del g_raw, g_rgb
                    out["parity"].update({"rel_linf_vs_cpu_ref": float(np.max(np.abs(g - cpu_rgb) / np.maximum(np.abs(cpu_rgb), 1e-3))),
                                          "tolerance": 1e-5, "spp": cpu_spp,
                                          "note": "CPU ref in libm math (as the OCaml runtime); in the shared pt_math mode the tests show bit-exact raw sums"})
                else:
                    out["parity"].update({"rel_linf_vs_cpu_ref": None, "tolerance": 1e-5, "note": "launch rehearsal: nothing was rendered"})
            except Exception as e:  # the oracle is test infrastructure; its absence must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": effective_cpus(), "kind": "port",
                                       "sample": f"unavailable: {e}"}
        # the headline fields are final here: make them durable before anything else is measured in (or beside) this process
        print("bench.py: headline line (final; the `workloads` block follows in the one line on stdout): " + json.dumps(out), file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()  # (the ranks that waited for rank 0's CPU leg)

    want_extras = not args.no_workloads and args.workload == "shirley_1080p_spp64_d8"
    if want_extras and world > 1:
        # config 5 -- what the >= 6x target is quoted on -- across the same ranks: 1 warm-up + 3 steps, its own parity
        job.close()
        if not rehearse:
            torch.cuda.empty_cache()
        res = None
        try:
            j5 = Job(m, SCALING_WORKLOAD, rank, world, dev, local_dev, backend, rehearse=rehearse)
            try:
                res = measure_job(m, j5, 3, 1)
            finally:
                j5.close()
        except Exception as e:  # noqa: BLE001
            res = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["workloads"] = {SCALING_WORKLOAD: res}
    elif want_extras and rank == 0 and not rehearse:
        # the other single-GPU configurations, in a child process with a deadline: whatever happens there, the headline stands
        job.close()
        del job
        torch.cuda.empty_cache()
        P.lib().ptx_release_workspaces()
        out["workloads"] = run_extras_in_child(args.workloads_timeout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if rehearse and rank == 0 and rehearsal_ok is False:
        raise SystemExit("band exchange rehearsal failed")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
