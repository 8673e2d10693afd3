#!/usr/bin/env python3
"""What ONE rank of an N-rank job costs per step on this GPU, for N = 1, 2, 4, 8, on the headline frame and on BASELINE
config 5 (4K, spp 256): the render of the rank's interleaved 8-row bands (busiest rank = rank 0), and -- for rank 0 --
the banded film pass over the gathered layout.  The exchange itself (RCCL sends of pad_rows x W x 24 bytes per peer) needs
N GPUs and is not timed here.  Writes profiles/<tag>_band_share.json.  usage: tools/band_share_timing.py [tag]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
from path_tracer_ocaml_amd import distributed as D

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {"note": "one MI355X; ms per step, mean of 5 after 2 warm-up steps; rank 0 = the busiest rank of the deal"}
stream = torch.cuda.current_stream().cuda_stream
for name, (w, h, spp, depth) in {"shirley_1080p_spp64_d8": (1920, 1080, 64, 8), "shirley_4k_spp256_d8": (3840, 2160, 256, 8)}.items():
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    rows = []
    for world in (1, 2, 4, 8):
        bg = D.BandGather(h, w, 0, world, torch.device("cuda", 0))
        rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        params = P.render_params(w, h, spp, depth, band_rows=D.BAND_ROWS, band_first=0, band_step=world)

        def timed(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        render_ms = timed(lambda: scene.render_raw_device(params, bg.part.data_ptr(), stream))
        film_ms = timed(lambda: P.film_resolve_banded_device(0, w, h, spp, bg.gathered.data_ptr(), world, D.BAND_ROWS, bg.pad_rows,
                                                             rgb.data_ptr(), stream))
        peer_mb = bg.pad_rows * w * 24 / 1e6
        row = {"world": world, "rank0_rows": bg.local_rows, "render_ms": render_ms, "film_ms_rank0": film_ms,
               "bytes_per_peer_MB": peer_mb, "send_ms_at_153GBps_per_link": peer_mb / 153e3 * 1e3 if world > 1 else 0.0}
        row["step_ms_estimate"] = render_ms + film_ms + row["send_ms_at_153GBps_per_link"]
        rows.append(row)
        print(name, row)
    for r in rows:
        r["speedup_vs_1_estimate"] = rows[0]["step_ms_estimate"] / r["step_ms_estimate"]
    out[name] = rows
    scene.close()
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"{tag}_band_share.json"), "w"), indent=1)
