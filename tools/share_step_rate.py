#!/usr/bin/env python3
"""Steps per second of rank 0's share of an N-rank job on one GPU (render of its bands + the banded film; no exchange), with the
waiting calls and with the queued ones (PTX_RENDER_ASYNC / ptx_film_resolve_banded_queue): what the host's per-step work costs
when a step is a few milliseconds.  usage: tools/share_step_rate.py [world=8] [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
from path_tracer_ocaml_amd import distributed as D

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
w, h, spp, depth = 1920, 1080, 64, 8
hs = H.shirley_spheres(w, h)
scene = P.Scene(hs.ptr, 0, keepalive=hs)
bg = D.BandGather(h, w, 0, world, torch.device("cuda", 0))
rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for queued in (False, True, False, True):
    params = P.render_params(w, h, spp, depth, band_rows=D.BAND_ROWS, band_first=0, band_step=world, asynchronous=queued)

    def step():
        scene.render_raw_device(params, bg.part.data_ptr(), stream)
        P.film_resolve_banded_device(0, w, h, spp, bg.gathered.data_ptr(), world, D.BAND_ROWS, bg.pad_rows, rgb.data_ptr(), stream,
                                     wait=not queued)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    print("world %d %s: %.3f ms per step" % (world, "queued" if queued else "waited", (time.perf_counter() - t0) / steps * 1e3))
