#!/usr/bin/env python3
"""Times ONE rank's share of the headline frame (bands rank, rank+world, ...) on this GPU, for several batch
sizes: the per-rank cost that bounds strong scaling at N = world.  usage: tools/band_share_timing.py [world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
from path_tracer_ocaml_amd import distributed as D

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h, spp, depth = 1920, 1080, 64, 8
hs = H.shirley_spheres(w, h)
scene = P.Scene(hs.ptr, 0, keepalive=hs)
stream = torch.cuda.current_stream().cuda_stream
for ppb in (0, 64, 32, 16, 8):
    params = P.render_params(w, h, spp, depth, band_rows=D.BAND_ROWS, band_first=0, band_step=world, passes_per_batch=ppb)
    rows = P.local_rows(params)
    part = torch.zeros((rows, w, 3), dtype=torch.float64, device="cuda")
    for _ in range(2):
        scene.render_raw_device(params, part.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        scene.render_raw_device(params, part.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"world {world} rank 0: rows {rows}  passes_per_batch {ppb or 'auto'}: {ms:.2f} ms/step  -> whole job {w*h*spp/ms*1e-3:.0f} Msamples/s if every rank matched")
