#!/usr/bin/env python3
"""Renders rank 0's share of an N-rank job of the headline frame a few times (for rocprofv3 --kernel-trace: where the
per-rank step goes when the share is small).  usage: tools/one_band_share.py [world=8] [reps=4] [passes_per_batch=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
from path_tracer_ocaml_amd import distributed as D

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ppb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w, h, spp, depth = 1920, 1080, 64, 8
hs = H.shirley_spheres(w, h)
scene = P.Scene(hs.ptr, 0, keepalive=hs)
bg = D.BandGather(h, w, 0, world, torch.device("cuda", 0))
params = P.render_params(w, h, spp, depth, band_rows=D.BAND_ROWS, band_first=0, band_step=world, passes_per_batch=ppb)
stream = torch.cuda.current_stream().cuda_stream
for k in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.render_raw_device(params, bg.part.data_ptr(), stream)
    torch.cuda.synchronize()
    print("render %d: %.3f ms" % (k, (time.perf_counter() - t0) * 1e3))
