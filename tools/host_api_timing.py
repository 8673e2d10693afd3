#!/usr/bin/env python3
"""ptx_render (host framebuffer API, what the CLI calls): first call (workspace allocation) vs steady state, against the
device-resident path bench.py times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
w, h, spp, depth = 1920, 1080, 64, 8
hs = H.shirley_spheres(w, h)
sc = P.Scene(hs.ptr, 0, keepalive=hs)
for k in range(4):
    t0 = time.perf_counter()
    rgb, st = sc.render(w, h, spp, depth)
    print(f"ptx_render call {k}: {1e3 * (time.perf_counter() - t0):.1f} ms wall, render_ms {st['render_ms']:.1f}")
raw = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
p = P.render_params(w, h, spp, depth)
for k in range(3):
    t0 = time.perf_counter()
    sc.render_raw_device(p, raw.data_ptr())
    torch.cuda.synchronize()
    print(f"ptx_render_raw_device call {k}: {1e3 * (time.perf_counter() - t0):.1f} ms wall")
# the same call into an image the caller has pinned (ptx_image_pin), with the frame's tail pipelined in PTX_FINAL_SLABS row slabs
import numpy as np
for slabs in ("1", "2", "4", "8"):
    os.environ["PTX_FINAL_SLABS"] = slabs
    fb = np.zeros((h, w, 3))
    sc.pin_image(fb)
    sc.render(w, h, spp, depth, out=fb)
    t0 = time.perf_counter()
    for _ in range(5):
        sc.render(w, h, spp, depth, out=fb)
    print(f"ptx_render into a pinned image, PTX_FINAL_SLABS={slabs}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms")
    sc.unpin_image()
