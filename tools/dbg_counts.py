#!/usr/bin/env python3
"""Segments per bounce (work counters of renders with max_bounces = 1..8): how the path population decays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
w, h, spp = 1920, 1080, 8
hs = H.shirley_spheres(w, h)
sc = P.Scene(hs.ptr, 0, keepalive=hs)
raw = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
prev = {"segments": 0, "nodes_tested": 0, "prims_tested": 0}
for d in range(1, 9):
    st = sc.render_raw_device(P.render_params(w, h, spp, d, count_work=True), raw.data_ptr())
    print("bounce", d - 1, {k: st[k] - prev[k] for k in prev}, "share of segments %.3f" % ((st["segments"] - prev["segments"]) / (w * h * spp)))
    prev = {k: st[k] for k in prev}
