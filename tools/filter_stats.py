#!/usr/bin/env python3
"""binary32 node-filter statistics from ptx_stats (count_work renders): undecided lane-tests and wave steps that ran the
binary64 fallback, per node test, on the bench scenes.  usage: tools/filter_stats.py [shirley|cornell|ganesha ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
w, h, spp, depth = 960, 540, 8, 8
for name in (sys.argv[1:] or ["shirley", "cornell", "ganesha"]):
    hs = {"shirley": lambda: H.shirley_spheres(w, h), "cornell": lambda: H.cornell_box(w, h, 12.0), "ganesha": lambda: H.ganesha_like(w, h, 150000, 7)}[name]()
    sc = P.Scene(hs.ptr, 0, keepalive=hs)
    raw = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    st = sc.render_raw_device(P.render_params(w, h, spp, depth, count_work=True), raw.data_ptr())
    und, fb, n = st["filter_undecided"], st["filter_fallback_steps"], max(st["nodes_tested"], 1)
    print(f"{name}: nodes tested {n}  undecided lane-tests {und} ({und / n:.2e})  wave steps through the fallback {fb} ({fb * 64 / n:.2e} of the steps if all 64 lanes were live)")
