#!/usr/bin/env python3
"""binary32 node-filter statistics (needs a -DPT_F32_FILTER_STATS=1 build via PTX_LIB): undecided lane-tests and
wave-level fallbacks per node test, on the headline scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
w, h, spp, depth = 960, 540, 8, 8
hs = H.shirley_spheres(w, h)
sc = P.Scene(hs.ptr, 0, keepalive=hs)
raw = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
st = sc.render_raw_device(P.render_params(w, h, spp, depth, count_work=True), raw.data_ptr())
und = st["floor_tested"] & 0xffffffff
fb = st["floor_tested"] >> 32
print(f"nodes tested {st['nodes_tested']}  undecided lane-tests {und} ({und / st['nodes_tested']:.2e})  wave fallbacks {fb} (per wave-step if 64 lanes: {fb * 64 / st['nodes_tested']:.2e})")
