#!/usr/bin/env python3
"""Where a progressive-photon-map run spends its time (ptx_ppm_stats).  usage: tools/ppm_timing.py [size iters photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
size = int(sys.argv[1]) if len(sys.argv) > 1 else 600
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
photons = int(sys.argv[3]) if len(sys.argv) > 3 else 75000
hs = H.cornell_box(size, size)
scene = P.Scene(hs.ptr, 0, keepalive=hs)
from path_tracer_ocaml_amd import abi
lights = H.lights_cornell(size, size)
params = abi.ppm_params(size, size, iterations=iters, photon_count=photons)
for rep in range(2):
    img, st = scene.ppm_render(params, lights)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()})
