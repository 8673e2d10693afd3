#!/usr/bin/env python3
"""ptx_scene_create wall time (BVH build + flattening + upload) for the bench scenes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
for name, mk in (("shirley", lambda: H.shirley_spheres(1920, 1080)), ("cornell", lambda: H.cornell_box(1024, 1024)),
                 ("ganesha_like_150k", lambda: H.ganesha_like(1920, 1080))):
    hs = mk()
    for rep in range(3):
        t0 = time.perf_counter()
        sc = P.Scene(hs.ptr, 0, keepalive=hs)
        ms = (time.perf_counter() - t0) * 1e3
        st = sc.stats()
        sc.close()
    print(f"{name}: ptx_scene_create {ms:.2f} ms (tree build {st['build_ms']:.2f} ms, {st['tree_nodes']} nodes)")
