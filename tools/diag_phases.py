#!/usr/bin/env python3
"""Where a k_bounce wave's life goes (queued rays' launches), from the -DPT_DIAG=5 and -DPT_DIAG=6 builds: ticks of the 100 MHz
clock summed over waves.  usage (GPU box): tools/diag_phases.py [shirley|cornell]   -- needs build_variants/libptx_diag5.so and
libptx_diag6.so (tools/build_variant.sh diag5 "-DPT_DIAG=5")"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:  # child: one library per process (PTX_LIB is read at import)
    sys.path.insert(0, ROOT)
    import torch
    import path_tracer_ocaml_amd as P
    from path_tracer_ocaml_amd import host as H
    name, mode = sys.argv[1], int(sys.argv[2])
    w, h, spp, depth = (1920, 1080, 64, 8) if name == "shirley" else (1024, 1024, 64, 16)
    hs = H.shirley_spheres(w, h) if name == "shirley" else H.cornell_box(w, h, 12.0)
    sc = P.Scene(hs.ptr, 0, keepalive=hs)
    raw = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    st = sc.render_raw_device(P.render_params(w, h, spp, depth, count_work=True), raw.data_ptr())
    life, walk, b, rest = st["segments"], st["nodes_tested"], st["prims_tested"], st["floor_tested"]
    if mode in (5, 8):
        print(f"{name}{' (camera rays)' if mode == 8 else ''}: wave life {life / 1e8:.3f} s summed; walks {walk / life:.3f}  shade steps + pushes {b / life:.3f}  rest {rest / life:.3f}; "
              f"{st['filter_fallback_steps']} walks of {walk / max(st['filter_fallback_steps'], 1) :.0f} clocks, "
              f"{st['filter_undecided']} shade steps of {b / max(st['filter_undecided'], 1) :.0f} clocks")
    elif mode == 7:
        print(f"{name}: survivors' pushes (bin key, reservation, stores) {b / life:.3f} of a wave's life, {b / max(st['filter_undecided'], 1):.0f} clocks per shade step")
    else:
        print(f"{name}: of the walks' time, leaf phases (packet scan + roots / element tests) {b / max(walk, 1):.3f}, node loop + begin {1 - b / max(walk, 1):.3f}")
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "shirley"
for mode in (5, 6, 7, 8):
    lib = os.path.join(ROOT, "build_variants", f"libptx_diag{mode}.so")
    if not os.path.exists(lib):
        continue
    env = dict(os.environ, PTX_LIB=lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), name, str(mode)], env=env, check=False)
