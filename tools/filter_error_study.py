#!/usr/bin/env python3
"""How wrong can the binary32 box filter of the walk be, and does it ever DECIDE wrongly?  ("the filter's margin".)

The walk decides Bbox.is_hit (bbox.ml:40-56, a boolean of binary64 quantities) in binary32 where a derived error bound allows it:
u~ = hi~ - lo~ is within 13 units of 2^-24 (mag + max|o|) max|1/d| (+ 2^-24 t) of the reference's hi - lo, the margin m2 is 32 such
units (+ 2^-21 t), |u~| >= m2 decides, everything else goes to the reference's own binary64 arithmetic.  That bound is a hand
derivation; this tool measures it.  It runs the PRODUCTION box test (PtTraverser::begin + test_box on the tagged per-octant record,
through ptx_debug_filter_error of a -DPT_FILTER_DEBUG=1 build) on adversarial (ray, node, closest hit so far) triples against the
real tree of the ganesha-like mesh, with the reference's binary64 lo / hi evaluated beside it on the device, and reports

  * decided tests whose outcome differs from the reference's boolean      -- must be 0 (a wrong decision is a wrong pixel);
  * tests whose FINAL outcome (fallback included) differs from it          -- must be 0;
  * the largest |u~ - (hi - lo)| relative to the margin and to the derived bound, and the share of undecided tests, per generator.

usage (on an MI355X box):  tools/build_variant.sh fdbg "-DPT_FILTER_DEBUG=1"
                           PTX_LIB=$PWD/build_variants/libptx_fdbg.so python tools/filter_error_study.py [millions of triples per generator, default 8] [seed]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_ocaml_amd as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

MAX_FINITE = 1.7976931348623157e308
U24 = 2.0 ** -24


def ulp32(x):
    """spacing of binary32 numbers at |x| (vectorised)"""
    a = np.abs(x).astype(np.float32)
    return (np.nextafter(a, np.float32(np.inf)) - a).astype(np.float64)


def unit(v):
    return v / np.linalg.norm(v, axis=1)[:, None]


def boundary_points(rng, box, kind):
    """points on the boxes' corners (0), edges (1) or faces (2): box is (n, 6) = mn.xyz, mx.xyz"""
    n = box.shape[0]
    mn, mx = box[:, :3], box[:, 3:]
    pick = rng.integers(0, 2, size=(n, 3)).astype(bool)
    p = np.where(pick, mx, mn)  # a corner
    if kind >= 1:  # let one axis run free along an edge
        ax = rng.integers(0, 3, size=n)
        t = rng.random(n)
        free = mn[np.arange(n), ax] + t * (mx[np.arange(n), ax] - mn[np.arange(n), ax])
        p[np.arange(n), ax] = free
    if kind >= 2:  # and a second one: a point of a face
        ax2 = (ax + 1 + rng.integers(0, 2, size=n)) % 3
        t = rng.random(n)
        p[np.arange(n), ax2] = mn[np.arange(n), ax2] + t * (mx[np.arange(n), ax2] - mn[np.arange(n), ax2])
    return p


def gen_edge_aimed(rng, n, bbox, scene_size, thin=False, axis_near=False, t_near=False):
    """rays aimed at the boundary of a node's box within a few binary32 ulps, from near, middling and far origins"""
    if thin:  # nodes whose box is thinnest (triangles' leaf boxes): what the mesh's fallback steps are made of
        ext = (bbox[:, 3:] - bbox[:, :3]).min(axis=1)
        cand = np.argsort(ext)[: max(1, bbox.shape[0] // 4)]
        nodes = cand[rng.integers(0, cand.size, size=n)]
    else:
        nodes = rng.integers(0, bbox.shape[0], size=n)
    box = bbox[nodes]
    p = boundary_points(rng, box, rng.integers(0, 3))
    p = p + rng.integers(-4, 5, size=(n, 3)) * ulp32(p)  # +-4 binary32 ulps of the coordinate off the boundary
    dist = scene_size * 10.0 ** rng.uniform(-3.0, 2.0, size=n)
    dirs = unit(rng.normal(size=(n, 3)))
    if axis_near:  # one direction component tiny: |1 / d| large
        ax = rng.integers(0, 3, size=n)
        dirs[np.arange(n), ax] *= 10.0 ** rng.uniform(-12.0, -3.0, size=n)
        dirs = unit(dirs)
    o = p - dirs * dist[:, None]
    d = unit(p - o)
    if t_near:  # the closest hit so far sits within a few binary32 ulps of the box's entry distance: the t-clamp's own path
        t = np.linalg.norm(p - o, axis=1)
        tmax = t + rng.integers(-4, 5, size=n) * ulp32(t)
        tmax = np.maximum(tmax, 0.0)
    else:
        tmax = np.full(n, MAX_FINITE)
    return o, d, nodes.astype(np.int32), tmax


def gen_random(rng, n, bbox, scene_size):
    """rays as a render makes them: origins in and around the scene, any direction, any node"""
    centre = 0.5 * (bbox[0, :3] + bbox[0, 3:])
    o = centre + rng.normal(size=(n, 3)) * scene_size * 10.0 ** rng.uniform(-1.0, 1.0, size=(n, 1))
    d = unit(rng.normal(size=(n, 3)))
    nodes = rng.integers(0, bbox.shape[0], size=n).astype(np.int32)
    tmax = np.where(rng.random(n) < 0.5, MAX_FINITE, scene_size * 10.0 ** rng.uniform(-2.0, 1.0, size=n))
    return o, d, nodes, tmax


def main():
    millions = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
    L = P.lib()
    if not hasattr(L, "ptx_debug_filter_error"):
        raise SystemExit("this library has no ptx_debug_filter_error: build it with -DPT_FILTER_DEBUG=1 and point PTX_LIB at it")
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.ptx_debug_filter_error.restype = C.c_int32
    L.ptx_debug_filter_error.argtypes = [C.c_void_p, C.c_int64, dp, dp, ip, dp, dp]
    desc = O.desc_ganesha_like(1920, 1080)
    gpu = P.Scene(desc.ptr, 0, keepalive=desc)
    bbox, info, _ = O.Scene(desc.ptr, desc).tree()
    scene_size = float(np.abs(bbox[0, 3:] - bbox[0, :3]).max())
    rng = np.random.default_rng(seed)
    gens = [
        ("edge-aimed, any node", lambda n: gen_edge_aimed(rng, n, bbox, scene_size)),
        ("edge-aimed, thinnest boxes", lambda n: gen_edge_aimed(rng, n, bbox, scene_size, thin=True)),
        ("edge-aimed, axis-near directions", lambda n: gen_edge_aimed(rng, n, bbox, scene_size, axis_near=True)),
        ("edge-aimed, t_max at the entry distance", lambda n: gen_edge_aimed(rng, n, bbox, scene_size, t_near=True)),
        ("edge-aimed, thin + axis-near + t_max near", lambda n: gen_edge_aimed(rng, n, bbox, scene_size, thin=True, axis_near=True, t_near=True)),
        ("random rays, random nodes", lambda n: gen_random(rng, n, bbox, scene_size)),
    ]
    batch = 2_000_000
    report = {"scene": "ganesha-like, %d nodes" % bbox.shape[0], "seed": seed, "library": os.environ.get("PTX_LIB", "in-tree"), "generators": {}}
    total = wrong_total = final_wrong_total = 0
    for name, gen in gens:
        n_left = int(millions * 1e6)
        agg = dict(n=0, filtered=0, undecided=0, wrong=0, final_wrong=0, max_err_over_margin=0.0, max_err_over_bound=0.0, max_err_over_margin_decided=0.0)
        while n_left > 0:
            n = min(batch, n_left)
            n_left -= n
            o, d, nodes, tmax = gen(n)
            # a direction component that cancelled to exactly 0 has an infinite reciprocal: the walk takes Base's NaN-propagating
            # min / max for such a ray (pt_slab_hit_exact), the diagnostic's reference beside it does not -- outside the filter's domain
            with np.errstate(divide="ignore"):
                keep = np.isfinite(1.0 / d).all(axis=1) & np.isfinite(o).all(axis=1)
            o, d, nodes, tmax = o[keep], d[keep], nodes[keep], tmax[keep]
            n = int(keep.sum())
            if n == 0:
                continue
            o, d, tmax, nodes = np.ascontiguousarray(o), np.ascontiguousarray(d), np.ascontiguousarray(tmax), np.ascontiguousarray(nodes)
            out = np.zeros((n, 6))
            rc = L.ptx_debug_filter_error(gpu._h, n, o.ctypes.data_as(dp), d.ctypes.data_as(dp), nodes.ctypes.data_as(ip), tmax.ctypes.data_as(dp), out.ctypes.data_as(dp))
            if rc != 0:
                raise SystemExit("ptx_debug_filter_error failed: %s" % P.last_error())
            u32, m2, lo, hi, dec, fin = out.T
            ref = lo <= hi  # the reference's boolean (NaN-free: finite 1 / d by construction)
            ok = np.isfinite(m2)  # rays the filter applies to (a NaN margin sends every test to binary64)
            decided = ok & (dec >= 0)
            agg["n"] += n
            agg["filtered"] += int(ok.sum())
            agg["undecided"] += int((ok & (dec < 0)).sum())
            agg["wrong"] += int((decided & ((dec > 0.5) != ref)).sum())
            agg["final_wrong"] += int(((fin > 0.5) != ref).sum())
            with np.errstate(invalid="ignore", over="ignore"):
                err = np.abs(u32 - (hi - lo))
                t32 = tmax.astype(np.float32).astype(np.float64)
                m2_t = np.where(t32 < 2.0 ** 120, t32 * 2.0 ** -21, 0.0)
                bound = (13.0 / 32.0) * np.maximum(m2 - m2_t, 0.0) + 0.125 * m2_t  # 13 of the margin's 32 units + 2^-24 t
                fin_err = ok & np.isfinite(err) & (m2 > 0)
                if fin_err.any():
                    agg["max_err_over_margin"] = max(agg["max_err_over_margin"], float((err[fin_err] / m2[fin_err]).max()))
                    agg["max_err_over_bound"] = max(agg["max_err_over_bound"], float((err[fin_err] / bound[fin_err]).max()))
                dd = decided & np.isfinite(err) & (m2 > 0)
                if dd.any():
                    agg["max_err_over_margin_decided"] = max(agg["max_err_over_margin_decided"], float((err[dd] / m2[dd]).max()))
        agg["undecided_share"] = agg["undecided"] / max(agg["filtered"], 1)
        report["generators"][name] = agg
        total += agg["n"]
        wrong_total += agg["wrong"]
        final_wrong_total += agg["final_wrong"]
        print("%-44s n %10d  filter applies %10d  undecided %7.4f  WRONG DECISIONS %d  wrong final %d  max err / margin %.4f (decided: %.4f)  max err / bound %.4f"
              % (name, agg["n"], agg["filtered"], agg["undecided_share"], agg["wrong"], agg["final_wrong"], agg["max_err_over_margin"],
                 agg["max_err_over_margin_decided"], agg["max_err_over_bound"]), flush=True)
    report["total"] = total
    report["wrong_decisions"] = wrong_total
    report["wrong_final_outcomes"] = final_wrong_total
    print(json.dumps(report))
    if wrong_total or final_wrong_total:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
