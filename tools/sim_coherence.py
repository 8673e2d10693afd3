#!/usr/bin/env python3
"""CPU model of wave utilisation for bounce-1 rays of the Shirley scene under different queue orders:
per-lane while-while traversal (one ray per lane, private stack) against the packet walk (64 rays share one
(node, mask) stack), for (a) the order the shade kernel produces today (512-entry workgroup windows binned by
direction octant), (b) larger windows sorted by (octant, origin cell), (c) a global sort.  Diagnostic only: decides
whether a ray sort is worth building.  usage: tools/sim_coherence.py [width height]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 360)
rng = np.random.default_rng(1)
d = O.desc_shirley(W, H)
A = d.arrays()
sc = O.Scene(d.ptr, d)
bbox, info, order = sc.tree()
cam = A["camera"]
cx_, cy_, cz_, cr_ = A["sphere_x"], A["sphere_y"], A["sphere_z"], A["sphere_r"]
n_nodes = len(bbox)
# leaf slot -> sphere (padding = -1)
leaf_prims = {}
for n in range(n_nodes):
    if info[n, 0]:
        sl = order[info[n, 2]: info[n, 2] + info[n, 3]]
        leaf_prims[n] = sl[sl >= 0]

def primary(xs, ys):
    u = (xs + rng.random(len(xs))) / W
    v = 1.0 - (ys + rng.random(len(xs))) / H
    dd = np.stack([cam[0] + cam[2] * u, cam[1] + cam[3] * v, -np.ones(len(xs))], 1)
    return dd / np.linalg.norm(dd, axis=1, keepdims=True)

def bounce_rays(xs, ys):
    dirs = primary(xs, ys)
    o = np.zeros_like(dirs)
    t, prim, _ = sc.intersect_rays(o, dirs)
    hit = prim >= 0
    p = dirs[hit] * t[hit, None]
    c = np.stack([cx_[prim[hit]], cy_[prim[hit]], cz_[prim[hit]]], 1)
    nrm = (p - c) / cr_[prim[hit], None]
    # cosine-weighted direction about the normal (every surface treated as Lambertian: the least coherent case)
    u1, u2 = rng.random(len(p)), rng.random(len(p))
    r, th = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(nrm[:, 0:1]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
    tx = np.cross(nrm, a); tx /= np.linalg.norm(tx, axis=1, keepdims=True)
    ty = np.cross(nrm, tx)
    nd = tx * (r * np.cos(th))[:, None] + ty * (r * np.sin(th))[:, None] + nrm * np.sqrt(1 - u1)[:, None]
    return p + 1e-3 * nd, nd

def octant(dv):
    return (dv[:, 0] >= 0) * 1 + (dv[:, 1] >= 0) * 2 + (dv[:, 2] >= 0) * 4

class Walk:
    """ordered traversal of a set of rays with ONE octant (so the child order is shared): records per ray the number of
    box tests and leaf slots, per call the number of (node, active rays) steps"""
    def __init__(self, o, dv):
        self.o, self.d, self.inv = o, dv, 1.0 / dv
        self.t = np.full(len(o), np.inf)
        self.visits = np.zeros(len(o), dtype=np.int64)
        self.slots = np.zeros(len(o), dtype=np.int64)
        self.node_steps = 0     # packet: one step per (node, any active ray)
        self.node_active = 0    # sum of active rays over those steps
        self.slot_steps = 0
        self.slot_active = 0
        self.dirs = octant(dv[:1])[0]
    def rec(self, node, idx):
        if len(idx) == 0:
            return
        self.node_steps += 1
        self.node_active += len(idx)
        self.visits[idx] += 1
        t0 = (bbox[node, 0:3] - self.o[idx]) * self.inv[idx]
        t1 = (bbox[node, 3:6] - self.o[idx]) * self.inv[idx]
        a = np.minimum(t0, t1).max(1)
        b = np.maximum(t0, t1).min(1)
        hit = np.maximum(a, 0.0) <= np.minimum(b, self.t[idx])
        idx = idx[hit]
        if len(idx) == 0:
            return
        if info[node, 0]:
            pr = leaf_prims[node]
            self.slot_steps += len(pr)
            self.slot_active += len(pr) * len(idx)
            self.slots[idx] += len(pr)
            for s in pr:
                f = np.array([cx_[s], cy_[s], cz_[s]]) - self.o[idx]
                dd = self.d[idx]
                bp = (f * dd).sum(1)
                aa = (dd * dd).sum(1)
                w = dd * (bp / aa)[:, None] - f
                disc = cr_[s] ** 2 - (w * w).sum(1)
                ok = disc >= 0
                q = bp + np.where(bp < 0, -1, 1) * np.sqrt(np.maximum(aa * disc, 0))
                cc = (f * f).sum(1) - cr_[s] ** 2
                with np.errstate(all="ignore"):
                    th = np.where(cc < 0, q / aa, cc / q)
                ok &= (th >= 0) & (th <= self.t[idx])
                self.t[idx[ok]] = th[ok]
        else:
            axis, lhs, rhs = info[node, 1], info[node, 2], info[node, 3]
            first, second = (lhs, rhs) if (self.dirs >> axis) & 1 else (rhs, lhs)
            self.rec(first, idx)
            self.rec(second, idx)

def evaluate(o, dv, label):
    """queue order = the given order; waves = consecutive 64 entries"""
    n = len(o) // 64 * 64
    tot_useful_nodes = tot_useful_slots = 0
    lane_node_steps = lane_slot_steps = 0      # per-lane model: max over the wave's rays
    pk_node_steps = pk_slot_steps = 0          # packet model: union of nodes, per octant subgroup
    for w0 in range(0, n, 64):
        oo, dd = o[w0:w0 + 64], dv[w0:w0 + 64]
        oc = octant(dd)
        vis = np.zeros(64, dtype=np.int64); slo = np.zeros(64, dtype=np.int64)
        for k in np.unique(oc):
            m = np.nonzero(oc == k)[0]
            wk = Walk(oo[m], dd[m])
            wk.rec(0, np.arange(len(m)))
            vis[m], slo[m] = wk.visits, wk.slots
            pk_node_steps += wk.node_steps
            pk_slot_steps += wk.slot_steps
        tot_useful_nodes += vis.sum(); tot_useful_slots += slo.sum()
        lane_node_steps += vis.max(); lane_slot_steps += slo.max()
    print(f"{label:46s} rays {n:7d}  nodes/ray {tot_useful_nodes / n:5.1f} slots/ray {tot_useful_slots / n:5.1f} | "
          f"per-lane util: node {tot_useful_nodes / (64 * lane_node_steps):.2f} slot {tot_useful_slots / (64 * lane_slot_steps):.2f} | "
          f"packet util: node {tot_useful_nodes / (64 * pk_node_steps):.2f} slot {tot_useful_slots / (64 * max(pk_slot_steps, 1)):.2f} | "
          f"wave steps per ray: per-lane {(lane_node_steps + lane_slot_steps) / n:.3f}  packet {(pk_node_steps + pk_slot_steps) / n:.3f}")

# bounce-1 rays in the order the GPU produces them: shade workgroups of 512 primary entries = 8 tiles of 8x8 pixels
# (a 64 x 8 pixel strip of one pass), survivors binned by octant inside the workgroup's slice
all_o, all_d, wg_id = [], [], []
g = 0
for ty in range(0, H - 7, 8):
    for tx in range(0, W - 63, 64):
        xs = np.concatenate([np.tile(np.arange(8), 8) + tx + 8 * k for k in range(8)])
        ys = np.concatenate([np.repeat(np.arange(8), 8) + ty for k in range(8)])
        o, dv = bounce_rays(xs, ys)
        k = np.argsort(octant(dv), kind="stable")
        all_o.append(o[k]); all_d.append(dv[k]); wg_id.append(np.full(len(k), g)); g += 1
o = np.concatenate(all_o); dv = np.concatenate(all_d); wg = np.concatenate(wg_id)
sub = slice(0, min(len(o), 64 * 700))
print(f"{len(o)} bounce-1 rays from {W}x{H}, {g} shade workgroups; evaluating {sub.stop}")
evaluate(o[sub], dv[sub], "as produced (512 window, octant bins)")
def cell_key(o, dv, cells):
    lo, hi = np.percentile(o, 1, axis=0), np.percentile(o, 99, axis=0)
    q = np.clip(((o - lo) / (hi - lo) * cells).astype(int), 0, cells - 1)
    # Morton-ish: interleave x and z cell (the scene is a slab in y), octant major
    return (octant(dv) * cells + q[:, 0]) * cells + q[:, 2]
for win in (4096, 32768, len(o)):
    for cells in (8, 32):
        oo, dd = o.copy(), dv.copy()
        for s in range(0, len(o), win):
            k = np.argsort(cell_key(o[s:s + win], dv[s:s + win], cells), kind="stable")
            oo[s:s + win], dd[s:s + win] = o[s:s + win][k], dv[s:s + win][k]
        evaluate(oo[sub], dd[sub], f"window {win if win < len(o) else 'all':>6}, key (octant, {cells}x{cells} cell)")
perm = rng.permutation(len(o))
evaluate(o[perm][sub], dv[perm][sub], "random order")

# ---- keys that predict the LENGTH of the walk instead of its place: elevation of the direction above the ground
# plane (rays that climb leave the slab of spheres after a few node tests, grazing rays cross the whole scene)
up = np.array([cx_[0], cy_[0], cz_[0]])  # the ground sphere's centre is straight "down" from the scene in camera space
up = -up / np.linalg.norm(up)
def elev_key(dv, levels, with_octant):
    e = np.clip(((dv @ up) + 1.0) * 0.5 * levels, 0, levels - 1e-9).astype(int)
    return (octant(dv) * levels + e) if with_octant else e
for win in (512, 4096):
    for levels, wo in ((4, True), (8, False), (16, False), (8, True)):
        oo, dd = o.copy(), dv.copy()
        for s in range(0, len(o), win):
            k = np.argsort(elev_key(dv[s:s + win], levels, wo), kind="stable")
            oo[s:s + win], dd[s:s + win] = o[s:s + win][k], dv[s:s + win][k]
        evaluate(oo[sub], dd[sub], f"window {win:>5}, key ({'octant, ' if wo else ''}elevation/{levels})")

# ---- how far any key could go: sort each 512-window by the TRUE number of node tests of each ray (an oracle key)
true_visits = np.zeros(len(o), dtype=np.int64)
for w0 in range(0, sub.stop, 64):
    oo, dd = o[w0:w0 + 64], dv[w0:w0 + 64]
    oc = octant(dd)
    for k in np.unique(oc):
        m = np.nonzero(oc == k)[0]
        wk = Walk(oo[m], dd[m]); wk.rec(0, np.arange(len(m)))
        true_visits[w0 + m] = wk.visits + 4 * wk.slots
for win in (512, 4096):
    oo, dd = o.copy(), dv.copy()
    for s in range(0, sub.stop, win):
        k = np.argsort(true_visits[s:s + win], kind="stable")
        oo[s:s + win], dd[s:s + win] = o[s:s + win][k], dv[s:s + win][k]
    evaluate(oo[sub], dd[sub], f"window {win:>5}, ORACLE key (true walk length)")
# elevation x height of the origin above the plane
def elev_height_key(o_, dv, le, lh):
    e = np.clip(((dv @ up) + 1.0) * 0.5 * le, 0, le - 1e-9).astype(int)
    hgt = (o_ - o_.mean(0)) @ up
    hq = np.clip(np.digitize(hgt, np.quantile(hgt, np.linspace(0, 1, lh + 1)[1:-1])), 0, lh - 1)
    return e * lh + hq
for le, lh in ((4, 2), (8, 2), (4, 4)):
    oo, dd = o.copy(), dv.copy()
    for s in range(0, len(o), 512):
        k = np.argsort(elev_height_key(o[s:s + 512], dv[s:s + 512], le, lh), kind="stable")
        oo[s:s + 512], dd[s:s + 512] = o[s:s + 512][k], dv[s:s + 512][k]
    evaluate(oo[sub], dd[sub], f"window   512, key (elevation/{le} x origin height/{lh})")

# ---- slab-crossing length: distance the ray travels before it leaves the slab that holds the small primitives
ctr = np.stack([cx_, cy_, cz_], 1)[cr_ < 10]
hh = ctr @ up
lo_h, hi_h = (hh - cr_[cr_ < 10]).min(), (hh + cr_[cr_ < 10]).max()
def slab_key(o_, dv, levels):
    e = dv @ up
    h0 = o_ @ up
    L = np.where(e > 0, (hi_h - h0) / np.maximum(e, 1e-9), (h0 - lo_h) / np.maximum(-e, 1e-9))
    L = np.clip(L, 1e-3, 1e3)
    q = np.log(L)
    edges = np.quantile(q, np.linspace(0, 1, levels + 1)[1:-1])
    return np.digitize(q, edges)
def slab_key_fixed(o_, dv, levels):  # fixed log-spaced edges (what a kernel can do without a quantile pass)
    e = dv @ up
    h0 = o_ @ up
    L = np.where(e > 0, (hi_h - h0) / np.maximum(e, 1e-9), (h0 - lo_h) / np.maximum(-e, 1e-9))
    q = np.log2(np.clip(L / (hi_h - lo_h), 2.0 ** -1, 2.0 ** (levels - 1.001)))  # in slab thicknesses
    return np.clip((q + 1).astype(int), 0, levels - 1)
for name, fn in (("slab length quantiles/8", lambda a, b: slab_key(a, b, 8)), ("slab length log2 steps/8", lambda a, b: slab_key_fixed(a, b, 8))):
    oo, dd = o.copy(), dv.copy()
    for s in range(0, len(o), 512):
        k = np.argsort(fn(o[s:s + 512], dv[s:s + 512]), kind="stable")
        oo[s:s + 512], dd[s:s + 512] = o[s:s + 512][k], dv[s:s + 512][k]
    evaluate(oo[sub], dd[sub], f"window   512, key ({name})")
print("slab", lo_h, hi_h)

# ---- tail cut with wave-local resume: a chunk stops when fewer than T rays are still walking; the stragglers' (tiny,
# stackless) states wait in a per-wave list and are resumed together once 64 - T of them have gathered
def tail_cut_model(cost, T, waves=64, resume_overhead=3.0):
    """cost: per-ray walk length in queue order.  A wave owns every `waves`-th chunk.  Returns wave steps per ray."""
    n = len(cost) // 64 * 64
    chunks = cost[:n].reshape(-1, 64).astype(float)
    steps = 0.0
    for w in range(waves):
        pend = []
        mine = chunks[w::waves]
        for k, c in enumerate(mine):
            last = k == len(mine) - 1
            def run(c, T_):
                nonlocal steps
                if T_ <= 1 or len(c) < T_:
                    steps += c.max() if len(c) else 0.0
                    return np.zeros(0)
                s = np.sort(c)[-T_]            # when the T-th longest finishes, T - 1 are left
                steps += s
                return c[c > s] - s
            left = run(c, T)
            pend.extend(left.tolist())
            while len(pend) >= 64 - T + 1 or (last and pend):
                take, pend = np.array(pend[:64]) + resume_overhead, pend[64:]
                left = run(take, 0 if (last and not pend) else T)
                pend.extend(left.tolist())
    return steps / n
cost_elev = None
oo, dd = o.copy(), dv.copy()
for s in range(0, len(o), 512):
    k = np.argsort(elev_key(dv[s:s + 512], 8, False), kind="stable")
    oo[s:s + 512], dd[s:s + 512] = o[s:s + 512][k], dv[s:s + 512][k]
cost = np.zeros(sub.stop)
for w0 in range(0, sub.stop, 64):
    a, b = oo[w0:w0 + 64], dd[w0:w0 + 64]
    oc = octant(b)
    for k in np.unique(oc):
        m = np.nonzero(oc == k)[0]
        wk = Walk(a[m], b[m]); wk.rec(0, np.arange(len(m)))
        cost[w0 + m] = wk.visits + 0.7 * wk.slots   # a slot step costs ~0.7 of a node step
print("tail-cut model on the elevation-sorted queue (combined node + slot steps per ray):")
for T in (0, 4, 8, 16, 24, 32):
    print(f"  T = {T:2d}: {tail_cut_model(cost, T):.3f} wave steps per ray")
