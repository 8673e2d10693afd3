"""CPU model of the camera rays' packet walk (pt_trace_packet) on the Shirley scene: per 8x8 tile, the wave-level node visits,
leaves taken, slots scanned in lockstep -- and how many of those slots a per-leaf cone test (one lane per slot: can ANY ray
of the tile's bounding cone reach this sphere's line test?) would leave.  Sizes the `PT_PACKET_CULL` idea before it is built.

  python tools/sim_packet_cull.py [width height [tile_stride]]
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
    stride = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    d = O.desc_shirley(w, h)
    arrs = d.arrays()
    sc = O.Scene(d.ptr, d)
    bbox, info, order = sc.tree()
    cam = arrs["camera"]
    sx, sy, sz, sr = arrs["sphere_x"], arrs["sphere_y"], arrs["sphere_z"], arrs["sphere_r"]
    rng = np.random.default_rng(1)
    tot = dict(tiles=0, node_visits=0, leaves=0, slots=0, slots_kept=0, slots_any_cand=0, root_iters=0, lane_node_tests=0)
    for ty in range(0, (h + 7) // 8, stride):
        for tx in range(0, (w + 7) // 8, stride):
            px = tx * 8 + (np.arange(64) & 7)
            py = ty * 8 + (np.arange(64) >> 3)
            valid = (px < w) & (py < h)
            cx = (px + rng.random(64)) / w
            cy = 1.0 - (py + rng.random(64)) / h
            dirs = np.stack([cam[0] + cam[2] * cx, cam[1] + cam[3] * cy, -np.ones(64)], axis=1)
            dirs /= np.linalg.norm(dirs, axis=1)[:, None]
            inv = 1.0 / dirs
            # the tile's cone
            axis = dirs[valid].mean(axis=0)
            axis /= np.linalg.norm(axis)
            cos_t = (dirs[valid] @ axis).min() * (1 - 1e-9)
            sin_t = np.sqrt(max(0.0, 1 - cos_t * cos_t)) * (1 + 1e-9)
            tbest = np.full(64, np.inf)
            sign = dirs[0] >= 0  # (one sign group per tile nearly always)
            tot["tiles"] += 1

            def visit(node, act):
                tot["node_visits"] += 1
                tot["lane_node_tests"] += int(act.sum())
                b = bbox[node]
                t0 = b[None, 0:3] * inv
                t1 = b[None, 3:6] * inv
                a = np.minimum(t0, t1).max(axis=1)
                bb = np.maximum(t0, t1).min(axis=1)
                hit = act & (np.maximum(a, 0.0) <= np.minimum(bb, tbest))
                if not hit.any():
                    return
                if info[node, 0] == 1:
                    first, length = info[node, 2], info[node, 3]
                    tot["leaves"] += 1
                    for k in range(length):
                        pid = order[first + k]
                        if pid < 0:
                            continue
                        tot["slots"] += 1
                        c = np.array([sx[pid], sy[pid], sz[pid]])
                        r = sr[pid]
                        # cone test (double cone: the reference's discriminant is the LINE's)
                        p = abs(c @ axis)
                        q = np.sqrt(max(c @ c - p * p, 0.0))
                        keep = not (q * cos_t - p * sin_t > r * (1 + 1e-6) + 1e-6 * np.sqrt(c @ c))
                        bp = dirs @ c
                        wv = dirs * bp[:, None] - c[None, :]
                        disc = r * r - (wv * wv).sum(axis=1)
                        cand = hit & (disc >= 0)
                        if cand.any():
                            tot["slots_any_cand"] += 1
                            assert keep, "cone test culled a sphere a ray hits"
                            tot["root_iters"] += 1
                            cc = c @ c - r * r
                            qq = np.where(bp < 0, bp - np.sqrt(np.maximum(disc, 0)), bp + np.sqrt(np.maximum(disc, 0)))
                            t = np.where(cc < 0, qq, cc / qq)
                            upd = cand & (t >= 0) & (t <= tbest)
                            tbest[upd] = t[upd]
                        if keep:
                            tot["slots_kept"] += 1
                    return
                ax, lhs, rhs = info[node, 1], info[node, 2], info[node, 3]
                first, second = (lhs, rhs) if sign[ax] else (rhs, lhs)
                visit(first, hit)
                visit(second, hit)

            visit(0, valid.copy())
    n = tot["tiles"]
    print({k: (v / n if k != "tiles" else v) for k, v in tot.items()})
    print("per tile: node visits %.1f (x ~30 instr), leaves %.2f, slots scanned %.1f (x ~20 instr), kept by the cone test %.1f, "
          "slots with a candidate %.1f" % (tot["node_visits"] / n, tot["leaves"] / n, tot["slots"] / n, tot["slots_kept"] / n,
                                          tot["slots_any_cand"] / n))


if __name__ == "__main__":
    main()
