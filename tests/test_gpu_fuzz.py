"""Random-geometry parity: Scene.intersect on the GPU against the oracle over scenes the stock ones never produce.

Round 2 put a binary32 filter in front of EVERY Bbox.is_hit (bbox.ml:40-56; kernels.hip test_box): it decides the
binary64 boolean from binary32 arithmetic with an error margin and hands what it cannot decide to the reference's own
arithmetic.  Its soundness is an argument about magnitudes, so this file attacks the magnitudes: radii over six decades,
whole scenes scaled from 2^-20 to 2^21 and beyond the binary32 range, origins far outside the scene, near-axis and
axis-aligned directions with enormous 1/d, un-normalised directions, and rays aimed at the EDGES of leaf boxes within a
few binary32 ulps (where entry and exit distances coincide and the filter must give up).  Bar: hit primitive, t_hit and
the work counters (nodes tested, slots tested) equal the oracle's bit for bit, and the undecided branch provably ran
(ptx_stats.filter_undecided > 0 where grazing rays are present).  Seeds are fixed."""
import ctypes as C
import os

import numpy as np
import pytest

from test_gpu_edge_cases import both, check_rays, make_desc, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import path_tracer_ocaml_amd as P
    assert P.lib().ptx_device_count() >= 1, P.last_error()
    return P


def sphere_soup(rng, n, scale, centre, decades=6.0):
    """n spheres, centres uniform in a cube of half-width `scale` around `centre`, radii log-uniform over `decades`."""
    c = centre + rng.uniform(-1.0, 1.0, (n, 3)) * scale
    r = scale * 0.2 * 10.0 ** rng.uniform(-decades, 0.0, n)
    return [(c[i, 0], c[i, 1], c[i, 2], r[i], int(rng.integers(0, 3))) for i in range(n)]


def triangle_soup(rng, n, scale, centre, decades=4.0):
    a = centre + rng.uniform(-1.0, 1.0, (n, 3)) * scale
    size = scale * 0.3 * 10.0 ** rng.uniform(-decades, 0.0, (n, 1))
    b = a + rng.normal(size=(n, 3)) * size
    c = a + rng.normal(size=(n, 3)) * size
    return [(a[i], b[i], c[i], int(rng.integers(0, 3))) for i in range(n)]


def ray_mix(rng, n, scale, centre, boxes=None):
    """Origins inside, near and FAR outside the scene; directions random, un-normalised, near-axis (|1/d| up to 1e18),
    exactly axis-aligned; and, if `boxes` (k, 6) is given, rays through points on box EDGES perturbed by a few binary32 ulps."""
    kinds = rng.integers(0, 6, n)
    o = centre + rng.uniform(-1.0, 1.0, (n, 3)) * scale * np.where(kinds[:, None] == 1, 40.0, 1.5)
    far = kinds == 2
    o[far] = centre + rng.normal(size=(far.sum(), 3)) * scale * 2.0 ** 22  # beyond the old omax guard when scale >= 1/4
    target = centre + rng.uniform(-1.0, 1.0, (n, 3)) * scale
    d = target - o
    with np.errstate(over="ignore"):
        d *= 10.0 ** rng.uniform(-3.0, 3.0, (n, 1))  # Scene.intersect does not ask for unit directions
    near_axis = kinds == 3
    ax = rng.integers(0, 3, n)
    d[near_axis, ax[near_axis]] *= 10.0 ** rng.uniform(-18.0, -6.0, near_axis.sum())
    on_axis = kinds == 4
    d[on_axis, ax[on_axis]] = np.where(rng.integers(0, 2, on_axis.sum()) == 0, 0.0, -0.0)
    if boxes is not None and len(boxes):
        graze = kinds == 5
        k = graze.sum()
        bx = boxes[rng.integers(0, len(boxes), k)]
        a1 = rng.integers(0, 3, k)
        a2 = (a1 + 1 + rng.integers(0, 2, k)) % 3
        a3 = 3 - a1 - a2
        pt = np.zeros((k, 3))
        rows = np.arange(k)
        # a point on the edge (axis a1 at its min or max, axis a2 at its min or max, axis a3 anywhere inside) ...
        pt[rows, a1] = bx[rows, a1 + 3 * rng.integers(0, 2, k)]
        pt[rows, a2] = bx[rows, a2 + 3 * rng.integers(0, 2, k)]
        lo3, hi3 = bx[rows, a3], bx[rows, a3 + 3]
        pt[rows, a3] = lo3 + (hi3 - lo3) * rng.uniform(0.0, 1.0, k)
        # ... moved by -4 .. +4 binary32 ulps of its magnitude
        pt += np.abs(pt) * rng.integers(-4, 5, (k, 3)) * 2.0 ** -24
        d[graze] = pt - o[graze]
    return o, d


def leaf_boxes(g_scene):
    bbox, info, _ = g_scene.tree()
    return bbox  # every node: internal boxes are unions of leaf boxes, their edges are grazed just the same


SCALES = [2.0 ** -20, 2.0 ** -7, 1.0, 2.0 ** 9, 2.0 ** 21]
# a soak run (tools/round_record.sh) adds PTX_FUZZ_SEEDS fresh seeds to the two LDS-resident soups below
# (PTX_FUZZ_SEED0 moves the range: a second soak takes other seeds than the first)
SOAK = [(100 + k, SCALES[k % len(SCALES)]) for k in range(int(os.environ.get("PTX_FUZZ_SEED0", "0")), int(os.environ.get("PTX_FUZZ_SEED0", "0")) + int(os.environ.get("PTX_FUZZ_SEEDS", "0")))]


@pytest.mark.parametrize("seed,scale", [(s, sc) for s, sc in enumerate(SCALES)] + [(7, 1.0), (8, 2.0 ** 21), (9, 2.0 ** -20)] + SOAK)
def test_sphere_soup_simd_leaf(P, oracle, seed, scale):
    from path_tracer_ocaml_amd import abi
    rng = np.random.default_rng(1000 + seed)
    centre = rng.uniform(-1.0, 1.0, 3) * scale * (50.0 if seed >= 7 else 3.0)  # seeds >= 7: the scene sits far from the origin
    d, keep = make_desc(abi, spheres=sphere_soup(rng, 300, scale, centre), leaf_kind=0, cutoff=16)
    o_scene, g_scene = both(P, oracle, d, keep)
    assert g_scene.stats()["traversal_in_lds"]
    o, dr = ray_mix(rng, 30000, scale, centre, leaf_boxes(g_scene))
    prims = check_rays(o_scene, g_scene, o, dr)
    assert (prims >= 0).sum() > 300 and (prims < 0).sum() > 300
    st = g_scene.intersect_rays(o, dr)[2]
    assert st["filter_undecided"] > 0, "no Bbox.is_hit was left to the binary64 fallback: the grazing rays did not graze"
    assert st["filter_fallback_steps"] > 0


@pytest.mark.parametrize("seed,scale", [(20, 2.0 ** -12), (21, 1.0), (22, 2.0 ** 14), (23, 2.0 ** 21)] + SOAK)
def test_mixed_soup_array_leaf_lds(P, oracle, seed, scale):
    """Array_leaf with spheres AND triangles (cornell's mixed leaf, cornell-box/bin/main.ml:93-155), LDS-resident."""
    from path_tracer_ocaml_amd import abi
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-1.0, 1.0, 3) * scale * 3.0
    d, keep = make_desc(abi, spheres=sphere_soup(rng, 60, scale, centre, 4.0), tris=triangle_soup(rng, 80, scale, centre),
                        leaf_kind=1, cutoff=3)
    o_scene, g_scene = both(P, oracle, d, keep)
    assert g_scene.stats()["traversal_in_lds"]
    o, dr = ray_mix(rng, 20000, scale, centre, leaf_boxes(g_scene))
    prims = check_rays(o_scene, g_scene, o, dr)
    assert (prims < 0).sum() > 100
    if scale >= 1.0:  # |det| < 1e-6 rejects every triangle of a tiny scene (triangle.ml:83): spheres only there
        assert (prims >= 0).sum() > 100
    assert g_scene.intersect_rays(o, dr)[2]["filter_undecided"] > 0


@pytest.mark.parametrize("oct_image", ["1", "0"])
@pytest.mark.parametrize("seed,scale", [(30, 1.0), (31, 2.0 ** 17), (32, 2.0 ** -9)])
def test_large_soup_walked_from_hbm(P, oracle, seed, scale, oct_image, monkeypatch):
    """More nodes than LDS holds: the threaded walk from HBM / L2 over the per-octant 32-byte node image (two loads per visit,
    PtThreadOctTag; the default) and over the shared image + skip table it replaces where a leaf does not fit its packed link
    (PTX_OCT_IMAGE=0) -- test_box, G32 branch.  Same hits, t and counters as the oracle on both."""
    from path_tracer_ocaml_amd import abi
    monkeypatch.setenv("PTX_OCT_IMAGE", oct_image)  # read when the scene handle is created
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-1.0, 1.0, 3) * scale * 5.0
    d, keep = make_desc(abi, spheres=sphere_soup(rng, 5000, scale, centre, 5.0), tris=triangle_soup(rng, 5000, scale, centre, 3.0),
                        leaf_kind=1, cutoff=4)
    o_scene, g_scene = both(P, oracle, d, keep)
    assert not g_scene.stats()["traversal_in_lds"]
    o, dr = ray_mix(rng, 40000, scale, centre, leaf_boxes(g_scene))
    prims = check_rays(o_scene, g_scene, o, dr)
    assert (prims >= 0).sum() > 300 and (prims < 0).sum() > 300
    assert g_scene.intersect_rays(o, dr)[2]["filter_undecided"] > 0


@pytest.mark.parametrize("leaf_kind,cutoff,scale", [(0, 16, 1e39), (1, 4, 1e39), (0, 16, 3e37), (1, 4, 1e300)])
def test_bounds_beyond_binary32_range(P, oracle, leaf_kind, cutoff, scale):
    """Coordinates the binary32 image cannot hold (or whose products overflow it): root_mag = +inf or the per-ray guard
    sends every test to binary64.  1e300: the quadratic's squares overflow binary64 itself -- same NaNs / misses on both sides."""
    from path_tracer_ocaml_amd import abi
    rng = np.random.default_rng(77)
    centre = np.array([0.3, -0.2, -4.0]) * scale
    d, keep = make_desc(abi, spheres=sphere_soup(rng, 120, scale, centre, 3.0), leaf_kind=leaf_kind, cutoff=cutoff)
    o_scene, g_scene = both(P, oracle, d, keep)
    o, dr = ray_mix(rng, 8000, scale, centre, leaf_boxes(g_scene))
    o[::2] = 0.0
    with np.errstate(over="ignore", invalid="ignore"):
        prims = check_rays(o_scene, g_scene, o, dr)
    if scale < 1e100:
        assert (prims >= 0).sum() > 100


def wall_soup(rng, n, scale, centre):
    """n axis-aligned rectangles (two triangles each): every triangle's box -- and every leaf that holds only coplanar ones -- is
    FLAT along one axis, the case PtTraverser::nested_hit settles in binary32."""
    tris = []
    for _ in range(n):
        k = int(rng.integers(0, 3))
        i, j = (k + 1) % 3, (k + 2) % 3
        p = centre + rng.uniform(-1.0, 1.0, 3) * scale
        ei, ej = scale * 10.0 ** rng.uniform(-3.0, 0.0, 2)
        q = [p.copy() for _ in range(4)]
        q[1][i] += ei
        q[2][i] += ei
        q[2][j] += ej
        q[3][j] += ej
        m = int(rng.integers(0, 3))
        tris += [(q[0], q[1], q[2], m), (q[0], q[2], q[3], m)]
    return tris


@pytest.mark.parametrize("seed,scale,cutoff", [(50, 1.0, 4), (51, 2.0 ** -11, 2), (52, 2.0 ** 18, 1), (53, 1.0, 1)])
def test_flat_boxes_axis_aligned_walls(P, oracle, seed, scale, cutoff):
    """Boxes of zero thickness: on every hit the slab test's hi - lo is exactly 0, so the generic filter can never decide it; the
    flat-box rule decides most of them in binary32 and must agree with the reference's boolean on all of them -- rays through
    the rectangles, along their planes, and through the edges of their (flat) boxes within +-4 binary32 ulps."""
    from path_tracer_ocaml_amd import abi
    rng = np.random.default_rng(1000 + seed)
    centre = rng.uniform(-1.0, 1.0, 3) * scale * 3.0
    d, keep = make_desc(abi, tris=wall_soup(rng, 200, scale, centre), leaf_kind=1, cutoff=cutoff)
    o_scene, g_scene = both(P, oracle, d, keep)
    assert g_scene.stats()["traversal_in_lds"]
    boxes = leaf_boxes(g_scene)
    flat = boxes[(boxes[:, :3] == boxes[:, 3:]).any(axis=1)]
    assert len(flat) > 20, "the scene was meant to have flat boxes"
    o, dr = ray_mix(rng, 30000, scale, centre, boxes)
    # a share of the rays exactly inside a wall's plane
    k = 3000
    bx = flat[rng.integers(0, len(flat), k)]
    ax = np.argmax(bx[:, :3] == bx[:, 3:], axis=1)
    o[:k, :] = centre + rng.uniform(-1.0, 1.0, (k, 3)) * scale
    o[np.arange(k), ax] = bx[np.arange(k), ax]
    dr[:k] = rng.normal(size=(k, 3))
    dr[np.arange(k), ax] = 0.0
    prims = check_rays(o_scene, g_scene, o, dr)
    assert (prims >= 0).sum() > 300 and (prims < 0).sum() > 300


def test_every_test_undecided_when_margin_is_everything(P, oracle):
    """A scene of identical concentric boxes seen edge-on: a large share of the tests is undecided; counters still equal."""
    from path_tracer_ocaml_amd import abi
    rng = np.random.default_rng(5)
    spheres = [(float(x), float(y), -8.0, 0.5, 0) for x in range(-6, 7) for y in range(-6, 7)]  # lattice: shared box planes
    d, keep = make_desc(abi, spheres=spheres, leaf_kind=0, cutoff=16)
    o_scene, g_scene = both(P, oracle, d, keep)
    # rays inside the planes y = k +- 0.5 and x = k +- 0.5 (exactly, and a few ulps off)
    n = 6000
    o = np.zeros((n, 3))
    o[:, 0] = rng.integers(-6, 7, n) + 0.5 * rng.choice([-1.0, 1.0], n)
    o[:, 1] = rng.integers(-6, 7, n) + 0.5 * rng.choice([-1.0, 1.0], n)
    o[:, :2] += np.abs(o[:, :2]) * rng.integers(-2, 3, (n, 2)) * 2.0 ** -24
    dr = np.tile([0.0, 0.0, -1.0], (n, 1))
    dr[: n // 2, 0] = rng.normal(size=n // 2) * 1e-9
    check_rays(o_scene, g_scene, o, dr)
    st = g_scene.intersect_rays(o, dr)[2]
    assert st["filter_undecided"] > n // 4


@pytest.mark.parametrize("seed,leaf_kind,cutoff", [(40, 0, 16), (41, 1, 4), (42, 1, 2)])
def test_random_scene_whole_paths(P, oracle, seed, leaf_kind, cutoff):
    """Whole paths (every material, checker UVs, glass chains) through random geometry in front of the camera:
    per-sample radiance bit-exact, counters equal."""
    from path_tracer_ocaml_amd import abi
    rng = np.random.default_rng(seed)
    centre = np.array([0.0, 0.0, -6.0])
    sph = sphere_soup(rng, 400, 2.5, centre, 1.0) + [(0.0, -1003.0, -6.0, 1000.0, 0)]  # + a checker ground, radius 1000
    tris = triangle_soup(rng, 200, 2.5, centre, 0.7) if leaf_kind == 1 else ()
    d, keep = make_desc(abi, spheres=sph, tris=tris, leaf_kind=leaf_kind, cutoff=cutoff)
    o_scene, g_scene = both(P, oracle, d, keep)
    w, h, spp, depth = 96, 48, 6, 10
    n = 6000
    xs, ys, ps = rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, spp, n)
    c, ct = o_scene.trace_samples(w, h, spp, depth, xs, ys, ps)
    g, st = g_scene.trace_samples(w, h, spp, depth, xs, ys, ps, count_work=True)
    assert np.array_equal(bits(g), bits(c))
    for k in ("segments", "nodes_tested", "prims_tested"):
        assert st[k] == ct[k], k
    assert ct["segments"] > 2 * n


@pytest.mark.parametrize("n_floor", [2, 4, 7])
def test_many_pretested_floor_triangles_with_a_tree_walked_from_hbm(P, oracle, n_floor):
    """Triangles tested before the tree (ganesha's Floor, main.ml:247-256): the walk from HBM / L2 keeps the first four in LDS and
    reads further ones from global memory.  2, 4 and 7 of them over a soup too large for LDS: hit primitive (floor slots included),
    t, and the floor / node / slot counters equal the oracle's; whole paths are traced per sample bit for bit."""
    from path_tracer_ocaml_amd import abi
    rng = np.random.default_rng(40 + n_floor)
    centre = np.array([0.0, 0.0, -30.0])
    d, keep = make_desc(abi, tris=triangle_soup(rng, 9000, 1.0, centre, 3.0), leaf_kind=1, cutoff=4)
    # n_floor large triangles fanned out below the soup (camera space: y down = lower values)
    fv = []
    for k in range(n_floor):
        a0, a1 = 2 * np.pi * k / n_floor, 2 * np.pi * (k + 1) / n_floor
        fv += [centre[0], centre[1] - 6.0, centre[2],
               centre[0] + 40 * np.cos(a0), centre[1] - 6.0 - 0.1 * k, centre[2] + 40 * np.sin(a0),
               centre[0] + 40 * np.cos(a1), centre[1] - 6.0 + 0.05 * k, centre[2] + 40 * np.sin(a1)]
    fv = np.array(fv, dtype=np.float64)
    fuv = np.tile(np.array([0.0, 0.0, 1.0, 0.0, 1.0, 1.0]), n_floor)
    fm = np.zeros(n_floor, dtype=np.int32)
    keep += [fv, fuv, fm]
    d.n_floor_triangles = n_floor
    d.floor_vertices = fv.ctypes.data_as(abi.c_double_p)
    d.floor_uv = fuv.ctypes.data_as(abi.c_double_p)
    d.floor_material = fm.ctypes.data_as(abi.c_int32_p)
    o_scene, g_scene = both(P, oracle, d, keep)
    assert not g_scene.stats()["traversal_in_lds"]
    n = 30000
    o = centre + rng.uniform(-2.0, 2.0, (n, 3)) + np.array([0.0, 3.0, 0.0])
    dr = rng.normal(size=(n, 3))
    dr[: n // 2, 1] = -np.abs(dr[: n // 2, 1]) - 0.3  # half of them aimed at the floor fan
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    t_c, p_c, ct = o_scene.intersect_rays(o, dr)
    t_g, p_g, st = g_scene.intersect_rays(o, dr)
    assert np.array_equal(p_g, p_c) and np.array_equal(bits(t_g), bits(t_c))
    for k in ("nodes_tested", "prims_tested", "floor_tested"):
        assert st[k] == ct[k], k
    n_slots = g_scene.stats()["leaf_slots"]
    assert st["floor_tested"] >= n and (p_c >= 0).sum() > 1000
    w, h, spp, depth = 64, 48, 3, 5
    xs, ys, ps = rng.integers(0, w, 3000), rng.integers(0, h, 3000), rng.integers(0, spp, 3000)
    g_rgb, _ = g_scene.trace_samples(w, h, spp, depth, xs, ys, ps)
    o_rgb, _ = o_scene.trace_samples(w, h, spp, depth, xs, ys, ps)
    assert np.array_equal(bits(g_rgb), bits(o_rgb))
