"""Edge cases the domain has: axis-aligned rays (0 * inf in the slab test), rays starting exactly on box planes,
coincident primitives (ties, un-splittable leaves), rays from inside spheres, degenerate triangles, tiny scenes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import path_tracer_ocaml_amd as P
    assert P.lib().ptx_device_count() >= 1, P.last_error()
    return P


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def make_desc(abi, spheres=(), tris=(), leaf_kind=1, cutoff=4, mats=None):
    """spheres: (x, y, z, r, mat); tris: (a, b, c, mat) with 3-vectors.  Returns (desc, keepalive)."""
    keep = []
    d = abi.SceneDesc()
    sp = np.array([s[:4] for s in spheres], dtype=np.float64).reshape(-1, 4)
    cols = [np.ascontiguousarray(sp[:, k]) for k in range(4)]
    sm = np.array([s[4] for s in spheres], dtype=np.int32)
    keep += cols + [sm]
    d.n_spheres = len(spheres)
    if len(spheres):
        d.sphere_x, d.sphere_y, d.sphere_z, d.sphere_r = [c.ctypes.data_as(abi.c_double_p) for c in cols]
        d.sphere_material = sm.ctypes.data_as(abi.c_int32_p)
    if len(tris):
        v = np.array([p for t in tris for p in t[:3]], dtype=np.float64).reshape(-1, 3)
        vc = [np.ascontiguousarray(v[:, k]) for k in range(3)]
        idx = np.arange(3 * len(tris), dtype=np.int32)
        uv = np.tile(np.array([0.0, 0.0, 1.0, 0.0, 1.0, 1.0]), len(tris))
        tm = np.array([t[3] for t in tris], dtype=np.int32)
        keep += vc + [idx, uv, tm]
        d.n_vertices, d.n_triangles = len(v), len(tris)
        d.vertex_x, d.vertex_y, d.vertex_z = [c.ctypes.data_as(abi.c_double_p) for c in vc]
        d.tri_indices = idx.ctypes.data_as(abi.c_int32_p)
        d.tri_uv = uv.ctypes.data_as(abi.c_double_p)
        d.tri_material = tm.ctypes.data_as(abi.c_int32_p)
    M = (abi.Material * 3)()
    T = (abi.Texture * 2)()
    T[0].kind = abi.PTX_TEX_SOLID
    T[0].even[:] = [0.8, 0.5, 0.3]
    T[1].kind = abi.PTX_TEX_CHECKER
    T[1].width, T[1].height = 8, 16
    T[1].even[:] = [0.9, 0.9, 0.9]
    T[1].odd[:] = [0.1, 0.2, 0.3]
    M[0].kind, M[0].texture = abi.PTX_MAT_LAMBERTIAN, 1
    M[1].kind, M[1].texture = abi.PTX_MAT_METAL, 0
    M[2].kind, M[2].index = abi.PTX_MAT_DIELECTRIC, 1.5
    keep += [M, T]
    d.n_materials, d.materials, d.n_textures, d.textures = 3, M, 2, T
    d.camera.lower_left_x, d.camera.lower_left_y, d.camera.view_x, d.camera.view_y = -1.0, -0.5, 2.0, 1.0
    d.background.kind = abi.PTX_BG_SKY
    d.background.horizon[:] = [1.0, 1.0, 1.0]
    d.background.zenith[:] = [0.5, 0.7, 1.0]
    d.leaf_kind, d.length_cutoff, d.num_bins = leaf_kind, cutoff, 32
    return d, keep


def both(P, oracle, d, keep):
    return oracle.Scene(C.pointer(d), keep), P.Scene(d, 0, keepalive=keep)


def check_rays(o_scene, g_scene, origins, dirs):
    t_c, p_c, ct = o_scene.intersect_rays(origins, dirs)
    t_g, p_g, st = g_scene.intersect_rays(origins, dirs)
    assert np.array_equal(p_g, p_c)
    assert np.array_equal(bits(t_g), bits(t_c))
    for k in ("nodes_tested", "prims_tested"):
        assert st[k] == ct[k], k
    return p_c


@pytest.mark.parametrize("leaf_kind,cutoff", [(0, 16), (1, 4)])
def test_axis_aligned_rays_and_origins_on_box_planes(P, oracle, leaf_kind, cutoff):
    """Direction components exactly 0 make 1/d = inf; an origin exactly on a slab plane then gives 0 * inf = NaN,
    which Base's NaN-propagating min/max turn into a MISS (bbox.ml:46-56).  Integer lattice: plenty of such rays."""
    from path_tracer_ocaml_amd import abi
    spheres = [(float(x), float(y), float(-6 - z), 0.5, (x + y + z) % 3) for x in range(-2, 3) for y in range(-2, 3) for z in range(3)]
    d, keep = make_desc(abi, spheres=spheres, leaf_kind=leaf_kind, cutoff=cutoff)
    o_scene, g_scene = both(P, oracle, d, keep)
    origins, dirs = [], []
    for ax in range(3):
        for sgn in (-1.0, 1.0, -0.0, 0.0):
            for a in np.arange(-3.0, 3.5, 0.5):
                for b in np.arange(-3.0, 3.5, 0.5):
                    dvec = np.zeros(3)
                    if sgn in (-1.0, 1.0) and not (np.signbit(sgn) and sgn == 0):
                        dvec[ax] = sgn
                    else:  # a zero component with a sign, plus a diagonal in the other two
                        dvec[ax] = sgn
                        dvec[(ax + 1) % 3] = 1.0
                        dvec[(ax + 2) % 3] = -1.0
                    o = np.array([0.0, 0.0, -7.0])
                    o[(ax + 1) % 3] += a
                    o[(ax + 2) % 3] += b
                    o[ax] += -10.0 * (1.0 if sgn >= 0 else -1.0) if dvec[ax] != 0 else 0.5  # 0.5 = exactly on bbox planes
                    origins.append(o)
                    dirs.append(dvec)
    prims = check_rays(o_scene, g_scene, np.array(origins), np.array(dirs))
    assert (prims >= 0).sum() > 100 and (prims < 0).sum() > 100


def test_coincident_spheres_tie_rule_and_unsplittable_leaf(P, oracle):
    """Identical centres: Proposal.create finds no finite scale -> one leaf with every element (shape_tree.ml:129-131,180);
    equal t: the LATER element wins (`t <= t_max`, shape_tree.ml:299-311 / lib.rs:171-176)."""
    from path_tracer_ocaml_amd import abi
    spheres = [(0.0, 0.0, -5.0, 1.0, k % 3) for k in range(7)]
    for leaf_kind, cutoff in ((1, 4), (0, 16)):
        d, keep = make_desc(abi, spheres=spheres, leaf_kind=leaf_kind, cutoff=cutoff)
        o_scene, g_scene = both(P, oracle, d, keep)
        assert g_scene.stats()["tree_nodes"] == 1
        rng = np.random.default_rng(1)
        dirs = np.array([[x, y, -1.0] for x in np.linspace(-0.3, 0.3, 15) for y in np.linspace(-0.3, 0.3, 15)])
        prims = check_rays(o_scene, g_scene, np.zeros_like(dirs), dirs)
        assert set(prims.tolist()) <= {-1, 6}  # the last of the coincident spheres
    # more than 16 coincident spheres cannot be one Simd_leaf packet
    d, keep = make_desc(abi, spheres=[(0.0, 0.0, -5.0, 1.0, 0)] * 20, leaf_kind=0, cutoff=16)
    with pytest.raises(P.PtxError, match="16 lanes"):
        P.Scene(d, 0, keepalive=keep)


def test_rays_from_inside_spheres_and_tiny_scenes(P, oracle):
    from path_tracer_ocaml_amd import abi
    d, keep = make_desc(abi, spheres=[(0.0, 0.0, 0.0, 3.0, 2), (0.5, 0.2, -1.0, 0.4, 1)], leaf_kind=0, cutoff=16)
    o_scene, g_scene = both(P, oracle, d, keep)
    rng = np.random.default_rng(2)
    dirs = rng.normal(size=(5000, 3))
    prims = check_rays(o_scene, g_scene, np.zeros((5000, 3)), dirs)  # the camera sits at the glass sphere's CENTRE
    # Reference quirks, reproduced bit for bit rather than "fixed":
    #  * inside a sphere (c < 0) the code takes t = q / a with q = b' + sign(b') sqrt(..) (lib.rs:151-157,
    #    sphere.ml:46-52); for b' < 0 that is the NEGATIVE root, so a ray that starts inside and points away from the
    #    centre misses its own sphere (the reference's scenes only ever start inside a sphere heading inwards);
    #  * at the exact centre b' = +-0 and the x86 packet code keys on the SIGN BIT of b' (lib.rs:153).
    assert 0.3 < (prims >= 0).mean() < 1.0
    off = np.tile([0.1, -0.2, 0.3], (5000, 1))
    inside = check_rays(o_scene, g_scene, off, dirs)
    toward_centre = (dirs @ -off[0]) >= 0  # b' = f . d with f = centre - origin
    assert (inside[toward_centre] >= 0).all() and (inside[~toward_centre] != 0).all()
    # whole renders of the same scene: per-sample parity, deep glass chains
    xs, ys, ps = rng.integers(0, 64, 4000), rng.integers(0, 32, 4000), rng.integers(0, 4, 4000)
    c, _ = o_scene.trace_samples(64, 32, 4, 12, xs, ys, ps)
    g, _ = g_scene.trace_samples(64, 32, 4, 12, xs, ys, ps)
    assert np.array_equal(bits(g), bits(c))


def test_degenerate_and_edge_on_triangles(P, oracle):
    """|det| < 1e-6 rejects (triangle.ml:83): zero-area triangles and rays in the triangle's plane never hit."""
    from path_tracer_ocaml_amd import abi
    a, b, c = np.array([-1.0, -1.0, -4.0]), np.array([1.0, -1.0, -4.0]), np.array([0.0, 1.0, -4.0])
    tris = [(a, b, c, 0), (a, a, b, 1), (a, b, (a + b) / 2, 1), (a + [0, 0, -1], b + [0, 0, -1], c + [0, 0, -1], 2)]
    d, keep = make_desc(abi, tris=tris, leaf_kind=1, cutoff=2)
    o_scene, g_scene = both(P, oracle, d, keep)
    rng = np.random.default_rng(3)
    dirs = np.concatenate([rng.normal(size=(3000, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1], [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    origins = np.concatenate([np.zeros((3000, 3)), [[-5.0, 0.0, -4.0], [0.0, -5.0, -4.0]]])  # two rays IN the plane z = -4
    prims = check_rays(o_scene, g_scene, origins, dirs)
    assert set(prims.tolist()) <= {-1, 0, 3}
    assert prims[-1] == -1 and prims[-2] == -1
    xs, ys, ps = rng.integers(0, 48, 3000), rng.integers(0, 24, 3000), rng.integers(0, 3, 3000)
    cc, _ = o_scene.trace_samples(48, 24, 3, 6, xs, ys, ps)
    gg, _ = g_scene.trace_samples(48, 24, 3, 6, xs, ys, ps)
    assert np.array_equal(bits(gg), bits(cc))


def test_render_params_validation(P, oracle):
    """check_params through the C ABI on a device scene: bad sizes, an out-of-range band, and unknown bits in `flags` (the
    field replaced `reserved`: a caller that left it uninitialised gets an error, not a silently queued frame)."""
    d = oracle.desc_shirley(32, 16)
    g = P.Scene(d.ptr, 0, keepalive=d)
    out = np.zeros((16, 32, 3))
    for bad, msg in (({"width": 0}, "dimensions"), ({"samples_per_pixel": 0}, "samples_per_pixel"), ({"max_bounces": 127}, "max_bounces"),
                     ({"band_step": 2, "band_first": 2}, "band_first"), ({"flags": 2}, "flags"), ({"flags": 0x40000001}, "flags")):
        p = P.render_params(32, 16, 2, 4)
        for k, v in bad.items():
            setattr(p, k, v)
        with pytest.raises(P.PtxError, match=msg):
            P._check(P.lib().ptx_render(g._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_double)), None, None, None))
    # PTX_RENDER_ASYNC handed to the host-framebuffer entry point is ignored, not honoured: the frame is complete on return
    p = P.render_params(32, 16, 2, 4, asynchronous=True)
    P._check(P.lib().ptx_render(g._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_double)), None, None, None))
    ref = oracle.Scene(d.ptr, d).render(32, 16, 2, 4)["rgb"]
    assert np.abs(out - ref).max() <= 1e-12
    g.close()
