"""BASELINE.json's full-size configurations on the GPU, checked through size-independent properties
(the oracle needs minutes per frame at these sizes):

* determinism: two renders give bit-identical raw sums;
* partition invariance: whole image == interleaved row bands of 2 / 5 ranks == any batch size (the sampler
  offset depends only on the global (x, y, pass), integrator.ml:98);
* a random subset of the frame's samples re-traced one by one is bit-identical to the oracle, and their sum
  per pixel (in pass order) equals the frame's raw sum for fully covered pixels;
* work counters are partition-invariant and match the oracle's on the sampled subset.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import path_tracer_ocaml_amd as P
    assert P.lib().ptx_device_count() >= 1, P.last_error()
    return P


def _raw(P, torch, scene, w, h, spp, depth, **kw):
    params = P.render_params(w, h, spp, depth, **kw)
    rows = P.local_rows(params)
    t = torch.zeros((rows, w, 3), dtype=torch.float64, device="cuda:0")
    st = scene.render_raw_device(params, t.data_ptr())
    return t, st, params


def _assemble(P, torch, scene, w, h, spp, depth, world, **kw):
    full = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    tot = {"segments": 0, "nodes_tested": 0, "prims_tested": 0, "floor_tested": 0}
    for rank in range(world):
        part, st, params = _raw(P, torch, scene, w, h, spp, depth, band_rows=32, band_first=rank, band_step=world, **kw)
        idx = torch.as_tensor([P.global_row(params, k) for k in range(part.shape[0])], device="cuda:0")
        full[idx] = part
        for k in tot:
            tot[k] += st[k]
    return full, tot


def _frame_pixels_against_the_oracle(oracle, od, scene, raw, w, h, spp, depth, seed, n_px=24):
    """Whole pixels of the FRAME against the oracle: every pass of n_px pixels traced one by one on the CPU, summed in pass order
    like render_tile's pass loop (integrator.ml:91-112), compared bit for bit with the frame's raw sums -- the frame itself
    (all its batches, both streams, 10^8 paths), not the list-mode entry point."""
    rng = np.random.default_rng(seed)
    px, py = rng.integers(0, w, n_px), rng.integers(0, h, n_px)
    xs, ys, ps = np.repeat(px, spp), np.repeat(py, spp), np.tile(np.arange(spp), n_px)
    o_rgb, _ = oracle.Scene(od.ptr, od).trace_samples(w, h, spp, depth, xs, ys, ps)
    per = o_rgb.reshape(n_px, spp, 3)
    sums = np.zeros((n_px, 3))
    for k in range(spp):
        sums = sums + per[:, k, :]
    got = raw.cpu().numpy()[py, px]
    assert np.array_equal(sums.view(np.uint64), got.view(np.uint64)), "frame raw sums != oracle per-sample sums in pass order"
    assert float(np.abs(sums).max()) > 0.0


def test_config2_shirley_1080p_spp64(P, oracle):
    torch = pytest.importorskip("torch")
    from path_tracer_ocaml_amd import host as H
    w, h, spp, depth = 1920, 1080, 64, 8
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    a, st_a, _ = _raw(P, torch, scene, w, h, spp, depth, count_work=True)
    b, _, _ = _raw(P, torch, scene, w, h, spp, depth)
    assert torch.equal(a.view(torch.int64), b.view(torch.int64)), "two renders differ"
    c, _, _ = _raw(P, torch, scene, w, h, spp, depth, passes_per_batch=5)  # 13 ragged batches
    assert torch.equal(a.view(torch.int64), c.view(torch.int64)), "batch size changed the result"
    d2, tot2 = _assemble(P, torch, scene, w, h, spp, depth, 2, count_work=True)
    assert torch.equal(a.view(torch.int64), d2.view(torch.int64)), "2-rank band sharding changed the result"
    d5, _ = _assemble(P, torch, scene, w, h, spp, depth, 5)
    assert torch.equal(a.view(torch.int64), d5.view(torch.int64)), "5-rank band sharding changed the result"
    for k in ("segments", "nodes_tested", "prims_tested"):
        assert tot2[k] == st_a[k], k
    assert st_a["samples"] == w * h * spp
    # oracle on a sample of the frame's own samples: whole pixels (all 64 passes) so sums can be compared
    rng = np.random.default_rng(5)
    px = rng.integers(0, w, 300)
    py = rng.integers(0, h, 300)
    xs = np.repeat(px, spp)
    ys = np.repeat(py, spp)
    ps = np.tile(np.arange(spp), 300)
    od = oracle.desc_shirley(w, h)
    o_rgb, o_ct = oracle.Scene(od.ptr, od).trace_samples(w, h, spp, depth, xs, ys, ps)
    g_rgb, g_st = scene.trace_samples(w, h, spp, depth, xs, ys, ps, count_work=True)
    assert np.array_equal(g_rgb.view(np.uint64), o_rgb.view(np.uint64))
    for k in ("segments", "nodes_tested", "prims_tested"):
        assert g_st[k] == o_ct[k]
    raw = a.cpu().numpy()
    sums = np.zeros((300, 3))
    per = o_rgb.reshape(300, spp, 3)
    for k in range(spp):  # pass order, like render_tile's pass loop
        sums = sums + per[:, k, :]
    assert np.array_equal(sums.view(np.uint64), raw[py, px].view(np.uint64)), "frame raw sums != oracle per-sample sums"
    scene.close()


def test_config3_cornell_1024_spp256_depth16(P, oracle):
    torch = pytest.importorskip("torch")
    from path_tracer_ocaml_amd import host as H
    w, h, spp, depth = 1024, 1024, 256, 16
    hs = H.cornell_box(w, h, 12.0)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    a, _, _ = _raw(P, torch, scene, w, h, spp, depth)
    d3, _ = _assemble(P, torch, scene, w, h, spp, depth, 3)
    assert torch.equal(a.view(torch.int64), d3.view(torch.int64))
    assert float(a.max()) > 0.0 and bool(torch.isfinite(a).all())
    rng = np.random.default_rng(6)
    n = 30000
    xs, ys, ps = rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, spp, n)
    od = oracle.desc_cornell(w, h, 12.0)
    o_rgb, o_ct = oracle.Scene(od.ptr, od).trace_samples(w, h, spp, depth, xs, ys, ps)
    g_rgb, g_st = scene.trace_samples(w, h, spp, depth, xs, ys, ps, count_work=True)
    assert np.array_equal(g_rgb.view(np.uint64), o_rgb.view(np.uint64))
    assert g_st["nodes_tested"] == o_ct["nodes_tested"] and g_st["prims_tested"] == o_ct["prims_tested"]
    _frame_pixels_against_the_oracle(oracle, od, scene, a, w, h, spp, depth, seed=16)  # 24 x 256 samples at depth 16
    scene.close()


def test_config4_ganesha_like_150k_triangles(P, oracle):
    torch = pytest.importorskip("torch")
    from path_tracer_ocaml_amd import host as H
    w, h, spp, depth = 1920, 1080, 64, 8
    hs = H.ganesha_like(w, h, 150000, 7)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    assert scene.stats()["tree_nodes"] > 80000
    a, _, _ = _raw(P, torch, scene, w, h, spp, depth)
    d4, _ = _assemble(P, torch, scene, w, h, spp, depth, 4)
    assert torch.equal(a.view(torch.int64), d4.view(torch.int64))
    rng = np.random.default_rng(7)
    n = 30000
    xs, ys, ps = rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, spp, n)
    od = oracle.desc_ganesha_like(w, h, 150000, 7)
    o_rgb, o_ct = oracle.Scene(od.ptr, od).trace_samples(w, h, spp, depth, xs, ys, ps)
    g_rgb, g_st = scene.trace_samples(w, h, spp, depth, xs, ys, ps, count_work=True)
    assert np.array_equal(g_rgb.view(np.uint64), o_rgb.view(np.uint64))
    for k in ("segments", "nodes_tested", "prims_tested", "floor_tested"):
        assert g_st[k] == o_ct[k], k
    _frame_pixels_against_the_oracle(oracle, od, scene, a, w, h, spp, depth, seed=17, n_px=48)
    scene.close()


def test_config5_shirley_4k_rows_of_one_rank_of_eight(P, oracle):
    """3840x2160 spp=256 is 2.1 G samples; one rank of an 8-rank job renders 1/8 of the rows.  Check that rank's
    rows against the oracle on sampled pixels and against a differently batched render."""
    torch = pytest.importorskip("torch")
    from path_tracer_ocaml_amd import host as H
    w, h, spp, depth = 3840, 2160, 256, 8
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    part, st, params = _raw(P, torch, scene, w, h, spp, depth, band_rows=32, band_first=3, band_step=8)
    rows = part.shape[0]
    assert st["samples"] == rows * w * spp
    part2, _, _ = _raw(P, torch, scene, w, h, spp, depth, band_rows=32, band_first=3, band_step=8, passes_per_batch=7)
    assert torch.equal(part.view(torch.int64), part2.view(torch.int64))
    rng = np.random.default_rng(8)
    lr = rng.integers(0, rows, 40)
    px = rng.integers(0, w, 40)
    gy = np.array([P.global_row(params, int(k)) for k in lr])
    xs, ys, ps = np.repeat(px, spp), np.repeat(gy, spp), np.tile(np.arange(spp), 40)
    od = oracle.desc_shirley(w, h)
    o_rgb, _ = oracle.Scene(od.ptr, od).trace_samples(w, h, spp, depth, xs, ys, ps)
    per = o_rgb.reshape(40, spp, 3)
    sums = np.zeros((40, 3))
    for k in range(spp):
        sums = sums + per[:, k, :]
    got = part.cpu().numpy()[lr, px]
    assert np.array_equal(sums.view(np.uint64), got.view(np.uint64))
    scene.close()


def test_config5_shirley_4k_all_eight_band_shares(P, oracle):
    """BASELINE configs[4] in full: all 8 ranks' shares of 3840x2160 spp=256 rendered one after the other on this GPU
    into the gathered [rank][pad_rows][W][3] layout exactly as bench.py --gpus 8 fills it (8-row bands), resolved by
    the banded film kernel, and compared bit for bit with the one-rank render of the whole frame; sampled pixels
    against the oracle.  Only the transport (RCCL sends) is not exercised here."""
    torch = pytest.importorskip("torch")
    from path_tracer_ocaml_amd import host as H
    from path_tracer_ocaml_amd import distributed as D
    w, h, spp, depth, world = 3840, 2160, 256, 8, 8
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    pad = D.max_local_rows(h, world)
    gathered = torch.zeros((world, pad, w, 3), dtype=torch.float64, device="cuda:0")
    samples = 0
    for rank in range(world):
        params = P.render_params(w, h, spp, depth, band_rows=D.BAND_ROWS, band_first=rank, band_step=world)
        assert P.local_rows(params) <= pad
        samples += scene.render_raw_device(params, gathered[rank].data_ptr())["samples"]
    assert samples == w * h * spp
    rgb8 = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    P.film_resolve_banded_device(0, w, h, spp, gathered.data_ptr(), world, D.BAND_ROWS, pad, rgb8.data_ptr())
    whole, _, _ = _raw(P, torch, scene, w, h, spp, depth)
    rgb1 = torch.zeros_like(rgb8)
    P.film_resolve_device(0, w, h, spp, whole.data_ptr(), rgb1.data_ptr())
    idx = torch.as_tensor(D.band_row_index(np.arange(h), world, D.BAND_ROWS, pad), device="cuda:0")
    assert torch.equal(gathered.view(world * pad, w, 3)[idx].view(torch.int64), whole.view(torch.int64)), "raw sums of the 8 shares != one-rank frame"
    assert torch.equal(rgb8.view(torch.int64), rgb1.view(torch.int64)), "banded film != film of the whole frame"
    assert bool(torch.isfinite(rgb8).all()) and float(rgb8.max()) > 0.5
    rng = np.random.default_rng(9)
    npx = 48
    px, py = rng.integers(0, w, npx), rng.integers(0, h, npx)
    xs, ys, ps = np.repeat(px, spp), np.repeat(py, spp), np.tile(np.arange(spp), npx)
    od = oracle.desc_shirley(w, h)
    o_rgb, _ = oracle.Scene(od.ptr, od).trace_samples(w, h, spp, depth, xs, ys, ps)
    per = o_rgb.reshape(npx, spp, 3)
    sums = np.zeros((npx, 3))
    for k in range(spp):
        sums = sums + per[:, k, :]
    got = whole.cpu().numpy()[py, px]
    assert np.array_equal(sums.view(np.uint64), got.view(np.uint64))
    scene.close()
