"""The in-library multi-GPU path (ptx_render_multi / ptx_render n_gpus / ptx_scene_replicate / the banded film) on
ONE GPU: the replicas share device 0, which exercises everything except the peer-to-peer copy itself -- one host
thread per replica, the interleaved band deal, the gathered [rank][pad_rows][W][3] layout, the film kernel's row
map, progress on the calling thread, error propagation from worker threads.  Hardware N > 1 is the driver's to run.

Bar: bit-identical post-gamma framebuffers (the raw sums are partition-invariant and the film arithmetic per pixel
is the same), integrator.ml:98,114-128,152-154."""
import os
import subprocess
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def P():
    import path_tracer_ocaml_amd as P
    assert P.lib().ptx_device_count() >= 1, P.last_error()
    return P


@pytest.mark.parametrize("kind,w,h,spp,depth,n", [("shirley", 200, 150, 8, 8, 3), ("shirley", 333, 97, 5, 4, 8),
                                                   ("cornell", 96, 96, 16, 16, 2), ("ganesha", 160, 90, 4, 8, 4)])
def test_render_multi_equals_single(P, kind, w, h, spp, depth, n):
    from path_tracer_ocaml_amd import host as H
    hs = {"shirley": lambda: H.shirley_spheres(w, h), "cornell": lambda: H.cornell_box(w, h, 12.0),
          "ganesha": lambda: H.ganesha_like(w, h, 6000, 7)}[kind]()
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    single, st1 = scene.render(w, h, spp, depth, count_work=True)
    reps = [scene] + [scene.replicate(0) for _ in range(n - 1)]
    assert reps[1].stats()["tree_nodes"] == scene.stats()["tree_nodes"]
    multi, stn = P.render_multi(reps, w, h, spp, depth, count_work=True)
    assert np.array_equal(bits(single), bits(multi)), f"{kind}: {n}-replica render differs from the one-GPU render"
    for k in ("samples", "segments", "nodes_tested", "prims_tested", "floor_tested"):
        assert stn[k] == st1[k], k
    one, _ = P.render_multi([scene], w, h, spp, depth)  # n = 1 is ptx_render
    assert np.array_equal(bits(single), bits(one))
    multi32, _ = P.render_multi(reps, w, h, spp, depth, band_rows=32)  # a different band height, same image
    assert np.array_equal(bits(single), bits(multi32))
    for r in reps[1:]:
        r.close()
    scene.close()


def test_progress_runs_on_the_calling_thread_and_sums_to_the_image(P):
    from path_tracer_ocaml_amd import host as H
    w, h, spp, depth = 320, 200, 12, 8
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    quiet, _ = scene.render(w, h, spp, depth, passes_per_batch=3)
    me = threading.get_ident()
    seen = []

    def cb(n):
        assert threading.get_ident() == me
        seen.append(n)

    # ptx_render with a progress callback keeps both streams busy now (it used to fall back to one stream)
    loud, _ = scene.render(w, h, spp, depth, progress=cb, passes_per_batch=3)
    assert sum(seen) == w * h and len(seen) == 4 and all(n > 0 for n in seen)
    assert np.array_equal(bits(quiet), bits(loud))
    reps = [scene, scene.replicate(0), scene.replicate(0)]
    seen.clear()
    multi, _ = P.render_multi(reps, w, h, spp, depth, progress=cb, passes_per_batch=3)
    assert sum(seen) == w * h
    assert np.array_equal(bits(quiet), bits(multi))
    for r in reps[1:]:
        r.close()
    scene.close()


def test_n_gpus_parameter_and_its_errors(P, monkeypatch):
    from path_tracer_ocaml_amd import host as H
    w, h, spp, depth = 128, 64, 4, 8
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    single, _ = scene.render(w, h, spp, depth)
    ndev = P.lib().ptx_device_count()
    monkeypatch.delenv("PTX_MULTI_ALIAS", raising=False)
    with pytest.raises(P.PtxError, match="HIP device"):
        scene.render(w, h, spp, depth, n_gpus=ndev + 1)
    monkeypatch.setenv("PTX_MULTI_ALIAS", "1")  # test hook: replicas may share devices
    for n in (2, 3):
        got, st = scene.render(w, h, spp, depth, n_gpus=n)
        assert np.array_equal(bits(single), bits(got)), n
        assert st["samples"] == w * h * spp
    # a failing worker reports through the caller's ptx_last_error
    a, b = scene.replicate(0), scene.replicate(0)
    with pytest.raises(P.PtxError, match="same handle"):
        P.render_multi([a, a], w, h, spp, depth)
    with pytest.raises(P.PtxError, match="max_bounces"):
        P.render_multi([a, b], w, h, spp, 500)
    a.close()
    b.close()
    scene.close()


def test_cli_gpus_flag(P, tmp_path):
    exe = os.path.join(ROOT, "path_tracer_ocaml_amd", "shirley_spheres")
    outs = []
    for n in (1, 2):
        out = tmp_path / f"g{n}.png"
        env = dict(os.environ, PTX_MULTI_ALIAS="1")
        r = subprocess.run([exe, "--dimension=160,80", "--samples-per-pixel=4", f"--gpus={n}", "-o", str(out)],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert "rendered in:" in r.stdout
        outs.append(out.read_bytes())
    assert outs[0] == outs[1]
    r = subprocess.run([exe, "--dimension=0,80"], capture_output=True, text=True)
    assert r.returncode == 124 and "dimension" in r.stderr
    r = subprocess.run([exe, "--dimension=16,8", "--gpus=99"], capture_output=True, text=True)
    assert r.returncode == 1 and "HIP device" in r.stderr


def test_banded_film_reads_the_gathered_layout_in_place(P):
    """ptx_film_resolve_banded_device on [rank][pad][W][3] == ptx_film_resolve_device on the un-permuted image."""
    torch = pytest.importorskip("torch")
    from path_tracer_ocaml_amd import distributed as D
    rng = np.random.default_rng(3)
    for (w, h, world, band_rows) in [(64, 50, 3, 8), (33, 17, 8, 1), (128, 270, 8, 8), (16, 9, 2, 32)]:
        pad = D.max_local_rows(h, world, band_rows) + 1  # a spare row: pad_rows need not be tight
        img = rng.uniform(0.0, 50.0, (h, w, 3))
        gathered = np.full((world, pad, w, 3), np.nan)
        for r, rows in enumerate(D.band_layout(h, world, band_rows)):
            gathered[r, : len(rows)] = img[rows]
        assert np.array_equal(D.ungather(gathered, h, world, band_rows), img)
        d_g = torch.from_numpy(gathered).cuda()
        d_i = torch.from_numpy(img).cuda()
        a = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        b = torch.zeros_like(a)
        P.film_resolve_banded_device(0, w, h, 7, d_g.data_ptr(), world, band_rows, pad, a.data_ptr())
        P.film_resolve_device(0, w, h, 7, d_i.data_ptr(), b.data_ptr())
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)) and bool(torch.isfinite(a).all())
    with pytest.raises(P.PtxError, match="pad_rows"):
        P.film_resolve_banded_device(0, 8, 64, 1, d_g.data_ptr(), 2, 8, 8, a.data_ptr())


def test_replicas_on_one_device_count_neither_peer_nor_staged(P):
    from path_tracer_ocaml_amd import host as H
    w, h = 96, 64
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    reps = [scene, scene.replicate(0)]
    _, st = P.render_multi(reps, w, h, 2, 4)
    assert st["peer_copies"] == 0 and st["staged_copies"] == 0
    reps[1].close()
    scene.close()


def test_peer_access_between_two_devices(P):
    """The one exchange of the path on real hardware: a replica on device 1, its raw sums copied into the root's gathered
    layout with peer access enabled in both directions (xGMI).  Needs two GPUs: SKIPPED, not passed, on a one-GPU box."""
    from path_tracer_ocaml_amd import host as H
    if P.lib().ptx_device_count() < 2:
        pytest.skip("needs >= 2 HIP devices (the driver's multi-GPU node)")
    w, h, spp, depth = 320, 200, 8, 8
    hs = H.shirley_spheres(w, h)
    scene = P.Scene(hs.ptr, 0, keepalive=hs)
    single, st1 = scene.render(w, h, spp, depth, count_work=True)
    n = min(P.lib().ptx_device_count(), 8)
    reps = [scene] + [scene.replicate(k) for k in range(1, n)]
    multi, stn = P.render_multi(reps, w, h, spp, depth, count_work=True)
    assert np.array_equal(bits(single), bits(multi))
    assert stn["peer_copies"] + stn["staged_copies"] == n - 1
    for k in ("samples", "segments", "nodes_tested", "prims_tested"):
        assert stn[k] == st1[k], k
    again, _ = scene.render(w, h, spp, depth, n_gpus=n)  # the same through ptx_render's n_gpus
    assert np.array_equal(bits(single), bits(again))
    for r in reps[1:]:
        r.close()
    scene.close()
    if stn["peer_copies"] != n - 1:
        # the bytes are right either way (asserted above); which way they travelled is a property of the node
        pytest.xfail(f"peer access was granted for {stn['peer_copies']} of {n - 1} replicas on this node: the rest were staged through host memory")


@pytest.mark.parametrize("world", [2, 3])
def test_bench_multi_rank_line_on_one_gpu(world):
    """`bench.py --gpus N --backend gloo` with the N ranks sharing this one GPU (host-staged exchange): the one-process-per-GPU path
    end to end -- self-launched ranks, band deal, queued renders, the exchange into rank 0, banded film -- and the line it prints:
    whole pixels of the TIMED, GATHERED frame equal the oracle's to 1e-5 and come from every rank's bands, the collective block
    says what the library saw.  (On N GPUs the backend is RCCL; that run is the driver's.)"""
    import json
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--workload", "shirley_600x300_spp32_d8", "--cpu-seconds", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d["n_gpus"] == world and d["value"] > 0 and d["scaling"] == "strong"
    tf = d["parity"]["timed_frame"]
    assert tf["rel_linf_vs_cpu_ref"] is not None and tf["rel_linf_vs_cpu_ref"] <= 1e-5, tf
    assert tf["ranks_covered"] == list(range(world)), tf
    assert d["parity"]["rel_linf_vs_cpu_ref"] <= 1e-5
    col = d["collective"]
    assert col["world"] == world and len(col["render_ms_per_rank"]) == world and all(x > 0 for x in col["render_ms_per_rank"])
    assert col["rank0_gather_ms"] > 0 and col["rank0_film_ms"] > 0
    assert d["cpu_baseline"]["value"] > 0
