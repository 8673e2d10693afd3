"""The host programs either side of the integrator, on the GPU, against the oracle:

* F2 (SURVEY section 8 f): a mesh that arrives as a PLY FILE -- `ganesha -ganesha-ply PATH` (ganesha/bin/main.ml:121-131,
  ply_format/src/ply.ml:340-352) -- rendered by the photon mapper the reference uses for it, and by the path integrator,
  compared with the oracle's scene built from the same mesh.
* the Python mirror of Render_command / Integrator (path_tracer_ocaml_amd/integrator.py): progress sums to W*H, the PNG
  equals the C++ CLI's byte for byte."""
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


def _env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return env


def test_ganesha_cli_renders_a_ply_file(P, oracle, tmp_path):
    """write a PLY -> `ganesha -ganesha-ply` (PLY reader, Mesh.create, floor, two spot lights, photon mapper on the GPU)
    -> PNG; against the ORACLE's photon mapper on the oracle's own build of the same mesh."""
    from PIL import Image
    from path_tracer_ocaml_amd import abi, host as H
    n_tri, w, h, iters, photons = 6000, 128, 72, 2, 20000
    ply = str(tmp_path / "mesh.ply")
    H.write_ganesha_like_ply(ply, n_tri, 7)
    out = str(tmp_path / "g.png")
    exe = os.path.join(ROOT, "path_tracer_ocaml_amd", "ganesha")
    res = subprocess.run([exe, "-ganesha-ply", ply, "-width", str(w), "-height", str(h), "-iterations", str(iters),
                          "-photon-count", str(photons), "-o", out], capture_output=True, text=True, env=_env(), timeout=600)
    assert res.returncode == 0, res.stderr
    for needle in ("dim = 128 x 72;", "#triangles = ", "tree depth = ", "build time = ", "#iteration = 1, radius = ", "elapsed ms: "):
        assert needle in res.stdout, res.stdout
    d = oracle.desc_ganesha_like(w, h, n_tri, 7)
    d.d.background.kind = abi.PTX_BG_BLACK
    o = oracle.Scene(d.ptr, d)
    o_img, o_st = o.ppm_render(abi.ppm_params(w, h, iterations=iters, photon_count=photons), o.lights_ganesha())
    assert o_st["neighbors"] > 1000 and o_img.max() > 0
    want = np.clip(H.ppm_gamma(o_img, iters) * 255.0, 0, 255).astype(np.uint8)
    got = np.array(Image.open(out).convert("RGB"))
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {want.size} bytes differ from the oracle's image"
    # -stop-after-bvh (ganesha/bin/main.ml:22-24,196-200)
    res = subprocess.run([exe, "-ganesha-ply", ply, "-width", "32", "-stop-after-bvh"], capture_output=True, text=True, env=_env(), timeout=300)
    assert res.returncode == 0 and "Stop after bvh build" in res.stdout and "#iteration" not in res.stdout


def test_ply_loaded_mesh_under_the_path_integrator(P, oracle, tmp_path):
    """BASELINE config 4's scene arriving through the PLY reader: per-sample radiance and counters equal the oracle's."""
    from path_tracer_ocaml_amd import host as H
    n_tri, w, h, spp, depth = 20000, 160, 90, 4, 8
    ply = str(tmp_path / "mesh.ply")
    H.write_ganesha_like_ply(ply, n_tri, 7)
    hs = H.ganesha_ply(ply, w, h)
    g = P.Scene(hs.ptr, 0, keepalive=hs)
    d = oracle.desc_ganesha_like(w, h, n_tri, 7)
    o = oracle.Scene(d.ptr, d)
    rng = np.random.default_rng(11)
    n = 8000
    xs, ys, ps = rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, spp, n)
    c, ct = o.trace_samples(w, h, spp, depth, xs, ys, ps)
    got, st = g.trace_samples(w, h, spp, depth, xs, ys, ps, count_work=True)
    assert np.array_equal(bits(got), bits(c))
    for k in ("segments", "nodes_tested", "prims_tested", "floor_tested"):
        assert st[k] == ct[k], k
    full, _ = g.render(w, h, spp, depth)
    ref = o.render(w, h, spp, depth, threads=4)["rgb"]
    assert float((np.abs(full - ref) / np.maximum(np.abs(ref), 1e-3)).max()) <= 1e-12


def test_python_integrator_mirror(P, tmp_path):
    """Integrator.create / render and Render_command.run of path_tracer_ocaml_amd/integrator.py: update_progress gets
    pixel areas summing to W*H on the calling thread (integrator.ml:150, render_command.ml:87-103) and the PNG it saves
    is the C++ CLI's, byte for byte."""
    from path_tracer_ocaml_amd import host as H, integrator as I
    ns = I.args_term().parse_args(["--dimension=200,100", "--samples-per-pixel=6", "--max-ray-bounces=5", "-o", str(tmp_path / "py.png")])
    args = I.args_of_namespace(ns)
    assert (args.width, args.height, args.samples_per_pixel, args.max_bounces, args.no_progress) == (200, 100, 6, 5, False)
    hs = H.shirley_spheres(args.width, args.height)
    image = np.zeros((args.height, args.width, 3))
    integ = I.Integrator.create(width=args.width, height=args.height, image=image, samples_per_pixel=args.samples_per_pixel,
                                max_bounces=args.max_bounces, scene=hs)
    me, seen = threading.get_ident(), []

    def update_progress(n):
        assert threading.get_ident() == me
        seen.append(n)

    assert integ.render(update_progress) is image
    assert sum(seen) == args.width * args.height and image.max() > 0.5
    lines = []
    image2, st = I.run(args, hs, echo=lines.append)
    assert np.array_equal(bits(image2), bits(image))
    assert len(lines) == 1 and lines[0].startswith("rendered in: ") and lines[0].endswith(" ms")
    exe = os.path.join(ROOT, "path_tracer_ocaml_amd", "shirley_spheres")
    cli_png = tmp_path / "cli.png"
    r = subprocess.run([exe, "--dimension=200,100", "--samples-per-pixel=6", "--max-ray-bounces=5", "--no-progress", "-o", str(cli_png)],
                       capture_output=True, text=True, env=_env())
    assert r.returncode == 0, r.stderr
    assert cli_png.read_bytes() == (tmp_path / "py.png").read_bytes()
    with pytest.raises(ValueError):
        I.Integrator.create(width=8, height=8, image=np.zeros((8, 8, 3), dtype=np.float32), samples_per_pixel=1, max_bounces=1, scene=hs)


@pytest.mark.parametrize("slabs", [1, 2, 3, 8])
def test_ptx_render_pipelines_the_last_accumulate_into_a_pinned_image(P, oracle, slabs, monkeypatch):
    """ptx_render into a pinned image: the frame's last accumulate runs in PTX_FINAL_SLABS row slabs, and slab k is filmed and copied
    to the host on a second stream while the later slabs are still being summed (the film reads one row beyond its slab, so it
    waits for slab k + 1).  Ragged slab heights, one and several batches: the framebuffer is the plain path's bit for bit."""
    monkeypatch.setenv("PTX_FINAL_SLABS", str(slabs))
    w, h, depth = 700, 515, 5
    d = oracle.desc_shirley(w, h)
    g = P.Scene(d.ptr, 0, keepalive=d)
    img = np.full((h, w, 3), -1.0)
    for spp, ppb in ((3, 0), (7, 2), (4, 4)):
        ref, _ = g.render(w, h, spp, depth, passes_per_batch=ppb)  # unpinned: staged copy, one film launch
        g.pin_image(img)
        img[:] = -1.0
        g.render(w, h, spp, depth, out=img, passes_per_batch=ppb)
        g.unpin_image()
        assert np.array_equal(img.view(np.uint64), ref.view(np.uint64)), (spp, ppb)
    g.close()


def test_ptx_render_into_pinned_and_unpinned_images(P, oracle):
    """ptx_render's way back to the host: a staged copy into any image, one DMA into an image the caller has pinned
    (ptx_image_pin: the caller promises to keep it mapped until ptx_image_unpin / ptx_scene_destroy -- the library never pins
    behind the caller's back, a DMA into a stale registration aborts the process).  Same framebuffer, bit for bit, on every path:
    unpinned, pinned, a sub-range of the pinned image, another image while one is pinned, after unpin, after re-pinning another."""
    w, h, spp, depth = 768, 512, 2, 4  # 9.4 MB: above the size below which a plain copy is used
    d = oracle.desc_shirley(w, h)
    g = P.Scene(d.ptr, 0, keepalive=d)
    ref, _ = g.render(w, h, spp, depth)
    img = np.full((h, w, 3), -1.0)
    g.render(w, h, spp, depth, out=img)
    assert np.array_equal(img.view(np.uint64), ref.view(np.uint64))
    big = np.full((2, h, w, 3), -1.0)
    g.pin_image(big)
    for k in range(2):  # both halves of the pinned allocation
        g.render(w, h, spp, depth, out=big[k])
        assert np.array_equal(big[k].view(np.uint64), ref.view(np.uint64)), f"pinned image, half {k}"
    other = np.full((h, w, 3), -1.0)
    g.render(w, h, spp, depth, out=other)  # not the pinned one: staged
    assert np.array_equal(other.view(np.uint64), ref.view(np.uint64))
    g.pin_image(other)  # replaces the first registration
    g.render(w, h, spp, depth, out=other)
    assert np.array_equal(other.view(np.uint64), ref.view(np.uint64))
    big[:] = -3.0
    g.render(w, h, spp, depth, out=big[0])  # no longer pinned: staged again
    assert np.array_equal(big[0].view(np.uint64), ref.view(np.uint64))
    g.unpin_image()
    g.unpin_image()  # idempotent
    g.render(w, h, spp, depth, out=other)
    assert np.array_equal(other.view(np.uint64), ref.view(np.uint64))
    g.pin_image(img)
    g.close()  # destroy releases the registration
    img[:] = 0.0
