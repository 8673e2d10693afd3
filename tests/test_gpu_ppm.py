"""Progressive photon mapping on the GPU against the oracle's restatement of
progressive-photon-map/src/progressive_photon_map.ml.  No fixture of the reference pins this integrator
(parity unpinned); the bar here is GPU == oracle: photon counts and ray counts exact, neighbour counts exact,
img_sum bit-exact (the photon list order, the photon tree and the summation order are all reproduced)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import path_tracer_ocaml_amd as P
    assert P.lib().ptx_device_count() >= 1, P.last_error()
    return P


def _check(P, oracle, desc, lights, params):
    o = oracle.Scene(desc.ptr, desc)
    g = P.Scene(desc.ptr, 0, keepalive=desc)
    o_img, o_st = o.ppm_render(params, lights)
    g_img, g_st = g.ppm_render(params, lights)
    for k in ("photons_stored", "photon_rays", "eye_rays", "neighbors"):
        assert g_st[k] == o_st[k], (k, g_st[k], o_st[k])
    assert g_st["last_radius"] == o_st["last_radius"]
    assert o_img.max() > 0
    nbad = int((g_img.view(np.uint64) != o_img.view(np.uint64)).sum())
    assert nbad == 0, f"{nbad} of {o_img.size} img_sum values differ from the oracle"
    g.close()
    return o_img, o_st


def test_ppm_cornell_point_light(P, oracle):
    from path_tracer_ocaml_amd import abi
    w = h = 160
    d = oracle.desc_cornell(w, h, 0.0)  # the reference's scene: no emitter, a point light
    img, st = _check(P, oracle, d, oracle.lights_cornell(w, h), abi.ppm_params(w, h, iterations=3, photon_count=20000))
    assert st["photons_stored"] > 50000 and st["neighbors"] > 1_000_000


def test_ppm_host_side_list_and_tree(P, oracle, monkeypatch):
    """PTX_PPM_HOST_LIST=1: the photon list and the photon tree are made on the host (the path small maps take)."""
    from path_tracer_ocaml_amd import abi
    monkeypatch.setenv("PTX_PPM_HOST_LIST", "1")
    w = h = 96
    d = oracle.desc_cornell(w, h, 0.0)
    _check(P, oracle, d, oracle.lights_cornell(w, h), abi.ppm_params(w, h, iterations=2, photon_count=12000))


def test_ppm_small_map_below_gpu_build_threshold(P, oracle):
    from path_tracer_ocaml_amd import abi
    w = h = 64
    d = oracle.desc_cornell(w, h, 0.0)
    _, st = _check(P, oracle, d, oracle.lights_cornell(w, h), abi.ppm_params(w, h, iterations=2, photon_count=2000))
    assert st["photons_stored"] < 2 * 8192


def test_ppm_ganesha_like_two_spot_lights(P, oracle):
    from path_tracer_ocaml_amd import abi
    w, h = 160, 90
    d = oracle.desc_ganesha_like(w, h, 20000)
    d.d.background.kind = abi.PTX_BG_BLACK
    lights = oracle.Scene(d.ptr, d).lights_ganesha()
    _check(P, oracle, d, lights, abi.ppm_params(w, h, iterations=2, photon_count=30000, max_bounces=4))


def test_ppm_shirley_simd_leaf_one_bounce_deep(P, oracle):
    """The photon pass on the Simd_leaf scene, deeper paths, alpha at its default."""
    from path_tracer_ocaml_amd import abi
    w, h = 120, 60
    d = oracle.desc_shirley(w, h)
    light = abi.Light()
    light.kind = abi.PTX_LIGHT_POINT
    light.position[:] = [0.0, 6.0, -12.0]
    light.color[:] = [1.0, 0.9, 0.8]
    light.power = 50.0
    _check(P, oracle, d, [light], abi.ppm_params(w, h, iterations=2, photon_count=15000, max_bounces=8))


def test_ppm_rejects_bad_arguments(P, oracle):
    from path_tracer_ocaml_amd import abi
    d = oracle.desc_cornell(16, 16, 0.0)
    g = P.Scene(d.ptr, 0, keepalive=d)
    with pytest.raises(P.PtxError):
        g.ppm_render(abi.ppm_params(16, 16, iterations=0), oracle.lights_cornell(16, 16))
    bad = abi.Light()
    bad.kind = 7
    with pytest.raises(P.PtxError):
        g.ppm_render(abi.ppm_params(16, 16, iterations=1, photon_count=100), [bad])
    g.close()


def test_cornell_box_cli(P, tmp_path):
    """cornell-box's Stdlib.Arg command line and prints, PNG rewritten after every iteration."""
    import os
    import subprocess
    from PIL import Image
    from path_tracer_ocaml_amd import abi, host as H
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "path_tracer_ocaml_amd", "cornell_box")
    out = str(tmp_path / "c.png")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([exe, "-width", "96", "-iterations", "2", "-photon-count", "8000", "-o", out], capture_output=True,
                         text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr
    for needle in ("#max-bounces = 4", "#photons/iter = 8000", "#iterations = 2", "-----", "#iteration = 1, radius = ",
                   "  photon map length = ", "render time = "):
        assert needle in res.stdout, res.stdout
    hs = H.cornell_box(96, 96, 0.0)
    hs.d.background.kind = abi.PTX_BG_BLACK
    img, _ = P.Scene(hs.ptr, 0, keepalive=hs).ppm_render(abi.ppm_params(96, 96, iterations=2, photon_count=8000), H.lights_cornell(96, 96))
    want = np.clip(H.ppm_gamma(img, 2) * 255.0, 0, 255).astype(np.uint8)
    got = np.array(Image.open(out).convert("RGB"))
    assert np.array_equal(got, want)
    assert subprocess.run([exe, "-bogus"], capture_output=True, env=env).returncode == 2
