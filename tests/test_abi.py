"""The C-ABI libraries load on a machine with no GPU and export every symbol their headers declare; the
ctypes mirrors match the C layouts; compute entry points refuse to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    import path_tracer_ocaml_amd as P
    L = P.lib()
    names = _declared("include/ptx.h", "ptx_")
    assert len(names) >= 17
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ptx.h but not exported by libptx_hip.so"
    assert sorted(P.EXPORTS) == sorted(n for n in names if n != "ptx_progress_fn")


def test_host_library_exports():
    from path_tracer_ocaml_amd import host as H
    L = H.lib()
    for n in _declared("path_tracer_ocaml_amd/host/host.h", "pth_"):
        assert hasattr(L, n), n


def test_struct_layouts_match_c(tmp_path):
    """sizeof / offsetof of every ABI struct, as the C compiler sees them, against the ctypes mirror."""
    from path_tracer_ocaml_amd import abi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ptx.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ptx_material), sizeof(ptx_texture),'
                   'sizeof(ptx_camera), sizeof(ptx_background), sizeof(ptx_scene_desc), sizeof(ptx_render_params),'
                   'sizeof(ptx_stats), offsetof(ptx_scene_desc, camera), offsetof(ptx_scene_desc, leaf_kind),'
                   'offsetof(ptx_stats, kernel_ms), offsetof(ptx_render_params, n_gpus), offsetof(ptx_stats, filter_undecided),'
                   'offsetof(ptx_stats, staged_copies));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.Material), C.sizeof(abi.Texture), C.sizeof(abi.Camera), C.sizeof(abi.Background),
            C.sizeof(abi.SceneDesc), C.sizeof(abi.RenderParams), C.sizeof(abi.Stats), abi.SceneDesc.camera.offset,
            abi.SceneDesc.leaf_kind.offset, abi.Stats.kernel_ms.offset, abi.RenderParams.n_gpus.offset,
            abi.Stats.filter_undecided.offset, abi.Stats.staged_copies.offset]
    assert got == want


def test_version_and_leaf_size():
    import path_tracer_ocaml_amd as P
    L = P.lib()
    assert L.ptx_version() == 6  # 6: ptx_stats.solo_launches; 5: ptx_image_pin / ptx_image_unpin, ptx_render_params.flags validated (unknown bits are an error); 4: PTX_KERNEL_BOUNCE (ptx_stats.kernel_ms / kernel_launches have 6 entries); 3: ptx_stats.filter_* / peer_copies / staged_copies; 2: n_gpus, ptx_render_multi, ptx_scene_replicate, banded film
    from path_tracer_ocaml_amd import abi
    assert abi.PTX_ABI_VERSION == 4
    assert L.ptx_leaf_size() == 16  # LEAF_SIZE, sphere-intersect-rs/src/lib.rs:13


def test_band_row_mapping_partitions_the_image():
    """ptx_local_rows / ptx_global_row: interleaved 32-row bands, every row owned by exactly one rank."""
    import path_tracer_ocaml_amd as P
    for height in (1, 31, 32, 33, 300, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(height, dtype=int)
            for rank in range(world):
                p = P.render_params(64, height, 1, 1, band_rows=32, band_first=rank, band_step=world)
                rows = P.local_rows(p)
                for k in range(rows):
                    seen[P.global_row(p, k)] += 1
                assert P.global_row(p, rows) == -1
            assert (seen == 1).all(), (height, world)


def test_no_cpu_fallback_without_a_device(oracle):
    """On a box with no GPU every compute entry point must FAIL, never compute on the host."""
    import path_tracer_ocaml_amd as P
    if P.lib().ptx_device_count() > 0:
        pytest.skip("a HIP device is present")
    d = oracle.desc_shirley(16, 16)
    with pytest.raises(P.PtxError, match="no CPU fallback"):
        P.Scene(d.ptr, 0, keepalive=d)
    host_only = P.Scene(d.ptr, -1, keepalive=d)
    with pytest.raises(P.PtxError, match="no CPU fallback"):
        host_only.render(16, 16, 1, 1)
    with pytest.raises(P.PtxError, match="no CPU fallback"):
        P.render_multi([host_only], 16, 16, 1, 1)
    with pytest.raises(P.PtxError, match="cannot be replicated"):
        host_only.replicate(0)
    with pytest.raises(P.PtxError):
        P.math_eval("sqrt", np.ones(4))


def test_scene_validation_errors(oracle):
    import path_tracer_ocaml_amd as P
    from path_tracer_ocaml_amd import abi
    d = oracle.desc_shirley(16, 16)
    bad = abi.SceneDesc()
    C.memmove(C.byref(bad), d.ptr, C.sizeof(bad))
    bad.n_spheres = 0
    with pytest.raises(P.PtxError, match="non-empty"):  # Shape_tree.create: expected non-empty list of shapes
        P.Scene(bad, -1, keepalive=d)
    C.memmove(C.byref(bad), d.ptr, C.sizeof(bad))
    bad.num_bins = 3
    with pytest.raises(P.PtxError, match="num_bins"):  # assert (num_bins >= 4), shape_tree.ml:253
        P.Scene(bad, -1, keepalive=d)
    C.memmove(C.byref(bad), d.ptr, C.sizeof(bad))
    bad.length_cutoff = 32
    with pytest.raises(P.PtxError, match="leaf_size"):
        P.Scene(bad, -1, keepalive=d)
