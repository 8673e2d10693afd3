"""The OCaml binding (bindings/ocaml/): the files exist, their stubs and externals agree, the patches apply to nothing
that is missing, and the argument marshalling the stub performs (ptx_ml_marshal.h) drives the real C ABI correctly.
OCaml itself is not in this image: ptx_stubs.c is not compiled here; everything it delegates to is."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "bindings", "ocaml")


def _flat_file(path, arr, leaf_kind=0, length_cutoff=16):
    """What Ptx.flatten (ptx.ml) makes of a scene description: materials and textures interned structurally in order of
    first use (spheres, then mesh faces, then floor triangles), the mesh and the floor passed through."""
    mat_rows, tex_rows, mat_ids, tex_ids = [], [], {}, {}

    def tex_id(t):
        row = tuple([1.0, t[1], t[2], *t[3:6], *t[6:9]] if int(t[0]) == 1 else [0.0, 0.0, 0.0, *t[3:6], 0.0, 0.0, 0.0])
        if row not in tex_ids:
            tex_ids[row] = len(tex_rows)
            tex_rows.append(row)
        return tex_ids[row]

    def mat_id(i):
        kind, tex, index, er, eg, eb = arr["materials"][i]
        key = (int(kind), tuple(arr["textures"][int(tex)]) if int(kind) != 2 else None, float(index) if int(kind) == 2 else 0.0, er, eg, eb)
        if key not in mat_ids:
            row = [2.0, 0.0, index, er, eg, eb] if int(kind) == 2 else [kind, float(tex_id(arr["textures"][int(tex)])), 0.0, er, eg, eb]
            mat_ids[key] = len(mat_rows)
            mat_rows.append(row)
        return mat_ids[key]

    sm = [mat_id(m) for m in arr["sphere_material"]]
    tm = [mat_id(m) for m in arr["tri_material"]]
    fm = [mat_id(m) for m in arr["floor_material"]]
    n, nv, nt, nf = len(sm), len(arr["vertex_x"]), len(tm), len(fm)
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", n, len(mat_rows), len(tex_rows), leaf_kind, length_cutoff, nv, nt, nf))
        for k in ("sphere_x", "sphere_y", "sphere_z", "sphere_r"):
            f.write(np.asarray(arr[k], dtype="<f8").tobytes())
        f.write(np.asarray(sm, dtype="<i4").tobytes())
        f.write(np.asarray(mat_rows, dtype="<f8").tobytes())
        f.write(np.asarray(tex_rows, dtype="<f8").tobytes())
        f.write(np.asarray(arr["camera"], dtype="<f8").tobytes())
        f.write(np.asarray(arr["background"], dtype="<f8").tobytes())
        for k in ("vertex_x", "vertex_y", "vertex_z"):
            f.write(np.asarray(arr[k], dtype="<f8").tobytes())
        f.write(np.asarray(arr["tri_indices"], dtype="<i4").tobytes())
        f.write(np.asarray(arr["tri_uv"], dtype="<f8").tobytes())
        f.write(np.asarray(tm, dtype="<i4").tobytes())
        f.write(np.asarray(arr["floor_vertices"], dtype="<f8").tobytes())
        f.write(np.asarray(arr["floor_uv"], dtype="<f8").tobytes())
        f.write(np.asarray(fm, dtype="<i4").tobytes())


def _ppm_file(path, params, lights):
    """params6 + lights11 as Ptx.ppm_render lays them out (ptx.ml)"""
    with open(path, "wb") as f:
        f.write(np.asarray([params.width, params.height, params.iterations, params.max_bounces, params.photon_count, params.alpha],
                           dtype="<f8").tobytes())
        f.write(struct.pack("<i", len(lights)))
        for l in lights:
            f.write(np.asarray([l.kind, *l.position, *l.direction, *l.color, l.power], dtype="<f8").tobytes())


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ocaml") / "driver")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", B,
                           os.path.join(ROOT, "tests", "c", "ocaml_binding_driver.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "path_tracer_ocaml_amd"), "-lptx_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "path_tracer_ocaml_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_files_and_symbols_agree():
    ml = open(os.path.join(B, "ptx.ml")).read()
    c = open(os.path.join(B, "ptx_stubs.c")).read()
    externals = set(re.findall(r'"(ptx_ml_[a-z_]+)"', ml))
    prims = set(re.findall(r"CAMLprim value (ptx_ml_[a-z_]+)\(", c))
    assert externals and externals <= prims, externals - prims
    # every C entry point the stub or its marshalling header calls is declared by include/ptx.h
    hdr = open(os.path.join(ROOT, "include", "ptx.h")).read()
    used = set(re.findall(r"\b(ptx_(?!ml_)[a-z_]+)\(", c + open(os.path.join(B, "ptx_ml_marshal.h")).read()))
    for name in used:
        assert re.search(r"\b" + name + r"\(", hdr), name
    # the record the stub indexes with Field(flat, i) has the field order it assumes
    fields = re.findall(r"^\s*[{;] (\w+) :", ml[ml.index("type flat ="):ml.index("type scene")], flags=re.M)
    assert fields == ["xs", "ys", "zs", "rs", "sphere_material", "materials", "textures", "camera", "background", "leaf_kind", "length_cutoff",
                      "vertex_x", "vertex_y", "vertex_z", "tri_indices", "tri_uv", "tri_material", "floor_vertices", "floor_uv", "floor_material"]
    for i, name in enumerate(fields):
        assert f"Field(flat, {i})" in c, name
    assert f"Field(flat, {len(fields)})" not in c
    # a stub never shares its name with a function of the marshalling header it includes (that was a hard compile error)
    marshal_fns = set(re.findall(r"^static \w[\w\s\*]*?\b(ptx_ml_[a-z_]+)\(", open(os.path.join(B, "ptx_ml_marshal.h")).read(), flags=re.M))
    assert marshal_fns and not (marshal_fns & prims), marshal_fns & prims
    assert "(foreign_stubs" in open(os.path.join(B, "dune")).read()
    for p in sorted(os.listdir(os.path.join(B, "patches"))):
        text = open(os.path.join(B, "patches", p)).read()
        assert text.startswith("--- a/") and "+++ b/" in text, p


def test_patches_apply_to_the_reference(tmp_path):
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("no reference checkout on this machine (the GPU box has none)")
    for p in sorted(os.listdir(os.path.join(B, "patches"))):
        text = open(os.path.join(B, "patches", p)).read()
        for rel in re.findall(r"^--- a/(\S+)", text, flags=re.M):
            dst = tmp_path / rel
            dst.parent.mkdir(parents=True, exist_ok=True)
            if not dst.exists():
                dst.write_bytes(open(os.path.join(ref, rel), "rb").read())
        subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(B, "patches", p)], cwd=tmp_path)
    assert "Make_gpu" in (tmp_path / "render_command/src/render_command.ml").read_text()
    assert "Make_gpu" in (tmp_path / "progressive-photon-map/src/progressive_photon_map.ml").read_text()
    for exe in ("cornell-box", "ganesha", "shirley_spheres"):
        text = (tmp_path / exe / "bin" / "main.ml").read_text()
        assert "gpus" in text and "Ptx.scene_create" in text, exe
        assert " ptx" in (tmp_path / exe / "bin" / "dune").read_text(), exe
    # only what the reference's interfaces export is used: Material.dielectric is not in material.mli
    for p in sorted(os.listdir(os.path.join(B, "patches"))):
        added = [l for l in open(os.path.join(B, "patches", p)).read().splitlines() if l.startswith("+")]
        assert not any("Material.dielectric" in l for l in added), p


@pytest.mark.parametrize("no_simd", [False, True])
def test_marshalling_builds_the_reference_tree(oracle, driver, tmp_path, no_simd):
    """Host-only scene (device -1) through ptx_ml_scene_create: same tree as the oracle's Shape_tree.create."""
    d = oracle.desc_shirley(600, 300, no_simd=no_simd)
    flat = str(tmp_path / "flat.bin")
    _flat_file(flat, d.arrays(), leaf_kind=1 if no_simd else 0, length_cutoff=4 if no_simd else 16)
    out = subprocess.check_output([driver, flat, "tree"], text=True)
    got = dict(zip(out.split()[0::2], map(int, out.split()[1::2])))
    info = oracle.Scene(d.ptr, d).info()
    assert got["leaf_size"] == 16
    assert (got["nodes"], got["depth"], got["leaves"], got["slots"]) == (info["nodes"], info["depth"], info["leaves"], info["slots"])


@pytest.mark.gpu
def test_marshalling_renders_what_ptx_render_renders(oracle, driver, tmp_path):
    import path_tracer_ocaml_amd as P
    w, h, spp, depth = 200, 100, 8, 8
    d = oracle.desc_shirley(w, h)
    flat, outp = str(tmp_path / "flat.bin"), str(tmp_path / "out.bin")
    _flat_file(flat, d.arrays())
    out = subprocess.check_output([driver, flat, "render", str(w), str(h), str(spp), str(depth), "1", outp], text=True)
    assert f"progress_pixels {w * h}" in out
    got = np.fromfile(outp, dtype=np.float64).reshape(h, w, 3)
    want, _ = P.Scene(d.ptr, 0, keepalive=d).render(w, h, spp, depth)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    ref = oracle.Scene(d.ptr, d).render(w, h, spp, depth, threads=8)["rgb"]
    assert float((np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)).max()) <= 1e-5


def test_stubs_typecheck_against_the_runtime_api():
    """ptx_stubs.c cannot be built without OCaml, but it can be TYPE-CHECKED: gcc -fsyntax-only against minimal declarations
    of the runtime's documented C interface (tests/c/mock_caml) catches what a regex cannot -- name clashes between stubs
    and the marshalling header, wrong arities, undeclared fields of ptx_ml_flat."""
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "tests", "c", "mock_caml"), "-I", os.path.join(ROOT, "include"), "-I", B,
                           os.path.join(B, "ptx_stubs.c")])


def _tree_of(driver, flat):
    out = subprocess.check_output([driver, flat, "tree"], text=True)
    return dict(zip(out.split()[0::2], map(int, out.split()[1::2])))


@pytest.mark.parametrize("scene", ["cornell", "ganesha"])
def test_marshalling_builds_the_triangle_scenes(oracle, driver, tmp_path, scene):
    """The mixed sphere + triangle leaf of cornell-box (main.ml:93-168) and the mesh + pre-tested floor of ganesha
    (main.ml:88-119,205-260) through ptx_ml_scene_create: same tree as the oracle's Shape_tree.create on the same scene."""
    d = oracle.desc_cornell(64, 64, 0.0) if scene == "cornell" else oracle.desc_ganesha_like(64, 36, 3000, 7)
    arr = d.arrays()
    assert len(arr["tri_material"]) > 0 and (scene == "cornell" or len(arr["floor_material"]) == 2)
    flat = str(tmp_path / "flat.bin")
    _flat_file(flat, arr, leaf_kind=1, length_cutoff=int(arr["build"][1]))
    got = _tree_of(driver, flat)
    info = oracle.Scene(d.ptr, d).info()
    assert (got["nodes"], got["depth"], got["leaves"], got["slots"]) == (info["nodes"], info["depth"], info["leaves"], info["slots"])


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cornell", "ganesha"])
def test_marshalling_renders_the_triangle_scenes(oracle, driver, tmp_path, scene):
    """Framebuffer through the OCaml marshalling == ptx_render on the same description, bit for bit (emitter, floor,
    per-face materials and texture coordinates all have to arrive)."""
    import path_tracer_ocaml_amd as P
    w, h, spp, depth = (96, 96, 8, 12) if scene == "cornell" else (128, 72, 4, 8)
    d = oracle.desc_cornell(w, h, 12.0) if scene == "cornell" else oracle.desc_ganesha_like(w, h, 5000, 7)
    arr = d.arrays()
    flat, outp = str(tmp_path / "flat.bin"), str(tmp_path / "out.bin")
    _flat_file(flat, arr, leaf_kind=1, length_cutoff=int(arr["build"][1]))
    out = subprocess.check_output([driver, flat, "render", str(w), str(h), str(spp), str(depth), "1", outp], text=True)
    assert f"progress_pixels {w * h}" in out
    got = np.fromfile(outp, dtype=np.float64).reshape(h, w, 3)
    want, _ = P.Scene(d.ptr, 0, keepalive=d).render(w, h, spp, depth)
    assert want.max() > 0.05
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cornell", "ganesha"])
def test_marshalling_photon_maps_what_ptx_ppm_render_does(oracle, driver, tmp_path, scene):
    """Ptx.ppm_render's marshalling (params, point / spot lights, iteration callback) against ptx_ppm_render: img_sum bit-equal."""
    import path_tracer_ocaml_amd as P
    from path_tracer_ocaml_amd import abi
    if scene == "cornell":
        w = h = 80
        d = oracle.desc_cornell(w, h, 0.0)
        lights = oracle.lights_cornell(w, h)
    else:
        w, h = 96, 54
        d = oracle.desc_ganesha_like(w, h, 5000, 7)
        d.d.background.kind = abi.PTX_BG_BLACK
        lights = oracle.Scene(d.ptr, d).lights_ganesha()
    params = abi.ppm_params(w, h, iterations=2, photon_count=10000)
    arr = d.arrays()
    flat, ppm, outp = str(tmp_path / "flat.bin"), str(tmp_path / "ppm.bin"), str(tmp_path / "out.bin")
    _flat_file(flat, arr, leaf_kind=1, length_cutoff=int(arr["build"][1]))
    _ppm_file(ppm, params, lights)
    out = subprocess.check_output([driver, flat, "ppm", ppm, outp], text=True)
    assert "iterations 2" in out
    got = np.fromfile(outp, dtype=np.float64).reshape(h, w, 3)
    want, st = P.Scene(d.ptr, 0, keepalive=d).ppm_render(params, lights)
    assert st["photons_stored"] > 1000 and want.max() > 0
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
