"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU builds only; SURVEY section 5 "race detection /
sanitizers").  `make asan` (path_tracer_ocaml_amd/host, oracle) builds three instrumented executables:

* host_asan_driver           the PLY reader (a parser of untrusted binary input: ply_format/src/ply.ml:208-235,288-352),
                             the scene builders, the PNG writer and the host BVH builder (shape_tree.ml:72-196);
* ocaml_binding_asan_driver  the OCaml stub's marshalling header with a host-only scene (device -1: no HIP call);
* oracle_asan_driver         the checker itself (oracle/pt_oracle.c) over the three stock scenes and the photon mapper.

A sanitizer report aborts the process: every run below must exit 0 with nothing from the sanitizers on stderr."""
import os
import struct
import subprocess

import numpy as np
import pytest

from test_ply import mesh_payload, write_ply

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = os.path.join(ROOT, "build", "asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "path_tracer_ocaml_amd", "host"), "asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    return ASAN


def run_clean(cmd, env=ENV, timeout=300):
    r = subprocess.run(cmd, capture_output=True, env=env, timeout=timeout)
    out, err = r.stdout.decode("utf-8", "replace"), r.stderr.decode("utf-8", "replace")  # error messages quote the file's bytes
    assert r.returncode == 0, (cmd, r.returncode, err[-3000:])
    assert "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    return out


HEADER = ["format binary_little_endian 1.0", "element vertex 50", "property float x", "property float y", "property float z",
          "element face 80", "property list uint8 int vertex_indices", "end_header"]


def valid_ply(path, seed=0):
    rng = np.random.default_rng(seed)
    verts = (rng.normal(size=(50, 3)) * 30).astype(np.float32)
    faces = [tuple(int(v) for v in rng.choice(50, 3, replace=False)) for _ in range(80)]
    write_ply(path, HEADER, mesh_payload(verts, faces))


def test_ply_reader_on_hostile_files(built, tmp_path):
    exe = os.path.join(built, "host_asan_driver")
    good = str(tmp_path / "good.ply")
    valid_ply(good)
    out = run_clean([exe, "ply", good])
    assert "loaded: vertex 50 rows 80" in out and "prims 80" in out
    cases = [
        (["format ascii 1.0", "element vertex 0", "property float x", "end_header"], b"", b"ply\n"),
        (["format binary_big_endian 1.0", "end_header"], b"", b"ply\n"),
        (["format binary_little_endian 1.0", "end_header"], b"", b"plx\n"),
        (["format binary_little_endian 1.0", "element vertex 1", "property float x"], b"", b"ply\n"),          # no end_header
        (["element vertex 1", "property float x", "end_header"], b"\0\0\0\0", b"ply\n"),
        (["format binary_little_endian 1.0", "element vertex 1", "property quux x", "end_header"], b"", b"ply\n"),
        (["format binary_little_endian 1.0", "element vertex 4", "property float x", "end_header"], b"\0" * 8, b"ply\n"),  # truncated
        (["format binary_little_endian 1.0", "element vertex -1", "property float x", "end_header"], b"", b"ply\n"),
        (["format binary_little_endian 1.0", "element vertex 99999999999999999999", "property float x", "end_header"], b"", b"ply\n"),
        (["format binary_little_endian 1.0", "element vertex 2000000000", "property double x", "end_header"], b"\0" * 64, b"ply\n"),
        (["format binary_little_endian 1.0", "element face 1", "property list uchar int vertex_indices", "end_header"], b"\xff" + b"\0" * 7, b"ply\n"),  # list longer than the file
        (["format binary_little_endian 1.0", "element face 3", "property list int int vertex_indices", "end_header"],
         struct.pack("<i", -5) + b"\0" * 32, b"ply\n"),                                                       # negative list length
        (["format binary_little_endian 1.0", "element face 1", "property list uint int vertex_indices", "end_header"],
         struct.pack("<I", 0xFFFFFFFF) + b"\0" * 32, b"ply\n"),                                               # 4 G entries claimed
        (HEADER, mesh_payload(np.zeros((50, 3), dtype=np.float32), [(0, 1, 2)] * 79 + [(0, 1, 77)]), b"ply\n"),  # index out of bounds
        (HEADER, mesh_payload(np.zeros((50, 3), dtype=np.float32), [(0, 1, 2)] * 79 + [(0, 1, -3)]), b"ply\n"),  # negative index
        (HEADER, mesh_payload(np.zeros((50, 3), dtype=np.float32), [(0, 1, 2, 3)] * 80), b"ply\n"),              # quads
        (["format binary_little_endian 1.0"] + ["element e%d 1" % k for k in range(2000)] + ["end_header"], b"", b"ply\n"),
        (["format binary_little_endian 1.0", "element vertex 3"] + ["property float p%d" % k for k in range(3000)] + ["end_header"], b"\0" * 100, b"ply\n"),
        (["format binary_little_endian 1.0", "element vertex 1", "property", "end_header"], b"", b"ply\n"),
        (["format binary_little_endian 1.0", "element", "end_header"], b"", b"ply\n"),
        (["format binary_little_endian 1.0", "element vertex 1", "property list", "end_header"], b"", b"ply\n"),
        ([], b"", b""),                                                                                          # empty file
        ([], b"", b"ply"),
    ]
    for k, (header, payload, magic) in enumerate(cases):
        p = str(tmp_path / f"bad{k}.ply")
        write_ply(p, header, payload, magic)
        out = run_clean([exe, "ply", p])
        assert "rejected:" in out or "no scene:" in out or "loaded:" in out, (k, out)
    run_clean([exe, "ply", str(tmp_path / "does_not_exist.ply")])


def test_ply_reader_on_mutated_files(built, tmp_path):
    """Byte-level mutations of a valid file (header and payload), truncations and garbage tails: whatever the reader
    decides, it decides it without reading or writing out of bounds."""
    exe = os.path.join(built, "host_asan_driver")
    good = str(tmp_path / "good.ply")
    valid_ply(good, seed=1)
    data = open(good, "rb").read()
    hdr_end = data.index(b"end_header\n") + len(b"end_header\n")
    rng = np.random.default_rng(9)
    outcomes = set()
    for k in range(120):
        b = bytearray(data)
        kind = k % 4
        if kind == 0:    # flip bytes in the header
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, hdr_end))] = int(rng.integers(0, 256))
        elif kind == 1:  # flip bytes in the payload (list lengths, indices, floats)
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(hdr_end, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2:  # truncate anywhere
            b = b[: int(rng.integers(0, len(b)))]
        else:            # replace a header number by a large / negative / non-numeric token
            tok = [b"50", b"80"][int(rng.integers(0, 2))]
            b = b.replace(tok, [b"-7", b"4294967296", b"1e9", b"0x10", b"", b"50 50"][int(rng.integers(0, 6))], 1)
        p = str(tmp_path / f"mut{k}.ply")
        open(p, "wb").write(bytes(b))
        out = run_clean([exe, "ply", p])
        outcomes.add("loaded" if "loaded:" in out else "rejected")
    assert outcomes == {"loaded", "rejected"}  # the mutations reached both sides of the parser


@pytest.mark.parametrize("scene", ["shirley", "shirley_array", "cornell", "ganesha"])
def test_scene_builders_and_host_bvh(built, scene):
    out = run_clean([os.path.join(built, "host_asan_driver"), "scene", scene])
    assert "nodes" in out and "depth" in out


def test_png_writer(built, tmp_path):
    p = str(tmp_path / "a.png")
    run_clean([os.path.join(built, "host_asan_driver"), "png", p])
    assert open(p, "rb").read(8) == b"\x89PNG\r\n\x1a\n"


@pytest.mark.parametrize("scene", ["shirley", "cornell", "ganesha"])
def test_ocaml_marshalling_with_a_host_only_scene(built, oracle, tmp_path, scene):
    """ptx_ml_scene_create (bindings/ocaml/ptx_ml_marshal.h) with device -1 under the sanitizers.  The library it calls into
    is not instrumented, the marshalling is; LeakSanitizer stays off because the HIP runtime the library links keeps
    process-lifetime allocations."""
    from test_ocaml_binding import _flat_file
    d = {"shirley": lambda: oracle.desc_shirley(64, 32), "cornell": lambda: oracle.desc_cornell(32, 32, 12.0),
         "ganesha": lambda: oracle.desc_ganesha_like(64, 36, 2000, 7)}[scene]()
    arr = d.arrays()
    flat = str(tmp_path / "flat.bin")
    _flat_file(flat, arr, leaf_kind=int(arr["build"][0]), length_cutoff=int(arr["build"][1]))
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1")
    out = run_clean([os.path.join(built, "ocaml_binding_asan_driver"), flat, "tree"], env=env)
    info = oracle.Scene(d.ptr, d).info()
    assert f"nodes {info['nodes']} depth {info['depth']}" in out


def test_the_oracle_itself(built):
    out = run_clean([os.path.join(built, "oracle_asan_driver")])
    assert out.count("nodes") == 6
