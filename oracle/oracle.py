"""ctypes loader for the CPU ORACLE (oracle/libpt_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (path_tracer_ocaml_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from path_tracer_ocaml_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

dp = abi.c_double_p
ip = abi.c_int32_p


def build(force=False):
    so = os.path.join(_HERE, "libpt_oracle.so")
    srcs = [os.path.join(_HERE, "pt_oracle.c"), os.path.join(_HERE, "..", "include", "ptx.h"),
            os.path.join(_HERE, "..", "path_tracer_ocaml_amd", "csrc", "pt_math.h")]
    stale = force or not os.path.exists(so) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale and os.path.exists(srcs[0]):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


_SO_OVERRIDE = None


def use_native_build():
    """bench.py's cpu_baseline leg: (re)build the oracle ON THIS MACHINE with the flags BASELINE.md section 4 states
    (-O3 -march=native -ffp-contract=off) into oracle/_native/ and make lib() load that copy.  The shipped
    libpt_oracle.so is -O2 -march=x86-64-v3 because it is built in one container and run in another.  Same C source,
    contraction off in both: the results are bit-identical, only the speed differs.  Returns the flags actually used."""
    global _SO_OVERRIDE, _LIB
    if _LIB is not None:
        return "-O2 -march=x86-64-v3 (already loaded)"
    out_dir = os.path.join(_HERE, "_native")
    so = os.path.join(out_dir, "libpt_oracle_native.so")
    flags = ["-O3", "-march=native", "-ffp-contract=off", "-fno-math-errno", "-fno-fast-math", "-fPIC", "-std=gnu11", "-fvisibility=hidden"]
    try:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", *flags, "-shared", "-o", so, os.path.join(_HERE, "pt_oracle.c"), "-lm", "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        _SO_OVERRIDE = so
        return "-O3 -march=native"
    except Exception:
        return "-O2 -march=x86-64-v3 (native rebuild failed)"


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = _SO_OVERRIDE or build()
    L = C.CDLL(so)
    L.orc_math.restype = C.c_double
    L.orc_math.argtypes = [C.c_int, C.c_double, C.c_double]
    L.orc_math_vec.argtypes = [C.c_int, C.c_int64, dp, dp, dp]
    L.orc_lds_alpha.argtypes = [C.c_int, dp]
    L.orc_lds_phi.restype = C.c_double
    L.orc_lds_phi.argtypes = [C.c_int]
    L.orc_lds_get.restype = C.c_double
    L.orc_lds_get.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_filter_binomial.argtypes = [C.c_int, C.c_int, dp, dp]
    L.orc_tile_split.argtypes = [C.c_int, C.c_int, C.c_int, ip, C.c_int]
    L.orc_film_tile_kat.argtypes = [C.c_int] * 7 + [dp, ip]
    L.orc_bbox_is_hit.argtypes = [dp, dp, dp, C.c_double, C.c_double]
    L.orc_bbox_mem.argtypes = [dp, dp]
    L.orc_sphere_intersect.argtypes = [dp, C.c_double, dp, dp, C.c_double, C.c_double, dp]
    L.orc_spheres_intersect_packet.argtypes = [dp, dp, dp, dp, C.c_int, dp, dp, C.c_double, C.c_double, dp]
    L.orc_triangle_intersect.argtypes = [dp, dp, dp, C.c_double, C.c_double, dp]
    L.orc_unit_square_to_hemisphere.argtypes = [C.c_double, C.c_double, dp]
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_scene_info.argtypes = [C.c_void_p, ip, dp]
    L.orc_scene_tree.argtypes = [C.c_void_p, dp, ip, ip]
    L.orc_camera_create.argtypes = [dp, dp, dp, C.c_double, C.c_double, dp, dp]
    L.orc_camera_ray.argtypes = [dp, C.c_double, C.c_double, dp]
    L.orc_md5.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    L.orc_random_floats.argtypes = [C.c_int64, C.c_int, dp]
    L.orc_desc_destroy.argtypes = [C.c_void_p]
    L.orc_desc_get.restype = C.POINTER(abi.SceneDesc)
    L.orc_desc_get.argtypes = [C.c_void_p]
    L.orc_desc_shirley.restype = C.c_void_p
    L.orc_desc_shirley.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64]
    L.orc_desc_cornell.restype = C.c_void_p
    L.orc_desc_cornell.argtypes = [C.c_int, C.c_int, C.c_double]
    L.orc_desc_ganesha_like.restype = C.c_void_p
    L.orc_desc_ganesha_like.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64]
    L.orc_trace_samples.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, ip, ip, ip, dp,
                                    C.POINTER(C.c_int64)]
    L.orc_intersect_rays.argtypes = [C.c_void_p, C.c_int64, dp, dp, dp, ip, C.POINTER(C.c_int64)]
    L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_int64), dp]
    L.orc_set_math.argtypes = [C.c_int]
    L.orc_ppm_render.argtypes = [C.c_void_p, C.POINTER(abi.PpmParams), C.POINTER(abi.Light), C.c_int, dp, C.POINTER(C.c_int64), dp]
    L.orc_lights_cornell.argtypes = [C.c_int, C.c_int, C.POINTER(abi.Light)]
    L.orc_lights_ganesha.argtypes = [C.c_void_p, C.POINTER(abi.Light)]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(dp)


def _ip(a):
    return a.ctypes.data_as(ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


COUNTER_NAMES = ("samples", "segments", "nodes_tested", "prims_tested", "floor_tested")


class Desc:
    """An oracle-owned scene description (ptx_scene_desc) built by one of the scene builders."""

    def __init__(self, handle):
        self._h = handle
        self.ptr = lib().orc_desc_get(handle)

    @property
    def d(self):
        return self.ptr.contents

    def arrays(self):
        """numpy copies of every array in the description (for comparing against the product's builders)."""
        d = self.d

        def arr(p, n, dt):
            if n == 0 or not p:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(p, shape=(n,)).copy()

        out = {
            "sphere_x": arr(d.sphere_x, d.n_spheres, np.float64), "sphere_y": arr(d.sphere_y, d.n_spheres, np.float64),
            "sphere_z": arr(d.sphere_z, d.n_spheres, np.float64), "sphere_r": arr(d.sphere_r, d.n_spheres, np.float64),
            "sphere_material": arr(d.sphere_material, d.n_spheres, np.int32),
            "vertex_x": arr(d.vertex_x, d.n_vertices, np.float64), "vertex_y": arr(d.vertex_y, d.n_vertices, np.float64),
            "vertex_z": arr(d.vertex_z, d.n_vertices, np.float64),
            "tri_indices": arr(d.tri_indices, 3 * d.n_triangles, np.int32),
            "tri_uv": arr(d.tri_uv, 6 * d.n_triangles, np.float64),
            "tri_material": arr(d.tri_material, d.n_triangles, np.int32),
            "floor_vertices": arr(d.floor_vertices, 9 * d.n_floor_triangles, np.float64),
            "floor_uv": arr(d.floor_uv, 6 * d.n_floor_triangles, np.float64),
            "floor_material": arr(d.floor_material, d.n_floor_triangles, np.int32),
        }
        mats = np.zeros((d.n_materials, 6))
        for i in range(d.n_materials):
            m = d.materials[i]
            mats[i] = [m.kind, m.texture, m.index, m.emit[0], m.emit[1], m.emit[2]]
        texs = np.zeros((d.n_textures, 9))
        for i in range(d.n_textures):
            t = d.textures[i]
            texs[i] = [t.kind, t.width, t.height, *t.even, *t.odd]
        out["materials"] = mats
        out["textures"] = texs
        out["camera"] = np.array([d.camera.lower_left_x, d.camera.lower_left_y, d.camera.view_x, d.camera.view_y])
        out["background"] = np.array([d.background.kind, *d.background.horizon, *d.background.zenith])
        out["build"] = np.array([d.leaf_kind, d.length_cutoff, d.num_bins])
        return out

    def close(self):
        if self._h:
            lib().orc_desc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def desc_shirley(width, height, no_simd=False, seed=42):
    return Desc(lib().orc_desc_shirley(width, height, int(no_simd), seed))


def desc_cornell(width, height, ceiling_emit=12.0):
    return Desc(lib().orc_desc_cornell(width, height, ceiling_emit))


def desc_ganesha_like(width, height, n_target=150000, seed=7):
    return Desc(lib().orc_desc_ganesha_like(width, height, n_target, seed))


class Scene:
    """Oracle scene (BVH built exactly as Shape_tree.create does)."""

    def __init__(self, desc_ptr, keepalive=None):
        self._keep = keepalive
        self._h = lib().orc_scene_create(desc_ptr)
        if not self._h:
            raise RuntimeError("orc_scene_create failed")

    def info(self):
        out = np.zeros(5, dtype=np.int32)
        ms = C.c_double()
        lib().orc_scene_info(self._h, _ip(out), C.byref(ms))
        return {"nodes": int(out[0]), "leaves": int(out[1]), "depth": int(out[2]), "slots": int(out[3]),
                "n_prims": int(out[4]), "build_ms": ms.value}

    def tree(self):
        inf = self.info()
        bbox = np.zeros((inf["nodes"], 6))
        info = np.zeros((inf["nodes"], 4), dtype=np.int32)
        order = np.zeros(max(inf["slots"], 1), dtype=np.int32)
        n = lib().orc_scene_tree(self._h, _dp(bbox), _ip(info), _ip(order))
        assert n == inf["nodes"]
        return bbox, info, order[: inf["slots"]]

    def trace_samples(self, width, height, spp, max_bounces, xs, ys, passes):
        xs, ys, passes = i32(xs), i32(ys), i32(passes)
        n = len(xs)
        rgb = np.zeros((n, 3))
        ct = (C.c_int64 * 5)()
        lib().orc_trace_samples(self._h, width, height, spp, max_bounces, n, _ip(xs), _ip(ys), _ip(passes), _dp(rgb), ct)
        return rgb, dict(zip(COUNTER_NAMES, [int(c) for c in ct]))

    def intersect_rays(self, origins, directions):
        o, d = f64(origins), f64(directions)
        n = o.shape[0]
        t = np.zeros(n)
        prim = np.zeros(n, dtype=np.int32)
        ct = (C.c_int64 * 5)()
        lib().orc_intersect_rays(self._h, n, _dp(o), _dp(d), _dp(t), _ip(prim), ct)
        return t, prim, dict(zip(COUNTER_NAMES, [int(c) for c in ct]))

    def render(self, width, height, spp, max_bounces, threads=1, want_raw=False, count=False):
        rgb = np.zeros((height, width, 3))
        raw = np.zeros((height, width, 3)) if want_raw else None
        ct = (C.c_int64 * 5)() if count else None
        ms = C.c_double()
        lib().orc_render(self._h, width, height, spp, max_bounces, threads, _dp(rgb), _dp(raw) if want_raw else None,
                         ct, C.byref(ms))
        out = {"rgb": rgb, "ms": ms.value}
        if want_raw:
            out["raw"] = raw
        if count:
            out["counters"] = dict(zip(COUNTER_NAMES, [int(c) for c in ct]))
        return out

    def ppm_render(self, params, lights):
        """Progressive_photon_map.Make(Scene).go without the gamma / PNG step: img_sum (H, W, 3), stats."""
        arr = (abi.Light * len(lights))(*lights)
        img = np.zeros((params.height, params.width, 3))
        st = (C.c_int64 * 4)()
        radius = C.c_double()
        rc = lib().orc_ppm_render(self._h, C.byref(params), arr, len(lights), _dp(img), st, C.byref(radius))
        if rc != 0:
            raise RuntimeError(f"orc_ppm_render failed: {rc}")
        return img, {"photons_stored": int(st[0]), "photon_rays": int(st[1]), "eye_rays": int(st[2]), "neighbors": int(st[3]),
                     "last_radius": radius.value}

    def lights_ganesha(self):
        out = (abi.Light * 2)()
        n = lib().orc_lights_ganesha(self._h, out)
        return [out[i] for i in range(n)]

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def math_vec(fn, a, b=None):
    """pt_math.h (mode 0) or libm (mode 1) on the host, elementwise; fn numbering as ptx_math_eval."""
    a = f64(a)
    bb = f64(b) if b is not None else None
    out = np.zeros_like(a)
    lib().orc_math_vec(fn, a.size, _dp(a), _dp(bb) if bb is not None else None, _dp(out))
    return out


def lds_get_vec(n_dim, offsets, dims):
    alpha = np.zeros(n_dim)
    lib().orc_lds_alpha(n_dim, _dp(alpha))
    x = 0.5 + alpha[np.asarray(dims)] * (1 + np.asarray(offsets)).astype(np.float64)
    return x - np.trunc(x)


def lights_cornell(width, height):
    out = (abi.Light * 1)()
    n = lib().orc_lights_cornell(width, height, out)
    return [out[i] for i in range(n)]


def set_math(mode):
    """0 = pt_math.h shared host/device functions (default); 1 = platform libm."""
    lib().orc_set_math(mode)
