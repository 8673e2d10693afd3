"""ctypes binding of libpt_host.so -- the host-side mirror of the reference's scene executables
(shirley_spheres / cornell-box / ganesha main.ml) and of Bimage_unix.Stb.write."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libpt_host.so")
_LIB = None

EXPORTS = ("pth_scene_desc", "pth_scene_free", "pth_camera_create", "pth_scene_shirley", "pth_scene_cornell",
           "pth_scene_ganesha_like", "pth_write_png", "pth_last_error", "pth_ply_load", "pth_ply_free", "pth_ply_count",
           "pth_ply_floats", "pth_ply_ints", "pth_ply_rows", "pth_scene_ganesha_ply", "pth_write_ganesha_like_ply", "pth_lights_cornell", "pth_lights_ganesha", "pth_ppm_gamma")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: make -C {os.path.join(_HERE, 'host')}")
        L = C.CDLL(HOST_LIB_PATH)
        L.pth_scene_desc.restype = C.POINTER(abi.SceneDesc)
        L.pth_scene_desc.argtypes = [C.c_void_p]
        L.pth_scene_free.argtypes = [C.c_void_p]
        L.pth_camera_create.argtypes = [abi.c_double_p, abi.c_double_p, abi.c_double_p, C.c_double, C.c_double,
                                        C.POINTER(abi.Camera), abi.c_double_p]
        L.pth_scene_shirley.restype = C.c_void_p
        L.pth_scene_shirley.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64]
        L.pth_scene_cornell.restype = C.c_void_p
        L.pth_scene_cornell.argtypes = [C.c_int32, C.c_int32, C.c_double]
        L.pth_scene_ganesha_like.restype = C.c_void_p
        L.pth_scene_ganesha_like.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint64]
        L.pth_write_png.argtypes = [C.c_char_p, C.c_int32, C.c_int32, abi.c_double_p]
        L.pth_last_error.restype = C.c_char_p
        L.pth_ply_load.restype = C.c_void_p
        L.pth_ply_load.argtypes = [C.c_char_p]
        L.pth_ply_free.argtypes = [C.c_void_p]
        L.pth_ply_count.restype = C.c_int64
        L.pth_ply_count.argtypes = [C.c_void_p, C.c_char_p]
        L.pth_ply_floats.restype = abi.c_double_p
        L.pth_ply_floats.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.pth_ply_ints.restype = C.POINTER(C.c_int64)
        L.pth_ply_ints.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.pth_ply_rows.restype = C.POINTER(C.c_int64)
        L.pth_ply_rows.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(abi.c_int32_p)]
        L.pth_scene_ganesha_ply.restype = C.c_void_p
        L.pth_scene_ganesha_ply.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
        L.pth_write_ganesha_like_ply.argtypes = [C.c_char_p, C.c_int32, C.c_uint64]
        L.pth_lights_cornell.argtypes = [C.c_int32, C.c_int32, C.POINTER(abi.Light)]
        L.pth_lights_ganesha.argtypes = [C.c_void_p, C.POINTER(abi.Light)]
        L.pth_ppm_gamma.argtypes = [abi.c_double_p, C.c_int64, C.c_int32, abi.c_double_p]
        _LIB = L
    return _LIB


class HostScene:
    """Owns a ptx_scene_desc built by one of the scene mirrors."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("scene construction failed")
        self._h = handle
        self.ptr = lib().pth_scene_desc(handle)

    @property
    def d(self):
        return self.ptr.contents

    def arrays(self):
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
            lib().pth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shirley_spheres(width, height, no_simd=False, seed=42):
    """shirley_spheres/bin/main.ml: Random.init 42; Shirley_spheres.spheres (); camera (width // height)."""
    return HostScene(lib().pth_scene_shirley(width, height, int(no_simd), seed))


def cornell_box(width, height, ceiling_emit=12.0):
    """cornell-box/bin/main.ml geometry + the documented ceiling emitter (path-integrator lighting)."""
    return HostScene(lib().pth_scene_cornell(width, height, ceiling_emit))


def ganesha_like(width, height, n_target=150000, seed=7):
    """ganesha/bin/main.ml camera / floor / material over the synthetic mesh."""
    return HostScene(lib().pth_scene_ganesha_like(width, height, n_target, seed))


def write_png(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    rc = lib().pth_write_png(path.encode(), w, h, rgb.ctypes.data_as(abi.c_double_p))
    if rc != 0:
        raise IOError(f"pth_write_png({path}) failed: {rc}")


class PlyError(RuntimeError):
    pass


class Ply:
    """Ply.of_bigstring (ply_format/src/ply.ml:340-352) through libpt_host.so."""

    def __init__(self, path):
        self._h = lib().pth_ply_load(path.encode())
        if not self._h:
            raise PlyError(lib().pth_last_error().decode())

    def count(self, key):
        return int(lib().pth_ply_count(self._h, key.encode()))

    def floats(self, element, prop):
        p = lib().pth_ply_floats(self._h, element.encode(), prop.encode())
        return None if not p else np.ctypeslib.as_array(p, shape=(self.count(element),)).copy()

    def ints(self, element, prop):
        p = lib().pth_ply_ints(self._h, element.encode(), prop.encode())
        return None if not p else np.ctypeslib.as_array(p, shape=(self.count(element),)).copy()

    def rows(self, list_property):
        lengths = abi.c_int32_p()
        p = lib().pth_ply_rows(self._h, list_property.encode(), C.byref(lengths))
        if not p:
            return None
        n = self.count(list_property)
        ln = np.ctypeslib.as_array(lengths, shape=(n,)).copy() if n else np.zeros(0, dtype=np.int32)
        flat = np.ctypeslib.as_array(p, shape=(int(ln.sum()),)).copy() if ln.sum() else np.zeros(0, dtype=np.int32)
        out, k = [], 0
        for m in ln:
            out.append(flat[k:k + m])
            k += m
        return out

    def close(self):
        if self._h:
            lib().pth_ply_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ganesha_ply(path, width, height):
    """ganesha/bin/main.ml with -ganesha-ply PATH."""
    h = lib().pth_scene_ganesha_ply(path.encode(), width, height)
    if not h:
        raise PlyError(lib().pth_last_error().decode())
    return HostScene(h)


def write_ganesha_like_ply(path, n_target=150000, seed=7):
    if lib().pth_write_ganesha_like_ply(path.encode(), n_target, seed) != 0:
        raise IOError(path)


def lights_cornell(width, height):
    """cornell-box/bin/main.ml:225-228."""
    out = (abi.Light * 1)()
    n = lib().pth_lights_cornell(width, height, out)
    return [out[i] for i in range(n)]


def lights_ganesha(scene):
    """ganesha/bin/main.ml:267-282, from the mesh's camera-space bounding box."""
    out = (abi.Light * 2)()
    n = lib().pth_lights_ganesha(scene._h, out)
    return [out[i] for i in range(n)]


def ppm_gamma(img_sum, n):
    """save_image (progressive_photon_map.ml:398-410): (sum / n) ** (1 / 2.2)."""
    a = np.ascontiguousarray(img_sum, dtype=np.float64)
    out = np.zeros_like(a)
    lib().pth_ppm_gamma(a.ctypes.data_as(abi.c_double_p), a.size, n, out.ctypes.data_as(abi.c_double_p))
    return out
