"""ctypes mirror of include/ptx.h (type declarations only -- no behaviour).

Every structure here must match the C header field for field; tests/test_abi.py
checks the sizes against the compiled library.
"""
import ctypes as C

PTX_ABI_VERSION = 4

PTX_MAT_LAMBERTIAN, PTX_MAT_METAL, PTX_MAT_DIELECTRIC = 0, 1, 2
PTX_TEX_SOLID, PTX_TEX_CHECKER = 0, 1
PTX_BG_BLACK, PTX_BG_SKY = 0, 1
PTX_LEAF_SIMD, PTX_LEAF_ARRAY = 0, 1
PTX_KERNEL_NAMES = ("generate", "trace", "shade", "accum", "film", "bounce")
PTX_N_KERNELS = 6
PTX_RENDER_ASYNC = 1

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("index", C.c_double), ("emit", C.c_double * 3)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("reserved", C.c_int32),
                ("even", C.c_double * 3), ("odd", C.c_double * 3)]


class Camera(C.Structure):
    _fields_ = [("lower_left_x", C.c_double), ("lower_left_y", C.c_double), ("view_x", C.c_double), ("view_y", C.c_double)]


class Background(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("horizon", C.c_double * 3), ("zenith", C.c_double * 3)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_spheres", C.c_int32), ("sphere_x", c_double_p), ("sphere_y", c_double_p), ("sphere_z", c_double_p),
        ("sphere_r", c_double_p), ("sphere_material", c_int32_p),
        ("n_vertices", C.c_int32), ("vertex_x", c_double_p), ("vertex_y", c_double_p), ("vertex_z", c_double_p),
        ("n_triangles", C.c_int32), ("tri_indices", c_int32_p), ("tri_uv", c_double_p), ("tri_material", c_int32_p),
        ("n_floor_triangles", C.c_int32), ("floor_vertices", c_double_p), ("floor_uv", c_double_p),
        ("floor_material", c_int32_p),
        ("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
        ("camera", Camera), ("background", Background),
        ("leaf_kind", C.c_int32), ("length_cutoff", C.c_int32), ("num_bins", C.c_int32), ("reserved", C.c_int32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32), ("max_bounces", C.c_int32),
        ("band_rows", C.c_int32), ("band_first", C.c_int32), ("band_step", C.c_int32),
        ("count_work", C.c_int32), ("time_kernels", C.c_int32), ("passes_per_batch", C.c_int32),
        ("n_gpus", C.c_int32), ("flags", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_int64), ("segments", C.c_int64), ("nodes_tested", C.c_int64), ("prims_tested", C.c_int64),
        ("floor_tested", C.c_int64), ("render_ms", C.c_double),
        ("kernel_ms", C.c_double * PTX_N_KERNELS), ("kernel_launches", C.c_int64 * PTX_N_KERNELS),
        ("tree_nodes", C.c_int32), ("tree_depth", C.c_int32), ("tree_leaves", C.c_int32), ("leaf_slots", C.c_int32),
        ("build_ms", C.c_double), ("traversal_in_lds", C.c_int32), ("bvh_built_on_gpu", C.c_int32),
        ("filter_undecided", C.c_int64), ("filter_fallback_steps", C.c_int64),
        ("peer_copies", C.c_int32), ("staged_copies", C.c_int32),
        ("solo_launches", C.c_int32), ("reserved_stats", C.c_int32),
    ]


PTX_LIGHT_POINT, PTX_LIGHT_SPOT = 0, 1


class Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("position", C.c_double * 3), ("direction", C.c_double * 3),
                ("color", C.c_double * 3), ("power", C.c_double)]


class PpmParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("iterations", C.c_int32), ("max_bounces", C.c_int32),
                ("photon_count", C.c_int32), ("reserved", C.c_int32), ("alpha", C.c_double)]


class PpmStats(C.Structure):
    _fields_ = [("photons_stored", C.c_int64), ("photon_rays", C.c_int64), ("eye_rays", C.c_int64), ("neighbors", C.c_int64),
                ("photon_ms", C.c_double), ("build_ms", C.c_double), ("gather_ms", C.c_double), ("total_ms", C.c_double),
                ("last_radius", C.c_double)]


def ppm_params(width=600, height=None, iterations=10, max_bounces=4, photon_count=75000, alpha=2.0 / 3.0):
    """Defaults of Progressive_photon_map.Args.parse (progressive_photon_map.ml:17-54)."""
    p = PpmParams()
    p.width, p.height = width, width if height is None else height
    p.iterations, p.max_bounces, p.photon_count, p.alpha = iterations, max_bounces, photon_count, alpha
    return p
