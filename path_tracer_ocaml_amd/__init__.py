"""path_tracer_ocaml_amd -- MI355X-native drop-in for the sampling hot path of dalev/path-tracer-ocaml.

The product is the C-ABI shared library ``libptx_hip.so`` (HIP kernels for gfx950 + host BVH builder,
sources in ``csrc/``, interface in ``include/ptx.h``).  This package is only the thin ctypes binding the
tests and the benchmark use, plus a Python mirror of the reference's operator surface
(``Integrator.create / render``, ``Render_command.Args``) in :mod:`path_tracer_ocaml_amd.integrator`.

There is NO CPU fallback: loading fails loudly if the HIP library is missing, and every compute entry
point returns an error when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTX_LIB overrides the library path (kernel-variant experiments); the default is the in-tree build
LIB_PATH = os.environ.get("PTX_LIB") or os.path.join(_HERE, "libptx_hip.so")
_LIB = None

dp = abi.c_double_p
ip = abi.c_int32_p


class PtxError(RuntimeError):
    pass


EXPORTS = (
    "ptx_version", "ptx_leaf_size", "ptx_last_error", "ptx_device_count", "ptx_scene_create", "ptx_scene_destroy",
    "ptx_scene_stats", "ptx_render", "ptx_local_rows", "ptx_global_row", "ptx_render_raw_device",
    "ptx_film_resolve_device", "ptx_trace_samples", "ptx_intersect_rays", "ptx_scene_tree", "ptx_lds_sample",
    "ptx_math_eval", "ptx_ppm_render", "ptx_debug_first_scatter", "ptx_render_multi", "ptx_scene_replicate",
    "ptx_film_resolve_banded_device", "ptx_film_resolve_banded_queue", "ptx_release_workspaces",
    "ptx_image_pin", "ptx_image_unpin",
)

PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64)


def lib():
    """Loads libptx_hip.so; raises PtxError if it has not been built (run __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PtxError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                       "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # PyTorch bundles its own libamdhip64.so.7.  Two HIP runtimes in one process cannot both own the GPU
    # ("No HIP GPUs are available" from whichever initialises second), so when torch is installed load it
    # FIRST: libptx_hip.so's DT_NEEDED libamdhip64.so.7 then binds to the copy that is already mapped.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional: the C ABI does not need it
        pass
    L = C.CDLL(LIB_PATH)
    L.ptx_version.restype = C.c_int32
    L.ptx_leaf_size.restype = C.c_int32
    L.ptx_last_error.restype = C.c_char_p
    L.ptx_device_count.restype = C.c_int32
    L.ptx_scene_create.restype = C.c_void_p
    L.ptx_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_int32]
    L.ptx_scene_destroy.argtypes = [C.c_void_p]
    L.ptx_scene_stats.argtypes = [C.c_void_p, C.POINTER(abi.Stats)]
    L.ptx_render.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), dp, C.POINTER(abi.Stats), C.c_void_p, C.c_void_p]
    L.ptx_local_rows.argtypes = [C.POINTER(abi.RenderParams)]
    L.ptx_global_row.argtypes = [C.POINTER(abi.RenderParams), C.c_int32]
    L.ptx_render_raw_device.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p,
                                        C.POINTER(abi.Stats)]
    L.ptx_film_resolve_device.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptx_film_resolve_banded_device.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_void_p, C.c_void_p]
    L.ptx_film_resolve_banded_queue.argtypes = L.ptx_film_resolve_banded_device.argtypes
    L.ptx_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(abi.RenderParams), dp, C.POINTER(abi.Stats),
                                   C.c_void_p, C.c_void_p]
    L.ptx_scene_replicate.restype = C.c_void_p
    L.ptx_scene_replicate.argtypes = [C.c_void_p, C.c_int32]
    L.ptx_release_workspaces.restype = None
    L.ptx_image_pin.argtypes = [C.c_void_p, dp, C.c_int64]
    L.ptx_image_unpin.argtypes = [C.c_void_p]
    L.ptx_trace_samples.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_int64, ip, ip, ip, dp,
                                    C.POINTER(abi.Stats)]
    L.ptx_intersect_rays.argtypes = [C.c_void_p, C.c_int64, dp, dp, dp, ip, C.POINTER(abi.Stats)]
    L.ptx_scene_tree.argtypes = [C.c_void_p, dp, ip, C.c_int32, ip, C.c_int32]
    L.ptx_lds_sample.argtypes = [C.c_int32, C.c_int32, C.c_int64, ip, ip, dp]
    L.ptx_math_eval.argtypes = [C.c_int32, C.c_int32, C.c_int64, dp, dp, dp]
    L.ptx_ppm_render.argtypes = [C.c_void_p, C.POINTER(abi.PpmParams), C.POINTER(abi.Light), C.c_int32, dp,
                                 C.POINTER(abi.PpmStats), C.c_void_p, C.c_void_p]
    _LIB = L
    return L


def last_error():
    return (lib().ptx_last_error() or b"").decode()


def _check(rc):
    if rc != 0:
        raise PtxError(f"ptx call failed ({rc}): {last_error()}")


def _dp(a):
    return a.ctypes.data_as(dp)


def _ip(a):
    return a.ctypes.data_as(ip)


def render_params(width, height, samples_per_pixel=1, max_bounces=8, band_rows=32, band_first=0, band_step=0,
                  count_work=False, time_kernels=False, passes_per_batch=0, n_gpus=0, asynchronous=False):
    p = abi.RenderParams()
    p.n_gpus = n_gpus
    p.flags = abi.PTX_RENDER_ASYNC if asynchronous else 0  # ptx_render_raw_device: queue the frame, do not wait for it
    p.width, p.height, p.samples_per_pixel, p.max_bounces = width, height, samples_per_pixel, max_bounces
    p.band_rows, p.band_first, p.band_step = band_rows, band_first, band_step
    p.count_work, p.time_kernels, p.passes_per_batch = int(count_work), int(time_kernels), passes_per_batch
    return p


def stats_dict(st):
    return {
        "samples": st.samples, "segments": st.segments, "nodes_tested": st.nodes_tested,
        "prims_tested": st.prims_tested, "floor_tested": st.floor_tested, "render_ms": st.render_ms,
        "kernel_ms": {n: st.kernel_ms[i] for i, n in enumerate(abi.PTX_KERNEL_NAMES)},
        "kernel_launches": {n: st.kernel_launches[i] for i, n in enumerate(abi.PTX_KERNEL_NAMES)},
        "tree_nodes": st.tree_nodes, "tree_depth": st.tree_depth, "tree_leaves": st.tree_leaves,
        "leaf_slots": st.leaf_slots, "build_ms": st.build_ms,
        "traversal_in_lds": bool(st.traversal_in_lds), "bvh_built_on_gpu": bool(st.bvh_built_on_gpu),
        "filter_undecided": st.filter_undecided, "filter_fallback_steps": st.filter_fallback_steps,
        "peer_copies": st.peer_copies, "staged_copies": st.staged_copies, "solo_launches": st.solo_launches,
    }


class Scene:
    """A scene resident in HBM on one GPU (ptx_scene): BVH built on the host like Shape_tree.create."""

    def __init__(self, desc, device=0, keepalive=None):
        """desc: ctypes pointer to (or instance of) abi.SceneDesc."""
        self._keep = keepalive
        self.device = device
        ptr = desc if isinstance(desc, C.POINTER(abi.SceneDesc)) else C.pointer(desc)
        self._h = lib().ptx_scene_create(ptr, device)
        if not self._h:
            raise PtxError(f"ptx_scene_create failed: {last_error()}")

    @classmethod
    def _adopt(cls, handle, device, keepalive=None):
        s = cls.__new__(cls)
        s._keep, s.device, s._h = keepalive, device, handle
        return s

    def replicate(self, device):
        """ptx_scene_replicate: the same flattened scene uploaded to another device (no second BVH build)."""
        h = lib().ptx_scene_replicate(self._h, device)
        if not h:
            raise PtxError(f"ptx_scene_replicate failed: {last_error()}")
        return Scene._adopt(h, device, keepalive=self)

    def stats(self):
        st = abi.Stats()
        _check(lib().ptx_scene_stats(self._h, C.byref(st)))
        return stats_dict(st)

    def tree(self):
        st = self.stats()
        n, slots = st["tree_nodes"], st["leaf_slots"]
        bbox = np.zeros((n, 6))
        info = np.zeros((n, 4), dtype=np.int32)
        order = np.zeros(max(slots, 1), dtype=np.int32)
        got = lib().ptx_scene_tree(self._h, _dp(bbox), _ip(info), n, _ip(order), slots)
        assert got == n
        return bbox, info, order[:slots]

    def render(self, width, height, samples_per_pixel, max_bounces, progress=None, out=None, **kw):
        """ptx_render: post-gamma f64 framebuffer (H, W, 3) on the host + stats.  `out`: the caller's (H, W, 3) float64
        C-contiguous array to fill (the reference renders into the Bimage it was given, render_command.ml:65)."""
        p = render_params(width, height, samples_per_pixel, max_bounces, **kw)
        if out is None:
            out = np.zeros((height, width, 3))
        elif out.shape != (height, width, 3) or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float64 array of shape (height, width, 3)")
        st = abi.Stats()
        cb = PROGRESS_FN(lambda user, n: progress(n)) if progress else None
        _check(lib().ptx_render(self._h, C.byref(p), _dp(out), C.byref(st), C.cast(cb, C.c_void_p) if cb else None, None))
        return out, stats_dict(st)

    def pin_image(self, image):
        """ptx_image_pin: page-lock the caller's (H, W, 3) float64 image for as long as renders go into it (one DMA per frame
        instead of a staged copy).  The image must stay alive until unpin_image() / close()."""
        if image.dtype != np.float64 or not image.flags["C_CONTIGUOUS"]:
            raise ValueError("image must be a C-contiguous float64 array")
        _check(lib().ptx_image_pin(self._h, _dp(image), image.size))
        self._pinned_image = image  # keeps it mapped while the registration exists

    def unpin_image(self):
        _check(lib().ptx_image_unpin(self._h))
        self._pinned_image = None

    def render_raw_device(self, params, d_raw_ptr, stream=None):
        """ptx_render_raw_device: raw per-pixel sums for this rank's rows into DEVICE memory."""
        st = abi.Stats()
        _check(lib().ptx_render_raw_device(self._h, C.byref(params), C.c_void_p(d_raw_ptr),
                                           C.c_void_p(stream) if stream else None, C.byref(st)))
        return stats_dict(st)

    def trace_samples(self, width, height, samples_per_pixel, max_bounces, xs, ys, passes, count_work=False):
        xs = np.ascontiguousarray(xs, dtype=np.int32)
        ys = np.ascontiguousarray(ys, dtype=np.int32)
        passes = np.ascontiguousarray(passes, dtype=np.int32)
        p = render_params(width, height, samples_per_pixel, max_bounces, count_work=count_work)
        out = np.zeros((len(xs), 3))
        st = abi.Stats()
        _check(lib().ptx_trace_samples(self._h, C.byref(p), len(xs), _ip(xs), _ip(ys), _ip(passes), _dp(out), C.byref(st)))
        return out, stats_dict(st)

    def intersect_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float64)
        d = np.ascontiguousarray(directions, dtype=np.float64)
        n = o.shape[0]
        t = np.zeros(n)
        prim = np.zeros(n, dtype=np.int32)
        st = abi.Stats()
        _check(lib().ptx_intersect_rays(self._h, n, _dp(o), _dp(d), _dp(t), _ip(prim), C.byref(st)))
        return t, prim, stats_dict(st)

    def ppm_render(self, params, lights):
        """ptx_ppm_render: Progressive_photon_map.Make(Scene).go without the gamma / PNG step -> img_sum (H, W, 3)."""
        arr = (abi.Light * len(lights))(*lights)
        img = np.zeros((params.height, params.width, 3))
        st = abi.PpmStats()
        _check(lib().ptx_ppm_render(self._h, C.byref(params), arr, len(lights), _dp(img), C.byref(st), None, None))
        return img, {f: getattr(st, f) for f, _ in abi.PpmStats._fields_}

    def close(self):
        if getattr(self, "_h", None):
            lib().ptx_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_multi(scenes, width, height, samples_per_pixel, max_bounces, progress=None, **kw):
    """ptx_render_multi: one image over several scene replicas (one per GPU) inside this process."""
    p = render_params(width, height, samples_per_pixel, max_bounces, **kw)
    out = np.zeros((height, width, 3))
    st = abi.Stats()
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    cb = PROGRESS_FN(lambda user, n: progress(n)) if progress else None
    _check(lib().ptx_render_multi(arr, len(scenes), C.byref(p), _dp(out), C.byref(st),
                                  C.cast(cb, C.c_void_p) if cb else None, None))
    return out, stats_dict(st)


def film_resolve_banded_device(device, width, height, samples_per_pixel, d_gathered_ptr, n_ranks, band_rows, pad_rows,
                               d_rgb_ptr, stream=None, wait=True):
    """wait=False: ptx_film_resolve_banded_queue -- the pass is queued on `stream`, the call does not wait for it."""
    fn = lib().ptx_film_resolve_banded_device if wait else lib().ptx_film_resolve_banded_queue
    _check(fn(device, width, height, samples_per_pixel, C.c_void_p(d_gathered_ptr), n_ranks, band_rows, pad_rows,
              C.c_void_p(d_rgb_ptr), C.c_void_p(stream) if stream else None))


def film_resolve_device(device, width, height, samples_per_pixel, d_raw_ptr, d_rgb_ptr, stream=None):
    _check(lib().ptx_film_resolve_device(device, width, height, samples_per_pixel, C.c_void_p(d_raw_ptr),
                                         C.c_void_p(d_rgb_ptr), C.c_void_p(stream) if stream else None))


def local_rows(params):
    return lib().ptx_local_rows(C.byref(params))


def global_row(params, local_row):
    return lib().ptx_global_row(C.byref(params), local_row)


def lds_sample(dimension, offsets, dims, device=0):
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    out = np.zeros(len(offsets))
    _check(lib().ptx_lds_sample(device, dimension, len(offsets), _ip(offsets), _ip(dims), _dp(out)))
    return out


MATH_FN = {"hypot": 0, "sin": 1, "cos": 2, "acos": 3, "atan2": 4, "pow5": 5, "sqrt": 6, "div": 7, "fma": 8,
           "rnorm3": 9, "rnorm_frame": 10, "sqrt_nonneg": 11, "rcp_mid": 12, "div_mid": 13, "sqrt_mid": 14}


def math_eval(fn, a, b=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    out = np.zeros_like(a)
    _check(lib().ptx_math_eval(device, MATH_FN[fn], a.size, _dp(a), _dp(bb) if bb is not None else None, _dp(out)))
    return out
