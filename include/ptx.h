/* ptx.h -- C ABI of the MI355X path-tracing integrator (libptx_hip.so).
 *
 * Drop-in boundary for ONE hot path of dalev/path-tracer-ocaml: the per-pixel
 * sampling integrator (Integrator.create / Integrator.render,
 * path_tracer/src/integrator.mli:4-16, driven by Render_command.Make.run,
 * render_command/src/render_command.ml:64-109).
 *
 * The reference's only FFI is per-leaf (one ray x one <=16-sphere packet):
 *   external spheres_intersect_native : coords -> float -> float -> Ray.t -> float_ref -> int
 *   external leaf_size : unit -> int          (shirley_spheres/bin/main.ml:162-172,
 *                                              sphere-intersect-rs/src/lib.rs:15-18,53-76)
 * A GPU cannot be called once per BVH leaf per ray, and the reference's
 * intersect / background / do_scatter are opaque OCaml closures
 * (render_command.mli:18-22), so the accelerated boundary sits one level up:
 * the host hands over a DECLARATIVE scene (what main.ml builds before it calls
 * Render_cmd.run) and gets the post-gamma f64 framebuffer back -- exactly what
 * Integrator.render leaves in its Bimage (integrator.ml:130-156).
 *
 * Conventions: plain pointers and sizes; the caller owns every input and output
 * buffer; the library copies what it keeps.  All geometry is ALREADY in camera
 * space (the reference pre-transforms it: shirley_spheres/bin/main.ml:258-260,
 * ganesha/bin/main.ml:74-79).  No exceptions cross the boundary: every call
 * returns 0 / a handle on success and a negative code / NULL on failure, with
 * ptx_last_error() giving the message.  One render per handle at a time.
 */
#ifndef PTX_H
#define PTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_ABI_VERSION 6

/* ---- materials: Material.t, path_tracer/src/material.ml:3-14 ---- */
#define PTX_MAT_LAMBERTIAN 0 /* Lambertian of Texture.t */
#define PTX_MAT_METAL 1      /* Metal of Texture.t (no fuzz) */
#define PTX_MAT_DIELECTRIC 2 /* Dielectric {index; index_inv = 1/index} */

typedef struct ptx_material {
  int32_t kind;
  int32_t texture; /* index into textures (Lambertian / Metal) */
  double index;    /* Dielectric only */
  /* Hit.emit slot (hit.ml:5).  The reference's Material.emit is constant black
   * (material.ml:59); non-zero values are the documented emitter extension used
   * by the cornell-box configuration only. */
  double emit[3];
} ptx_material;

/* ---- textures: Texture.solid / Texture.checker, path_tracer/src/texture.ml:16-31 ---- */
#define PTX_TEX_SOLID 0
#define PTX_TEX_CHECKER 1

typedef struct ptx_texture {
  int32_t kind;
  int32_t width, height; /* checker ~width ~height (the code uses width-1, height-1) */
  int32_t reserved;
  double even[3]; /* solid colour, or the checker's "even" solid */
  double odd[3];
} ptx_texture;

/* ---- camera: the four fields Camera.ray reads, path_tracer/src/camera.ml:50-53,93-102 ---- */
typedef struct ptx_camera {
  double lower_left_x, lower_left_y, view_x, view_y;
} ptx_camera;

/* ---- background: Scene.background closure made declarative ---- */
#define PTX_BG_BLACK 0
#define PTX_BG_SKY 1 /* lerp t horizon zenith, t = .5*(normalize(dir).y + 1); main.ml:104-110 */

typedef struct ptx_background {
  int32_t kind;
  int32_t reserved;
  double horizon[3]; /* Color.white in the reference */
  double zenith[3];  /* escape_color (.5,.7,1) */
} ptx_background;

/* ---- leaf flavours: Shape_tree.Leaf implementations ---- */
#define PTX_LEAF_SIMD 0  /* Simd_leaf: <=16-sphere SoA packets, Rust x86 arithmetic (lib.rs:102-178) */
#define PTX_LEAF_ARRAY 1 /* Array_leaf: linear scan, Sphere.intersect / Triangle.intersect */

typedef struct ptx_scene_desc {
  /* spheres (Sphere.t: centre, radius, material), SoA like Simd_leaf.coords */
  int32_t n_spheres;
  const double* sphere_x;
  const double* sphere_y;
  const double* sphere_z;
  const double* sphere_r;
  const int32_t* sphere_material;

  /* triangle mesh (ganesha Mesh.t: SoA vertices + index triples; cornell Face.t) */
  int32_t n_vertices;
  const double* vertex_x;
  const double* vertex_y;
  const double* vertex_z;
  int32_t n_triangles;
  const int32_t* tri_indices;  /* 3 per triangle: a, b, c */
  const double* tri_uv;        /* 6 per triangle: (ua,va),(ub,vb),(uc,vc) */
  const int32_t* tri_material; /* 1 per triangle */

  /* triangles tested BEFORE the tree, clipping t_max (ganesha Floor, main.ml:205-298) */
  int32_t n_floor_triangles;
  const double* floor_vertices; /* 9 per triangle */
  const double* floor_uv;       /* 6 per triangle */
  const int32_t* floor_material;

  int32_t n_materials;
  const ptx_material* materials;
  int32_t n_textures;
  const ptx_texture* textures;

  ptx_camera camera;
  ptx_background background;

  /* Shape_tree.create ?num_bins (default 32), Leaf.length_cutoff, leaf flavour.
   * The tree is built over [triangles in order] @ [spheres in order]
   * (cornell-box/bin/main.ml:213-218). */
  int32_t leaf_kind;
  int32_t length_cutoff; /* 16 for SIMD (lib.rs:13), 4 / 2 / 8 for the array leaves */
  int32_t num_bins;      /* 0 -> 32 */
  int32_t reserved;      /* BVH builder: 0 auto (GPU for >= 4096 primitives, else host), 1 host, 2 GPU -- same tree */
} ptx_scene_desc;

/* ---- render parameters: Render_command.Args.t (render_command.ml:7-14) ---- */
typedef struct ptx_render_params {
  int32_t width, height;
  int32_t samples_per_pixel;
  int32_t max_bounces;
  /* Image rows are dealt to ranks in horizontal bands: rank `band_first` of
   * `band_step` renders bands band_first, band_first+band_step, ... of
   * `band_rows` rows each.  band_step <= 1 renders the whole image. */
  int32_t band_rows;
  int32_t band_first;
  int32_t band_step;
  /* 1: also count BVH nodes tested / primitive slots tested (slower; parity vs
   * the instrumented oracle and the algorithmic-bytes figure). */
  int32_t count_work;
  /* 1: bracket every kernel launch with HIP events (per-kernel ms in ptx_stats). */
  int32_t time_kernels;
  /* samples per wavefront batch in passes; 0 = library default */
  int32_t passes_per_batch;
  /* ptx_render only: GPUs of this node to spread the image over, inside this one process (SURVEY section 8 B3 / E;
   * the reference's Domainslib pool over tiles, integrator.ml:136-151, becomes one host thread per device over
   * interleaved row bands).  0 or 1 = the scene's own device only.  Devices used: the scene's, then the following
   * ordinals (mod ptx_device_count()); replicas of the scene are made on first use and kept with the handle. */
  int32_t n_gpus;
  /* PTX_RENDER_ASYNC (ptx_render_raw_device only): return as soon as the frame is QUEUED on `stream` -- the caller's next use
   * of the buffer must be ordered after it on that stream (as a following ptx_* call or a collective on it is), or wait for
   * the stream.  `stats` then carries no timings or counters; count_work / time_kernels renders always wait.  For hosts
   * that pipeline frames (one rank of a multi-GPU job: the next frame is queued while this one's bands travel). */
  int32_t flags;
} ptx_render_params;
#define PTX_RENDER_ASYNC 1

#define PTX_KERNEL_GENERATE 0
#define PTX_KERNEL_TRACE 1
#define PTX_KERNEL_SHADE 2
#define PTX_KERNEL_ACCUM 3
#define PTX_KERNEL_FILM 4
#define PTX_KERNEL_BOUNCE 5 /* trace + shade of one bounce in ONE launch (scenes whose tree fits LDS); then TRACE / SHADE count only what ran separately */
#define PTX_N_KERNELS 6

typedef struct ptx_stats {
  int64_t samples;       /* W * rows * spp actually rendered */
  int64_t segments;      /* rays traced (Scene.intersect calls, integrator.ml:35) */
  int64_t nodes_tested;  /* Bbox.is_hit evaluations (shape_tree.ml:203), if count_work */
  int64_t prims_tested;  /* leaf slots tested incl. NaN padding, if count_work */
  int64_t floor_tested;  /* floor triangle tests, if count_work */
  double render_ms;      /* host wall time of the call (after the final sync) */
  double kernel_ms[PTX_N_KERNELS]; /* summed HIP-event time per kernel kind, if time_kernels */
  int64_t kernel_launches[PTX_N_KERNELS];
  int32_t tree_nodes, tree_depth, tree_leaves, leaf_slots;
  double build_ms; /* BVH build + upload at ptx_scene_create */
  int32_t traversal_in_lds; /* 1: tree + leaf packets fit the per-workgroup LDS copy; 0: traversed from HBM / L2 */
  int32_t bvh_built_on_gpu; /* 1: csrc/bvh_build_gpu.inc built the tree, 0: the host builder (same tree) */
  /* if count_work: Bbox.is_hit evaluations the binary32 filter in front of the binary64 slab test could NOT decide (they
   * then ran the reference's binary64 arithmetic, bbox.ml:40-56), and the wave steps that entered that branch.  Both
   * are 0 for walks that never use the filter; the parity tests assert the branch is exercised. */
  int64_t filter_undecided;
  int64_t filter_fallback_steps;
  /* ptx_render with n_gpus > 1 / ptx_render_multi: how each replica's raw sums reached the root device --
   * peer_copies = device-to-device with peer access enabled (xGMI), staged_copies = hipMemcpyPeer without peer
   * access (the runtime stages through host memory).  Replicas that share the root's device count in neither. */
  int32_t peer_copies;
  int32_t staged_copies;
  /* if count_work: launches of the one-kernel-per-bounce path that found their input small enough (PTX_SOLO_ENTRIES) to run all
   * remaining bounces of their batch by themselves; the batch's later launches return at once (ABI 6) */
  int32_t solo_launches;
  int32_t reserved_stats;
} ptx_stats;

/* ---- progressive photon mapping (progressive-photon-map/src/progressive_photon_map.ml) ---- */
#define PTX_LIGHT_POINT 0 /* Light.create_point ~position ~power ~color  (:64-84) */
#define PTX_LIGHT_SPOT 1  /* Light.create_spot ~position ~direction ~color ~power (:86-110) */

typedef struct ptx_light {
  int32_t kind;
  int32_t reserved;
  double position[3];  /* camera space, like every other coordinate */
  double direction[3]; /* spot only (not normalised by the caller) */
  double color[3];     /* BEFORE the power scaling */
  double power;
} ptx_light;

/* Progressive_photon_map.Args.t (:7-16) */
typedef struct ptx_ppm_params {
  int32_t width, height;
  int32_t iterations;   /* default 10 */
  int32_t max_bounces;  /* default 4 */
  int32_t photon_count; /* default 75000 */
  int32_t reserved;
  double alpha;         /* default 2/3 */
} ptx_ppm_params;

typedef struct ptx_ppm_stats {
  int64_t photons_stored;  /* Photon_map.length summed over iterations */
  int64_t photon_rays;     /* segments traced from the lights */
  int64_t eye_rays;        /* segments traced from the camera */
  int64_t neighbors;       /* photons accepted by the radiance estimates */
  double photon_ms, build_ms, gather_ms, total_ms;
  double last_radius;
} ptx_ppm_stats;

typedef struct ptx_scene ptx_scene; /* opaque */

typedef void (*ptx_progress_fn)(void* user, int64_t pixels_done);

/* ---- entry points ---- */
int32_t ptx_version(void);
/* replaces `leaf_size : unit -> int` (lib.rs:15-18) */
int32_t ptx_leaf_size(void);
const char* ptx_last_error(void);

/* number of HIP devices visible; negative on error */
int32_t ptx_device_count(void);

/* Builds the BVH on the host exactly as Shape_tree.create does (shape_tree.ml:252-263),
 * flattens it and uploads everything to HIP device `device`.
 * device == -1 builds a HOST-ONLY scene (nothing uploaded): only ptx_scene_tree / ptx_scene_stats /
 * ptx_scene_destroy accept it; every compute entry point returns an error (there is no CPU fallback). */
ptx_scene* ptx_scene_create(const ptx_scene_desc* desc, int32_t device);
void ptx_scene_destroy(ptx_scene* scene);
/* copies the build statistics (tree_* and build_ms fields) */
int32_t ptx_scene_stats(const ptx_scene* scene, ptx_stats* out);

/* Replaces Integrator.render (integrator.ml:130-156) for the whole image on one GPU:
 * rgb_out is HOST memory, width*height*3 doubles, index (y*W + x)*3 + c, y = 0 at the
 * top, post-gamma -- the contents of the reference's Bimage after render. */
int32_t ptx_render(ptx_scene* scene, const ptx_render_params* params, double* rgb_out,
                   ptx_stats* stats, ptx_progress_fn progress, void* user);

/* The same render over several GPUs of one node inside ONE process, for hosts that are a single process (the OCaml
 * executable, the C++ CLI): scenes[k] is a replica of the scene on the k-th device (ptx_scene_replicate; scenes[0]
 * may be the original), image rows are dealt in interleaved bands of params->band_rows rows (0 -> 8), one host
 * thread per scene renders its bands (integrator.ml:138-146), the raw sums travel to scenes[0]'s device as one
 * peer-to-peer copy per replica (xGMI), and the film pass runs there.  n_scenes = 1 is ptx_render bit for bit; any
 * n_scenes gives bit-identical raw sums (the sampler offset depends only on the global pixel, integrator.ml:98).
 * progress is invoked on the CALLING thread only.  Scenes may share a device (tests on a one-GPU box). */
int32_t ptx_render_multi(ptx_scene* const* scenes, int32_t n_scenes, const ptx_render_params* params,
                         double* rgb_out, ptx_stats* stats, ptx_progress_fn progress, void* user);

/* A replica of `scene` on HIP device `device`: the flattened tree / slots / materials the original kept on the host
 * are uploaded again (no second BVH build).  Independent handle: destroy it with ptx_scene_destroy. */
ptx_scene* ptx_scene_replicate(const ptx_scene* scene, int32_t device);

/* Frees the calling thread's cached GPU-BVH-builder buffers (they otherwise live as long as the thread). */
void ptx_release_workspaces(void);

/* Optional: page-lock the caller's framebuffer for as long as it will be rendered into.  ptx_render / ptx_render_multi into
 * exactly this image (rgb_out inside [image, image + n_doubles)) then fill it with ONE DMA instead of copying through a staging
 * buffer (1080p: ~1.1 ms instead of ~2.8).  The reference's image lives as long as the run (the Bimage of
 * render_command/src/render_command.ml:64-70), which is the caller this is for.  CONTRACT: the image must stay mapped until
 * ptx_image_unpin or ptx_scene_destroy -- a DMA into a registration whose pages were unmapped aborts the process -- which is
 * why the library never pins an image behind the caller's back.  One pinned image per handle (pinning another releases the
 * first).  Returns 0, or an error code (the caller may ignore it: renders then take the staged copy). */
int32_t ptx_image_pin(ptx_scene* scene, double* image, int64_t n_doubles);
int32_t ptx_image_unpin(ptx_scene* scene);

/* Device-resident form, for one rank of a multi-GPU job and for benchmarking with no
 * PCIe traffic in the timed region.  d_raw_out is DEVICE memory holding this rank's
 * rows compactly: ptx_local_rows(params) * width * 3 doubles of raw per-pixel radiance
 * sums (no filter, no gamma).  `stream` is a hipStream_t (NULL = default stream).
 * The call returns after the work is enqueued AND complete (it syncs the stream). */
int32_t ptx_local_rows(const ptx_render_params* params);
/* global image row of local row k (or -1) */
int32_t ptx_global_row(const ptx_render_params* params, int32_t local_row);
int32_t ptx_render_raw_device(ptx_scene* scene, const ptx_render_params* params,
                              double* d_raw_out, void* stream, ptx_stats* stats);

/* Film: 3x3 binomial reconstruction (Filter_kernel.Binomial order 5 radius 1,
 * filter_kernel.ml:49-85) with the reference's unnormalised image border
 * (integrator.ml:114-128), then sqrt(v / spp) (integrator.ml:152-154).
 * d_raw_full: DEVICE, height*width*3 raw sums in image row order; d_rgb_out: DEVICE. */
int32_t ptx_film_resolve_device(int32_t device, int32_t width, int32_t height,
                                int32_t samples_per_pixel, const double* d_raw_full,
                                double* d_rgb_out, void* stream);

/* The film pass reading the GATHERED multi-rank layout in place: d_gathered is DEVICE memory
 * [n_ranks][pad_rows][width][3], slice r = rank r's compact rows exactly as ptx_render_raw_device wrote them with
 * band_first = r, band_step = n_ranks, band_rows (pad_rows >= every rank's ptx_local_rows).  No un-permute copy. */
int32_t ptx_film_resolve_banded_device(int32_t device, int32_t width, int32_t height, int32_t samples_per_pixel,
                                       const double* d_gathered, int32_t n_ranks, int32_t band_rows, int32_t pad_rows,
                                       double* d_rgb_out, void* stream);
/* The same pass QUEUED on `stream` without waiting for it (the twin of PTX_RENDER_ASYNC: a rank that pipelines frames). */
int32_t ptx_film_resolve_banded_queue(int32_t device, int32_t width, int32_t height, int32_t samples_per_pixel,
                                      const double* d_gathered, int32_t n_ranks, int32_t band_rows, int32_t pad_rows,
                                      double* d_rgb_out, void* stream);

/* Per-sample radiance for explicit (x, y, pass) triples -- the value Integrator's
 * trace_path returns (integrator.ml:106).  Host in / host out, n*3 doubles.
 * Used by the parity tests (bit-exact against the oracle). */
int32_t ptx_trace_samples(ptx_scene* scene, const ptx_render_params* params, int64_t n,
                          const int32_t* xs, const int32_t* ys, const int32_t* passes,
                          double* rgb_out, ptx_stats* stats);

/* Closest-hit queries on the device for explicit rays (Scene.intersect,
 * shirley_spheres/bin/main.ml:273-277): n rays (origin, direction: 3 doubles each),
 * outputs t_hit (DBL_MAX-as-miss is NOT used: prim_out = -1 on a miss) and the index of
 * the primitive in the build list ([triangles] @ [spheres]; floor triangles are
 * n_triangles + n_spheres + i). */
int32_t ptx_intersect_rays(ptx_scene* scene, int64_t n, const double* origins,
                           const double* directions, double* t_out, int32_t* prim_out,
                           ptx_stats* stats);

/* Replaces Progressive_photon_map.Make(Scene).go (:420-451) up to, but not including, the per-iteration
 * gamma + PNG write: img_sum_out (HOST, width*height*3, row 0 = top as Bimage stores it) receives the
 * reference's img_sum after `iterations` iterations, i.e. the sum over iterations of estimate / photon_count.
 * Scene.bbox is the bounding box of the scene's tree; the eye pass and the photon pass use the scene's camera.
 * iteration_cb (optional) is called on the calling thread after every iteration with the running img_sum. */
typedef void (*ptx_ppm_iteration_fn)(void* user, int32_t iteration, double radius, int64_t photon_map_length,
                                     const double* img_sum);
int32_t ptx_ppm_render(ptx_scene* scene, const ptx_ppm_params* params, const ptx_light* lights, int32_t n_lights,
                       double* img_sum_out, ptx_ppm_stats* stats, ptx_ppm_iteration_fn iteration_cb, void* user);

/* Flattened tree, for inspection / parity of the builder: returns the node count and,
 * if the pointers are non-NULL, copies per node: bbox (6 doubles: min xyz, max xyz),
 * and 4 ints: {is_leaf, axis (0,1,2; -1 for leaves), lhs | first slot, rhs | slot count}.
 * prim_order (optional) receives, per leaf slot, the build-list primitive index or -1
 * for a NaN padding slot. */
int32_t ptx_scene_tree(const ptx_scene* scene, double* bbox_out, int32_t* info_out,
                       int32_t node_capacity, int32_t* prim_order_out, int32_t slot_capacity);

/* Sampler: Low_discrepancy_sequence.create / get evaluated on the device
 * (low_discrepancy_sequence.ml:27-36): out[i] = get ~offset:offsets[i] ~dimension:dims[i]
 * for a sampler of `dimension` dimensions. */
int32_t ptx_lds_sample(int32_t device, int32_t dimension, int64_t n, const int32_t* offsets,
                       const int32_t* dims, double* out);

/* pt_math.h functions evaluated on the device, for the host==device bit-identity test.
 * fn: 0 hypot(a,b) 1 sin(a) 2 cos(a) 3 acos(a) 4 atan2(a,b) 5 pow5(a) 6 sqrt(a) 7 a/b
 *     8 fma(a,b,b) */
int32_t ptx_math_eval(int32_t device, int32_t fn, int64_t n, const double* a, const double* b,
                      double* out);

/* Diagnostic (parity tooling, tools/diag_scatter.py): the state of the listed paths after ONE segment (camera ray,
 * trace, shade of bounce 0) -- alive_out[i] = 1 with ray_out[6i..] = (origin, direction) and attn_out[3i..] when
 * sample i scattered, 0 when it terminated.  This is how the hipcc miscompile of the dielectric branch was isolated
 * (DESIGN.md section 2).  Not part of the rendering path; needs max_bounces >= 2. */
int32_t ptx_debug_first_scatter(ptx_scene* scene, const ptx_render_params* params, int64_t n, const int32_t* xs,
                                const int32_t* ys, const int32_t* passes, double* ray_out, double* attn_out,
                                int32_t* alive_out);

#ifdef __cplusplus
}
#endif
#endif /* PTX_H */
