/* ptx_stubs.c -- OCaml C stubs for libptx_hip.so (include/ptx.h), bound by ptx.ml's `external`s.
 *
 * Same kind of binding as the reference's only FFI, `external spheres_intersect_native ... [@@noalloc]` and
 * `external leaf_size` (shirley_spheres/bin/main.ml:162-172, implemented in sphere-intersect-rs/src/lib.rs:15-18,53-76
 * and linked by shirley_spheres/bin/dune:8-33): plain `external` C symbols, OCaml values unpacked by hand, pointers
 * BORROWED for the duration of the call -- but one call per render instead of one per BVH leaf per ray.
 *
 * Needs the OCaml runtime headers (<caml/...>), so it is not compiled into anything in this repository's image; the
 * marshalling it performs lives in ptx_ml_marshal.h and is exercised without OCaml by tests/test_ocaml_binding.py, which
 * also type-checks THIS file (gcc -fsyntax-only) against minimal declarations of the runtime's API (tests/c/mock_caml).
 *
 * Value conventions:
 *   floatarray          Double_array_tag block: (double*) v is the flat data, Wosize_val(v) / Double_wosize the length
 *                       (the reference reads `coords` the same way: header >> 10 words, lib.rs:26-36)
 *   Bigarray.Array1     Caml_ba_data_val(v), Caml_ba_array_val(v)->dim[0]
 *   int                 Long_val / Val_long
 *   scene handle        custom block holding the ptx_scene* and an in-use count, finalised with ptx_scene_destroy.  The count
 *                       is raised for the duration of a render (which runs with the runtime lock released): Ptx.scene_destroy
 *                       from another thread or domain meanwhile raises Failure instead of freeing what the render reads
 *   errors              caml_failwith (ptx_last_error ()): no error code ever reaches OCaml unraised
 *   callbacks           update_progress / on_iteration run on the CALLING thread (the library's contract).  The runtime
 *                       lock is RELEASED for the duration of the render (other domains and threads keep running; the
 *                       reference's own integrator leaves them running too) and re-taken around each callback.  A
 *                       callback that raises does not unwind through the library's C++ frames: the exception is caught
 *                       (caml_callback_exn), kept, no further callbacks are made, and it is re-raised after the library
 *                       call has returned.
 */
#define CAML_NAME_SPACE
#include <caml/alloc.h>
#include <caml/bigarray.h>
#include <caml/callback.h>
#include <caml/custom.h>
#include <caml/fail.h>
#include <caml/memory.h>
#include <caml/mlvalues.h>
#include <caml/threads.h>

#include <stdlib.h>

#include "ptx_ml_marshal.h"

typedef struct ptx_ml_handle {
  ptx_scene* scene;
  /* renders running on the scene right now (atomic: OCaml 5 domains run in parallel); + PTX_ML_EXCLUSIVE while image_pin /
   * image_unpin change what a render reads (the registered image): they need the scene to themselves */
  int in_use;
  /* the pinned image, kept alive from here: a generational global root in a malloc'ed cell (a custom block's own data may move
   * with the block, a root's address may not), NULL while nothing is pinned.  While it exists the Bigarray cannot be collected,
   * so neither a pin that outlives the caller's last reference nor the finaliser (whose order against the image's is not
   * defined) can leave the library holding a registration of freed memory. */
  value* pin_root;
} ptx_ml_handle;
#define PTX_ML_EXCLUSIVE (1 << 20)
#define Handle_val(v) ((ptx_ml_handle*)Data_custom_val(v))
#define Scene_val(v) (Handle_val(v)->scene)

static void ptx_ml_drop_pin_root(ptx_ml_handle* h) {
  if (h->pin_root) {
    caml_remove_generational_global_root(h->pin_root);
    free(h->pin_root);
    h->pin_root = NULL;
  }
}
/* the finaliser: runs only when nothing reaches the handle any more, so no render can be using it.  ptx_scene_destroy ends the
 * pin (the image is still alive: the root below is dropped after it) */
static void ptx_ml_scene_finalize(value v) {
  if (Scene_val(v)) {
    ptx_scene_destroy(Scene_val(v));
    Scene_val(v) = NULL;
  }
  ptx_ml_drop_pin_root(Handle_val(v));
}
/* a render takes the scene for the time the runtime lock is released; NULL = already destroyed, or being re-pinned (*busy) */
static ptx_scene* ptx_ml_scene_acquire(value handle, int* busy) {
  ptx_ml_handle* h = Handle_val(handle);
  *busy = 0;
  const int n = __atomic_add_fetch(&h->in_use, 1, __ATOMIC_ACQ_REL);
  ptx_scene* s = __atomic_load_n(&h->scene, __ATOMIC_ACQUIRE);
  if (n >= PTX_ML_EXCLUSIVE) { /* image_pin / image_unpin is at work on another thread or domain */
    *busy = 1;
    s = NULL;
  }
  if (!s) __atomic_sub_fetch(&h->in_use, 1, __ATOMIC_ACQ_REL);
  return s;
}
static void ptx_ml_scene_release(value handle) { __atomic_sub_fetch(&Handle_val(handle)->in_use, 1, __ATOMIC_ACQ_REL); }
/* image_pin / image_unpin: the scene with NO render running, and none starting until ptx_ml_scene_release_exclusive.
 * 1 = taken; 0 = a render (or another pin) is running: nothing changed */
static int ptx_ml_scene_acquire_exclusive(value handle) {
  ptx_ml_handle* h = Handle_val(handle);
  if (__atomic_add_fetch(&h->in_use, PTX_ML_EXCLUSIVE, __ATOMIC_ACQ_REL) != PTX_ML_EXCLUSIVE) {
    __atomic_sub_fetch(&h->in_use, PTX_ML_EXCLUSIVE, __ATOMIC_ACQ_REL);
    return 0;
  }
  return 1;
}
static void ptx_ml_scene_release_exclusive(value handle) { __atomic_sub_fetch(&Handle_val(handle)->in_use, PTX_ML_EXCLUSIVE, __ATOMIC_ACQ_REL); }

static struct custom_operations ptx_ml_scene_ops = {
    "dalev.path_tracer.ptx_scene", ptx_ml_scene_finalize, custom_compare_default, custom_hash_default,
    custom_serialize_default,       custom_deserialize_default, custom_compare_ext_default, custom_fixed_length_default};

static const double* floatarray_data(value v) { return (const double*)v; }
static int32_t floatarray_length(value v) { return (int32_t)(Wosize_val(v) / Double_wosize); }
static int32_t int32_ba_length(value v) { return (int32_t)Caml_ba_array_val(v)->dim[0]; }
static const int32_t* int32_ba_data(value v) { return (const int32_t*)Caml_ba_data_val(v); }

/* external leaf_size : unit -> int = "ptx_ml_leaf_size"   (replaces `leaf_size`, lib.rs:15-18) */
CAMLprim value ptx_ml_leaf_size(value unit) {
  (void)unit;
  return Val_long(ptx_leaf_size());
}

/* external device_count : unit -> int = "ptx_ml_device_count" */
CAMLprim value ptx_ml_device_count(value unit) {
  (void)unit;
  return Val_long(ptx_device_count());
}

/* external scene_create_flat : flat -> int -> scene = "ptx_ml_scene_create_stub"
 * flat = { xs; ys; zs; rs; sphere_material; materials; textures; camera; background; leaf_kind; length_cutoff;
 *          vertex_x; vertex_y; vertex_z; tri_indices; tri_uv; tri_material; floor_vertices; floor_uv; floor_material }
 * (field order of Ptx.flat in ptx.ml) */
CAMLprim value ptx_ml_scene_create_stub(value flat, value device) {
  CAMLparam2(flat, device);
  CAMLlocal1(handle);
  ptx_ml_flat f;
  memset(&f, 0, sizeof f);
  f.xs = floatarray_data(Field(flat, 0));
  f.ys = floatarray_data(Field(flat, 1));
  f.zs = floatarray_data(Field(flat, 2));
  f.rs = floatarray_data(Field(flat, 3));
  f.n_spheres = floatarray_length(Field(flat, 0));
  if (floatarray_length(Field(flat, 1)) != f.n_spheres || floatarray_length(Field(flat, 2)) != f.n_spheres ||
      floatarray_length(Field(flat, 3)) != f.n_spheres || int32_ba_length(Field(flat, 4)) != f.n_spheres)
    caml_invalid_argument("Ptx.scene_create: sphere arrays differ in length");
  f.sphere_material = int32_ba_data(Field(flat, 4));
  f.materials = floatarray_data(Field(flat, 5));
  f.n_materials = floatarray_length(Field(flat, 5)) / 6;
  f.textures = floatarray_data(Field(flat, 6));
  f.n_textures = floatarray_length(Field(flat, 6)) / 9;
  if (floatarray_length(Field(flat, 7)) != 4 || floatarray_length(Field(flat, 8)) != 7)
    caml_invalid_argument("Ptx.scene_create: camera needs 4 floats, background 7");
  f.camera = floatarray_data(Field(flat, 7));
  f.background = floatarray_data(Field(flat, 8));
  f.leaf_kind = (int32_t)Long_val(Field(flat, 9));
  f.length_cutoff = (int32_t)Long_val(Field(flat, 10));
  /* the triangle mesh (ganesha Mesh.t / cornell-box Face.t) */
  f.n_vertices = floatarray_length(Field(flat, 11));
  if (floatarray_length(Field(flat, 12)) != f.n_vertices || floatarray_length(Field(flat, 13)) != f.n_vertices)
    caml_invalid_argument("Ptx.scene_create: vertex arrays differ in length");
  f.vertex_x = floatarray_data(Field(flat, 11));
  f.vertex_y = floatarray_data(Field(flat, 12));
  f.vertex_z = floatarray_data(Field(flat, 13));
  f.n_triangles = int32_ba_length(Field(flat, 14)) / 3;
  if (int32_ba_length(Field(flat, 14)) != 3 * f.n_triangles || floatarray_length(Field(flat, 15)) != 6 * f.n_triangles ||
      int32_ba_length(Field(flat, 16)) != f.n_triangles)
    caml_invalid_argument("Ptx.scene_create: a mesh needs 3 indices, 6 texture coordinates and 1 material per triangle");
  f.tri_indices = int32_ba_data(Field(flat, 14));
  f.tri_uv = floatarray_data(Field(flat, 15));
  f.tri_material = int32_ba_data(Field(flat, 16));
  /* triangles tested before the tree (ganesha Floor) */
  f.n_floor_triangles = int32_ba_length(Field(flat, 19));
  if (floatarray_length(Field(flat, 17)) != 9 * f.n_floor_triangles || floatarray_length(Field(flat, 18)) != 6 * f.n_floor_triangles)
    caml_invalid_argument("Ptx.scene_create: a floor triangle needs 9 coordinates and 6 texture coordinates");
  f.floor_vertices = floatarray_data(Field(flat, 17));
  f.floor_uv = floatarray_data(Field(flat, 18));
  f.floor_material = int32_ba_data(Field(flat, 19));
  /* no OCaml allocation between reading the pointers above and the end of ptx_ml_scene_create: nothing can move
   * (the runtime lock stays held here: the floatarrays live in the OCaml heap) */
  ptx_scene* s = ptx_ml_scene_create(&f, (int32_t)Long_val(device));
  if (!s) caml_failwith(ptx_ml_scene_create_error());
  handle = caml_alloc_custom(&ptx_ml_scene_ops, sizeof(ptx_ml_handle), 0, 1);
  Handle_val(handle)->scene = s;
  Handle_val(handle)->in_use = 0;
  Handle_val(handle)->pin_root = NULL;
  CAMLreturn(handle);
}

/* external scene_destroy : scene -> unit = "ptx_ml_scene_destroy_stub"   (optional: the finaliser does the same) */
CAMLprim value ptx_ml_scene_destroy_stub(value handle) {
  ptx_ml_handle* h = Handle_val(handle);
  /* take the scene out of the handle first, then look at the count: a render that starts after this sees NULL, one that
   * started before it is counted */
  ptx_scene* s = __atomic_exchange_n(&h->scene, NULL, __ATOMIC_ACQ_REL);
  if (!s) return Val_unit;
  if (__atomic_load_n(&h->in_use, __ATOMIC_ACQUIRE) > 0) {
    __atomic_store_n(&h->scene, s, __ATOMIC_RELEASE); /* put it back: the finaliser or a later call frees it */
    caml_failwith("Ptx.scene_destroy: a render is running on this scene");
  }
  ptx_scene_destroy(s); /* (ends a pin) */
  ptx_ml_drop_pin_root(h);
  return Val_unit;
}

/* external scene_tree_stats : scene -> int * int * int * int = "ptx_ml_scene_tree_stats_stub"   (depth, nodes, leaves, slots) */
CAMLprim value ptx_ml_scene_tree_stats_stub(value handle) {
  CAMLparam1(handle);
  CAMLlocal1(tuple);
  ptx_scene* s = Scene_val(handle);
  if (!s) caml_invalid_argument("Ptx.scene_tree_stats: scene already destroyed");
  ptx_stats st;
  if (ptx_scene_stats(s, &st) != 0) caml_failwith(ptx_last_error());
  tuple = caml_alloc_tuple(4);
  Store_field(tuple, 0, Val_long(st.tree_depth));
  Store_field(tuple, 1, Val_long(st.tree_nodes));
  Store_field(tuple, 2, Val_long(st.tree_leaves));
  Store_field(tuple, 3, Val_long(st.leaf_slots));
  CAMLreturn(tuple);
}

/* external image_pin : scene -> image -> unit = "ptx_ml_image_pin_stub" / external image_unpin : scene -> unit (ptx_image_pin).
 * Both change the registration a running render reads (its last step is the DMA into the image): they take the scene
 * exclusively and raise Failure while a render runs on another thread or domain; a render that starts meanwhile raises Failure
 * too.  The pinned Bigarray is kept alive from the handle (ptx_ml_handle.pin_root). */
CAMLprim value ptx_ml_image_pin_stub(value handle, value image) {
  CAMLparam2(handle, image);
  if (!Scene_val(handle)) caml_invalid_argument("Ptx.image_pin: scene already destroyed");
  if (!ptx_ml_scene_acquire_exclusive(handle)) caml_failwith("Ptx.image_pin: a render is running on this scene");
  ptx_ml_handle* h = Handle_val(handle); /* (nothing below allocates on the OCaml heap: the block does not move) */
  ptx_scene* s = __atomic_load_n(&h->scene, __ATOMIC_ACQUIRE);
  int32_t rc = -1;
  if (s) rc = ptx_image_pin(s, (double*)Caml_ba_data_val(image), (int64_t)Caml_ba_array_val(image)->dim[0]);
  if (s && rc == 0) {
    if (!h->pin_root) {
      h->pin_root = (value*)malloc(sizeof(value));
      if (h->pin_root) {
        *h->pin_root = image;
        caml_register_generational_global_root(h->pin_root);
      } else { /* no cell for the root: do not keep a pin nothing keeps alive */
        (void)ptx_image_unpin(s);
        rc = -1;
      }
    } else {
      caml_modify_generational_global_root(h->pin_root, image);
    }
  } else if (s) {
    ptx_ml_drop_pin_root(h); /* ptx_image_pin ends the previous pin before it tries the new one */
  }
  ptx_ml_scene_release_exclusive(handle);
  if (!s) caml_invalid_argument("Ptx.image_pin: scene already destroyed");
  if (rc != 0) caml_failwith(h->pin_root || rc != -1 ? ptx_last_error() : "Ptx.image_pin: out of memory");
  CAMLreturn(Val_unit);
}
CAMLprim value ptx_ml_image_unpin_stub(value handle) {
  CAMLparam1(handle);
  if (!ptx_ml_scene_acquire_exclusive(handle)) caml_failwith("Ptx.image_unpin: a render is running on this scene");
  ptx_ml_handle* h = Handle_val(handle);
  ptx_scene* s = __atomic_load_n(&h->scene, __ATOMIC_ACQUIRE);
  if (s) (void)ptx_image_unpin(s);
  ptx_ml_drop_pin_root(h);
  ptx_ml_scene_release_exclusive(handle);
  CAMLreturn(Val_unit);
}

/* What a callback trampoline needs: the closure (a GC root registered by the stub that owns this struct) and the first
 * exception a callback raised.  While `raised` is set no further callbacks are made. */
typedef struct ptx_ml_cb {
  value* closure;
  value* exn; /* GC root; Val_unit until a callback raises */
  int raised;
} ptx_ml_cb;

static void ptx_ml_progress(void* user, int64_t pixels_done) {
  ptx_ml_cb* cb = (ptx_ml_cb*)user;
  if (cb->raised) return;
  caml_acquire_runtime_system(); /* the render runs with the lock released */
  value r = caml_callback_exn(*cb->closure, Val_long(pixels_done));
  if (Is_exception_result(r)) {
    *cb->exn = Extract_exception(r);
    cb->raised = 1;
  }
  caml_release_runtime_system();
}

/* external render_flat : scene -> int -> int -> int -> int -> int -> image -> (int -> unit) -> unit
 *   = "ptx_ml_render_stub_bytecode" "ptx_ml_render_stub"
 * (scene, width, height, samples_per_pixel, max_bounces, gpus, Bimage data as a float64 Bigarray of W*H*3, update_progress)
 * replaces Integrator.create ... |> Integrator.render ~update_progress (render_command.ml:71-104) */
CAMLprim value ptx_ml_render_stub(value handle, value width, value height, value spp, value max_bounces, value gpus, value image,
                                  value update_progress) {
  CAMLparam5(handle, width, height, spp, max_bounces);
  CAMLxparam3(gpus, image, update_progress);
  CAMLlocal1(exn);
  const intnat w = Long_val(width), h = Long_val(height);
  if (Caml_ba_array_val(image)->dim[0] != w * h * 3) caml_invalid_argument("Ptx.render: image must hold width * height * 3 floats");
  int busy = 0;
  ptx_scene* s = ptx_ml_scene_acquire(handle, &busy);
  if (!s && busy) caml_failwith("Ptx.render: the scene's image is being pinned or unpinned on another thread");
  if (!s) caml_invalid_argument("Ptx.render: scene already destroyed");
  double* out = (double*)Caml_ba_data_val(image); /* Bigarray data lives outside the OCaml heap: stable while the lock is released */
  const int32_t i_spp = (int32_t)Long_val(spp), i_mb = (int32_t)Long_val(max_bounces), i_gpus = (int32_t)Long_val(gpus);
  exn = Val_unit;
  ptx_ml_cb cb = {&update_progress, &exn, 0};
  caml_release_runtime_system();
  const int32_t rc = ptx_ml_render(s, (int32_t)w, (int32_t)h, i_spp, i_mb, i_gpus, out, ptx_ml_progress, &cb);
  caml_acquire_runtime_system();
  ptx_ml_scene_release(handle);
  if (cb.raised) caml_raise(exn); /* update_progress raised: the render completed, its exception surfaces here */
  if (rc != 0) caml_failwith(ptx_last_error());
  CAMLreturn(Val_unit);
}

CAMLprim value ptx_ml_render_stub_bytecode(value* argv, int argn) {
  (void)argn;
  return ptx_ml_render_stub(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7]);
}

/* runs with the runtime lock held */
static void ptx_ml_call_on_iteration(ptx_ml_cb* cb, int32_t iteration, double radius, int64_t photon_map_length) {
  CAMLparam0();
  CAMLlocal2(boxed_radius, r);
  boxed_radius = caml_copy_double(radius);
  r = caml_callback3_exn(*cb->closure, Val_long(iteration), boxed_radius, Val_long(photon_map_length));
  if (Is_exception_result(r)) {
    *cb->exn = Extract_exception(r);
    cb->raised = 1;
  }
  CAMLreturn0;
}

/* the iteration callback of the library hands over ITS host copy of the running sum; the OCaml side reads its own Bigarray
 * (save_image reads img_sum.data, progressive_photon_map.ml:406-418), so the sum is copied there first */
typedef struct ptx_ml_ppm_cb {
  ptx_ml_cb cb;
  double* img_sum;
  size_t n;
} ptx_ml_ppm_cb;

static void ptx_ml_on_iteration_copy(void* user, int32_t iteration, double radius, int64_t photon_map_length, const double* img_sum) {
  ptx_ml_ppm_cb* p = (ptx_ml_ppm_cb*)user;
  if (p->cb.raised) return;
  memcpy(p->img_sum, img_sum, sizeof(double) * p->n);
  caml_acquire_runtime_system(); /* the render runs with the lock released */
  ptx_ml_call_on_iteration(&p->cb, iteration, radius, photon_map_length);
  caml_release_runtime_system();
}

/* external ppm_render_flat : scene -> floatarray -> floatarray -> img_sum -> (int -> float -> int -> unit) -> unit
 *   = "ptx_ml_ppm_render_stub"
 * replaces the loop of Progressive_photon_map.Make(Scene).go (progressive_photon_map.ml:433-450) */
CAMLprim value ptx_ml_ppm_render_stub(value handle, value params, value lights, value img_sum, value on_iteration) {
  CAMLparam5(handle, params, lights, img_sum, on_iteration);
  CAMLlocal1(exn);
  if (floatarray_length(params) != 6) caml_invalid_argument("Ptx.ppm_render: params needs 6 floats");
  const int32_t n_lights = floatarray_length(lights) / 11;
  if (floatarray_length(lights) != 11 * n_lights || n_lights < 1 || n_lights > 64)
    caml_invalid_argument("Ptx.ppm_render: 1 to 64 lights of 11 floats each");
  /* params / lights live in the OCaml heap and the lock is about to be released: copy them out */
  double p6[6], l11[64 * 11];
  memcpy(p6, floatarray_data(params), sizeof p6);
  memcpy(l11, floatarray_data(lights), sizeof(double) * 11 * (size_t)n_lights);
  const intnat w = (intnat)p6[0], h = (intnat)p6[1];
  if (w <= 0 || h <= 0 || Caml_ba_array_val(img_sum)->dim[0] != w * h * 3)
    caml_invalid_argument("Ptx.ppm_render: img_sum must hold width * height * 3 floats");
  int busy = 0;
  ptx_scene* s = ptx_ml_scene_acquire(handle, &busy);
  if (!s && busy) caml_failwith("Ptx.ppm_render: the scene's image is being pinned or unpinned on another thread");
  if (!s) caml_invalid_argument("Ptx.ppm_render: scene already destroyed");
  exn = Val_unit;
  ptx_ml_ppm_cb cb = {{&on_iteration, &exn, 0}, (double*)Caml_ba_data_val(img_sum), (size_t)(w * h * 3)};
  caml_release_runtime_system();
  const int32_t rc = ptx_ml_ppm_render(s, p6, l11, n_lights, cb.img_sum, ptx_ml_on_iteration_copy, &cb);
  caml_acquire_runtime_system();
  ptx_ml_scene_release(handle);
  if (cb.cb.raised) caml_raise(exn);
  if (rc != 0) caml_failwith(ptx_last_error());
  CAMLreturn(Val_unit);
}
