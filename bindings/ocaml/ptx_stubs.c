/* ptx_stubs.c -- OCaml C stubs for libptx_hip.so (include/ptx.h), bound by ptx.ml's `external`s.
 *
 * Same kind of binding as the reference's only FFI, `external spheres_intersect_native ... [@@noalloc]` and
 * `external leaf_size` (shirley_spheres/bin/main.ml:162-172, implemented in sphere-intersect-rs/src/lib.rs:15-18,53-76
 * and linked by shirley_spheres/bin/dune:8-33): plain `external` C symbols, OCaml values unpacked by hand, pointers
 * BORROWED for the duration of the call -- but one call per render instead of one per BVH leaf per ray.
 *
 * Needs the OCaml runtime headers (<caml/...>), so it is not compiled in this repository's image; the marshalling it
 * performs lives in ptx_ml_marshal.h and is exercised without OCaml by tests/test_ocaml_binding.py.
 *
 * Value conventions:
 *   floatarray          Double_array_tag block: (double*) v is the flat data, Wosize_val(v) / Double_wosize the length
 *                       (the reference reads `coords` the same way: header >> 10 words, lib.rs:26-36)
 *   Bigarray.Array1     Caml_ba_data_val(v), Caml_ba_array_val(v)->dim[0]
 *   int                 Long_val / Val_long
 *   scene handle        custom block holding the ptx_scene*, finalised with ptx_scene_destroy
 *   errors              caml_failwith (ptx_last_error ()): no error code ever reaches OCaml unraised
 *   progress closure    called through caml_callback on the CALLING thread (ptx_render's contract), so the runtime lock
 *                       is held and no foreign thread ever touches the OCaml heap
 */
#define CAML_NAME_SPACE
#include <caml/alloc.h>
#include <caml/bigarray.h>
#include <caml/callback.h>
#include <caml/custom.h>
#include <caml/fail.h>
#include <caml/memory.h>
#include <caml/mlvalues.h>

#include "ptx_ml_marshal.h"

#define Scene_val(v) (*((ptx_scene**)Data_custom_val(v)))

static void ptx_ml_scene_finalize(value v) {
  if (Scene_val(v)) {
    ptx_scene_destroy(Scene_val(v));
    Scene_val(v) = NULL;
  }
}

static struct custom_operations ptx_ml_scene_ops = {
    "dalev.path_tracer.ptx_scene", ptx_ml_scene_finalize, custom_compare_default, custom_hash_default,
    custom_serialize_default,       custom_deserialize_default, custom_compare_ext_default, custom_fixed_length_default};

static const double* floatarray_data(value v) { return (const double*)v; }
static int32_t floatarray_length(value v) { return (int32_t)(Wosize_val(v) / Double_wosize); }

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

/* external scene_create : flat -> int -> scene = "ptx_ml_scene_create_stub"
 * flat = { xs; ys; zs; rs; sphere_material; materials; textures; camera; background; leaf_kind; length_cutoff }
 * (field order of Ptx.flat in ptx.ml) */
CAMLprim value ptx_ml_scene_create_stub(value flat, value device) {
  CAMLparam2(flat, device);
  CAMLlocal1(handle);
  ptx_ml_flat f;
  f.xs = floatarray_data(Field(flat, 0));
  f.ys = floatarray_data(Field(flat, 1));
  f.zs = floatarray_data(Field(flat, 2));
  f.rs = floatarray_data(Field(flat, 3));
  f.n_spheres = floatarray_length(Field(flat, 0));
  if (floatarray_length(Field(flat, 1)) != f.n_spheres || floatarray_length(Field(flat, 2)) != f.n_spheres ||
      floatarray_length(Field(flat, 3)) != f.n_spheres || Caml_ba_array_val(Field(flat, 4))->dim[0] != f.n_spheres)
    caml_invalid_argument("Ptx.scene_create: sphere arrays differ in length");
  f.sphere_material = (const int32_t*)Caml_ba_data_val(Field(flat, 4));
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
  /* no OCaml allocation between reading the pointers above and the end of ptx_ml_scene_create: nothing can move */
  ptx_scene* s = ptx_ml_scene_create(&f, (int32_t)Long_val(device));
  if (!s) caml_failwith(ptx_last_error());
  handle = caml_alloc_custom(&ptx_ml_scene_ops, sizeof(ptx_scene*), 0, 1);
  Scene_val(handle) = s;
  CAMLreturn(handle);
}

/* external scene_destroy : scene -> unit = "ptx_ml_scene_destroy_stub"   (optional: the finaliser does the same) */
CAMLprim value ptx_ml_scene_destroy_stub(value handle) {
  ptx_ml_scene_finalize(handle);
  return Val_unit;
}

static void ptx_ml_progress(void* user, int64_t pixels_done) {
  /* `user` points at a GC root registered by the caller below; the closure may allocate */
  caml_callback(*(value*)user, Val_long(pixels_done));
}

/* external render : scene -> int -> int -> int -> int -> int -> image -> (int -> unit) -> unit
 *   = "ptx_ml_render_bytecode" "ptx_ml_render"
 * (scene, width, height, samples_per_pixel, max_bounces, gpus, Bimage data as a float64 Bigarray of W*H*3, update_progress)
 * replaces Integrator.create ... |> Integrator.render ~update_progress (render_command.ml:71-104) */
CAMLprim value ptx_ml_render(value handle, value width, value height, value spp, value max_bounces, value gpus, value image,
                             value update_progress) {
  CAMLparam5(handle, width, height, spp, max_bounces);
  CAMLxparam3(gpus, image, update_progress);
  ptx_scene* s = Scene_val(handle);
  if (!s) caml_invalid_argument("Ptx.render: scene already destroyed");
  const intnat w = Long_val(width), h = Long_val(height);
  if (Caml_ba_array_val(image)->dim[0] != w * h * 3) caml_invalid_argument("Ptx.render: image must hold width * height * 3 floats");
  double* out = (double*)Caml_ba_data_val(image); /* Bigarray data lives outside the OCaml heap: stable across callbacks */
  const int32_t rc = ptx_ml_render(s, (int32_t)w, (int32_t)h, (int32_t)Long_val(spp), (int32_t)Long_val(max_bounces),
                                   (int32_t)Long_val(gpus), out, ptx_ml_progress, &update_progress);
  if (rc != 0) caml_failwith(ptx_last_error());
  CAMLreturn(Val_unit);
}

CAMLprim value ptx_ml_render_bytecode(value* argv, int argn) {
  (void)argn;
  return ptx_ml_render(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7]);
}
