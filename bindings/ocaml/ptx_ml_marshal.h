/* ptx_ml_marshal.h -- the argument marshalling of the OCaml stub, free of OCaml headers.
 *
 * ptx_stubs.c (which needs <caml/...> and cannot be compiled in this image) only unpacks OCaml values into the
 * FLAT view below -- borrowed pointers into floatarrays / Bigarrays, exactly the convention of the reference's
 * existing stub (value coords -> f64 slices read in place, sphere-intersect-rs/src/lib.rs:53-76) -- and calls
 * these two functions.  Everything that can go wrong between "flat OCaml data" and the C ABI of include/ptx.h
 * therefore lives here, where tests/test_ocaml_binding.py can compile and run it (gcc, no OCaml needed).
 *
 * Flat layout (what bindings/ocaml/ptx.ml produces):
 *   xs, ys, zs, rs     floatarray, one entry per sphere, CAMERA space (Sphere.transform ~f:(Camera.transform camera),
 *                      shirley_spheres/bin/main.ml:258-260)
 *   sphere_material    int32 Bigarray, index into `materials`
 *   materials          floatarray, 6 per material : kind (0 Lambertian, 1 Metal, 2 Dielectric), texture index,
 *                      refraction index, emit r g b          (Material.t, path_tracer/src/material.ml:3-14)
 *   textures           floatarray, 9 per texture  : kind (0 solid, 1 checker), width, height, even r g b, odd r g b
 *                      (Texture.solid / Texture.checker, path_tracer/src/texture.ml:16-31)
 *   camera             floatarray, 4 : lower_left_x, lower_left_y, view_x, view_y   (camera.ml:50-53)
 *   background         floatarray, 7 : kind (0 black, 1 sky), horizon r g b, zenith r g b  (main.ml:104-110)
 */
#ifndef PTX_ML_MARSHAL_H
#define PTX_ML_MARSHAL_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ptx.h"

typedef struct ptx_ml_flat {
  int32_t n_spheres;
  const double *xs, *ys, *zs, *rs;
  const int32_t* sphere_material;
  int32_t n_materials;
  const double* materials; /* 6 per material */
  int32_t n_textures;
  const double* textures;  /* 9 per texture */
  const double* camera;     /* 4 */
  const double* background; /* 7 */
  int32_t leaf_kind;        /* PTX_LEAF_SIMD (Simd_leaf) / PTX_LEAF_ARRAY (Array_leaf, --no-simd) */
  int32_t length_cutoff;    /* Leaf.length_cutoff: leaf_size () = 16 / 4 (main.ml:121,175) */
} ptx_ml_flat;

/* NULL + ptx_last_error() on failure, like ptx_scene_create.  Nothing of `f` is referenced after the call returns. */
static ptx_scene* ptx_ml_scene_create(const ptx_ml_flat* f, int32_t device) {
  if (!f || f->n_spheres < 0 || f->n_materials <= 0 || f->n_textures < 0) return NULL;
  ptx_material* mats = (ptx_material*)calloc((size_t)f->n_materials, sizeof *mats);
  ptx_texture* texs = (ptx_texture*)calloc((size_t)(f->n_textures > 0 ? f->n_textures : 1), sizeof *texs);
  if (!mats || !texs) {
    free(mats);
    free(texs);
    return NULL;
  }
  for (int32_t i = 0; i < f->n_materials; ++i) {
    const double* m = f->materials + 6 * (size_t)i;
    mats[i].kind = (int32_t)m[0];
    mats[i].texture = (int32_t)m[1];
    mats[i].index = m[2];
    memcpy(mats[i].emit, m + 3, sizeof mats[i].emit);
  }
  for (int32_t i = 0; i < f->n_textures; ++i) {
    const double* t = f->textures + 9 * (size_t)i;
    texs[i].kind = (int32_t)t[0];
    texs[i].width = (int32_t)t[1];
    texs[i].height = (int32_t)t[2];
    memcpy(texs[i].even, t + 3, sizeof texs[i].even);
    memcpy(texs[i].odd, t + 6, sizeof texs[i].odd);
  }
  ptx_scene_desc d;
  memset(&d, 0, sizeof d);
  d.n_spheres = f->n_spheres;
  d.sphere_x = f->xs; d.sphere_y = f->ys; d.sphere_z = f->zs; d.sphere_r = f->rs;
  d.sphere_material = f->sphere_material;
  d.n_materials = f->n_materials;
  d.materials = mats;
  d.n_textures = f->n_textures;
  d.textures = texs;
  d.camera.lower_left_x = f->camera[0]; d.camera.lower_left_y = f->camera[1];
  d.camera.view_x = f->camera[2]; d.camera.view_y = f->camera[3];
  d.background.kind = (int32_t)f->background[0];
  memcpy(d.background.horizon, f->background + 1, sizeof d.background.horizon);
  memcpy(d.background.zenith, f->background + 4, sizeof d.background.zenith);
  d.leaf_kind = f->leaf_kind;
  d.length_cutoff = f->length_cutoff;
  d.num_bins = 0; /* Shape_tree.create's default, 32 */
  ptx_scene* s = ptx_scene_create(&d, device); /* copies everything it keeps */
  free(mats);
  free(texs);
  return s;
}

/* Integrator.render into the Bimage's f64 RGB buffer (W*H*3, row 0 = top).  0 or a negative ptx error code. */
static int32_t ptx_ml_render(ptx_scene* s, int32_t width, int32_t height, int32_t samples_per_pixel, int32_t max_bounces,
                             int32_t n_gpus, double* image, ptx_progress_fn progress, void* user) {
  ptx_render_params p;
  memset(&p, 0, sizeof p);
  p.width = width;
  p.height = height;
  p.samples_per_pixel = samples_per_pixel;
  p.max_bounces = max_bounces;
  p.n_gpus = n_gpus;
  return ptx_render(s, &p, image, NULL, progress, user);
}

#endif /* PTX_ML_MARSHAL_H */
