/* ptx_ml_marshal.h -- the argument marshalling of the OCaml stub, free of OCaml headers.
 *
 * ptx_stubs.c (which needs <caml/...> and cannot be compiled in this image) only unpacks OCaml values into the
 * FLAT view below -- borrowed pointers into floatarrays / Bigarrays, exactly the convention of the reference's
 * existing stub (value coords -> f64 slices read in place, sphere-intersect-rs/src/lib.rs:53-76) -- and calls
 * these functions.  Everything that can go wrong between "flat OCaml data" and the C ABI of include/ptx.h
 * therefore lives here, where tests/test_ocaml_binding.py can compile and run it (gcc, no OCaml needed).
 *
 * Flat layout (what bindings/ocaml/ptx.ml's `flatten` produces; every coordinate in CAMERA space):
 *   xs, ys, zs, rs     floatarray, one entry per sphere (Sphere.transform ~f:(Camera.transform camera),
 *                      shirley_spheres/bin/main.ml:258-260; cornell-box/bin/main.ml:70-91,218)
 *   sphere_material    int32 Bigarray, index into `materials`
 *   materials          floatarray, 6 per material : kind (0 Lambertian, 1 Metal, 2 Dielectric), texture index,
 *                      refraction index, emit r g b          (Material.t, path_tracer/src/material.ml:3-14; Hit.emit, hit.ml:5)
 *   textures           floatarray, 9 per texture  : kind (0 solid, 1 checker), width, height, even r g b, odd r g b
 *                      (Texture.solid / Texture.checker, path_tracer/src/texture.ml:16-31)
 *   camera             floatarray, 4 : lower_left_x, lower_left_y, view_x, view_y   (camera.ml:50-53)
 *   background         floatarray, 7 : kind (0 black, 1 sky), horizon r g b, zenith r g b  (main.ml:104-110)
 *   vertex_x/y/z       floatarray, one entry per mesh vertex (ganesha Mesh.t, ganesha/bin/main.ml:37-85; cornell-box's
 *                      Face.t keeps three private vertices per triangle, cornell-box/bin/main.ml:5-28)
 *   tri_indices        int32 Bigarray, 3 per triangle: a, b, c (ganesha Face.t, main.ml:92-110)
 *   tri_uv             floatarray, 6 per triangle: Face.tex_coords (ua, va), (ub, vb), (uc, vc)
 *   tri_material       int32 Bigarray, 1 per triangle, index into `materials`
 *   floor_vertices     floatarray, 9 per floor triangle: a, b, c -- tested BEFORE the tree, first hit clips t_max
 *                      (ganesha Floor.intersect, main.ml:205-260,286-298)
 *   floor_uv, floor_material   6 per floor triangle / int32 per floor triangle
 *   lights             floatarray, 11 per light: kind (0 point, 1 spot), position xyz, direction xyz, color rgb, power
 *                      (Progressive_photon_map.Light.create_point / create_spot, progressive_photon_map.ml:59-137)
 *   ppm params         floatarray, 6: width, height, iterations, max_bounces, photon_count, alpha (Args.t, :7-16)
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
  int32_t leaf_kind;        /* PTX_LEAF_SIMD (Simd_leaf) / PTX_LEAF_ARRAY (Array_leaf) */
  int32_t length_cutoff;    /* Leaf.length_cutoff: leaf_size () = 16 / 4 (main.ml:121,175) / 2 (cornell) / 8 (ganesha) */
  /* triangle mesh: all zero / NULL when the scene has none */
  int32_t n_vertices;
  const double *vertex_x, *vertex_y, *vertex_z;
  int32_t n_triangles;
  const int32_t* tri_indices; /* 3 per triangle */
  const double* tri_uv;       /* 6 per triangle */
  const int32_t* tri_material;
  int32_t n_floor_triangles;
  const double* floor_vertices; /* 9 per triangle */
  const double* floor_uv;       /* 6 per triangle */
  const int32_t* floor_material;
} ptx_ml_flat;

/* Why the last ptx_ml_scene_create of this thread returned NULL: a message of THIS layer (bad counts, no memory), or NULL when
 * the library refused the scene and ptx_last_error() holds the reason.  The stub raises ptx_ml_scene_create_error() first. */
static __thread const char* ptx_ml_create_error_;
static __attribute__((unused)) const char* ptx_ml_scene_create_error(void) { return ptx_ml_create_error_ ? ptx_ml_create_error_ : ptx_last_error(); }

/* NULL + ptx_ml_scene_create_error() on failure.  Nothing of `f` is referenced after the call returns. */
static ptx_scene* ptx_ml_scene_create(const ptx_ml_flat* f, int32_t device) {
  ptx_ml_create_error_ = NULL;
  if (!f || f->n_spheres < 0 || f->n_triangles < 0 || f->n_floor_triangles < 0 || f->n_textures < 0) {
    ptx_ml_create_error_ = "Ptx.scene_create: negative element count";
    return NULL;
  }
  if (f->n_materials <= 0) {
    ptx_ml_create_error_ = "Ptx.scene_create: the material table is empty";
    return NULL;
  }
  ptx_material* mats = (ptx_material*)calloc((size_t)f->n_materials, sizeof *mats);
  ptx_texture* texs = (ptx_texture*)calloc((size_t)(f->n_textures > 0 ? f->n_textures : 1), sizeof *texs);
  if (!mats || !texs) {
    free(mats);
    free(texs);
    ptx_ml_create_error_ = "Ptx.scene_create: out of memory for the material / texture tables";
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
  d.n_vertices = f->n_vertices;
  d.vertex_x = f->vertex_x; d.vertex_y = f->vertex_y; d.vertex_z = f->vertex_z;
  d.n_triangles = f->n_triangles;
  d.tri_indices = f->tri_indices;
  d.tri_uv = f->tri_uv;
  d.tri_material = f->tri_material;
  d.n_floor_triangles = f->n_floor_triangles;
  d.floor_vertices = f->floor_vertices;
  d.floor_uv = f->floor_uv;
  d.floor_material = f->floor_material;
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

/* Progressive_photon_map.Make(Scene).go's loop (progressive_photon_map.ml:420-451) without the prints and the PNG:
 * params6 = width, height, iterations, max_bounces, photon_count, alpha; lights11 = 11 doubles per light (above).
 * img_sum (W*H*3) receives the final sum; iteration_cb gets the running sum after every iteration. */
static int32_t ptx_ml_ppm_render(ptx_scene* s, const double* params6, const double* lights11, int32_t n_lights, double* img_sum,
                                 ptx_ppm_iteration_fn iteration_cb, void* user) {
  if (!params6 || n_lights < 0 || n_lights > 64 || (n_lights > 0 && !lights11)) return -1;
  ptx_ppm_params p;
  memset(&p, 0, sizeof p);
  p.width = (int32_t)params6[0];
  p.height = (int32_t)params6[1];
  p.iterations = (int32_t)params6[2];
  p.max_bounces = (int32_t)params6[3];
  p.photon_count = (int32_t)params6[4];
  p.alpha = params6[5];
  ptx_light lights[64];
  memset(lights, 0, sizeof lights);
  for (int32_t i = 0; i < n_lights; ++i) {
    const double* l = lights11 + 11 * (size_t)i;
    lights[i].kind = (int32_t)l[0];
    memcpy(lights[i].position, l + 1, sizeof lights[i].position);
    memcpy(lights[i].direction, l + 4, sizeof lights[i].direction);
    memcpy(lights[i].color, l + 7, sizeof lights[i].color);
    lights[i].power = l[10];
  }
  return ptx_ppm_render(s, &p, lights, n_lights, img_sum, NULL, iteration_cb, user);
}

#endif /* PTX_ML_MARSHAL_H */
