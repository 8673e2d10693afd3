/* Drives ptx_scene_create / ptx_render / ptx_ppm_render through bindings/ocaml/ptx_ml_marshal.h with flat arrays laid out
 * exactly as bindings/ocaml/ptx.ml's `flatten` produces them (the OCaml stub only turns OCaml values into these pointers).
 * usage: driver <flat.bin> tree                                   -> host-only scene (device -1), prints tree statistics
 *        driver <flat.bin> render W H SPP BOUNCES GPUS out.bin    -> ptx_ml_render, framebuffer to out.bin
 *        driver <flat.bin> ppm <ppm.bin> out.bin                  -> ptx_ml_ppm_render; ppm.bin = 6 params, n_lights, 11 per light
 * flat.bin: int32 header {n_spheres, n_materials, n_textures, leaf_kind, length_cutoff, n_vertices, n_triangles, n_floor}, then
 * xs ys zs rs | sphere_material | materials | textures | camera | background | vertex_x y z | tri_indices | tri_uv |
 * tri_material | floor_vertices | floor_uv | floor_material */
#include <stdio.h>
#include <stdlib.h>

#include "ptx_ml_marshal.h"

static long long g_pixels;
static void on_progress(void* user, int64_t n) {
  (void)user;
  g_pixels += n;
}

static int g_iterations;
static double g_last_running_sum;
static void on_iteration(void* user, int32_t iteration, double radius, int64_t length, const double* img_sum) {
  (void)user;
  (void)radius;
  if (iteration != g_iterations || length <= 0) g_iterations = -1000; /* iterations arrive in order, 0, 1, ... */
  ++g_iterations;
  g_last_running_sum = img_sum[0];
}

static void* slurp(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (bytes && fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t hdr[8];
  if (fread(hdr, sizeof hdr, 1, f) != 1) return 2;
  ptx_ml_flat fl;
  memset(&fl, 0, sizeof fl);
  fl.n_spheres = hdr[0]; fl.n_materials = hdr[1]; fl.n_textures = hdr[2]; fl.leaf_kind = hdr[3]; fl.length_cutoff = hdr[4];
  fl.n_vertices = hdr[5]; fl.n_triangles = hdr[6]; fl.n_floor_triangles = hdr[7];
  const size_t ns = (size_t)hdr[0], nv = (size_t)hdr[5], nt = (size_t)hdr[6], nf = (size_t)hdr[7];
  fl.xs = slurp(f, sizeof(double) * ns);
  fl.ys = slurp(f, sizeof(double) * ns);
  fl.zs = slurp(f, sizeof(double) * ns);
  fl.rs = slurp(f, sizeof(double) * ns);
  fl.sphere_material = slurp(f, sizeof(int32_t) * ns);
  fl.materials = slurp(f, sizeof(double) * 6 * (size_t)hdr[1]);
  fl.textures = slurp(f, sizeof(double) * 9 * (size_t)hdr[2]);
  fl.camera = slurp(f, sizeof(double) * 4);
  fl.background = slurp(f, sizeof(double) * 7);
  fl.vertex_x = slurp(f, sizeof(double) * nv);
  fl.vertex_y = slurp(f, sizeof(double) * nv);
  fl.vertex_z = slurp(f, sizeof(double) * nv);
  fl.tri_indices = slurp(f, sizeof(int32_t) * 3 * nt);
  fl.tri_uv = slurp(f, sizeof(double) * 6 * nt);
  fl.tri_material = slurp(f, sizeof(int32_t) * nt);
  fl.floor_vertices = slurp(f, sizeof(double) * 9 * nf);
  fl.floor_uv = slurp(f, sizeof(double) * 6 * nf);
  fl.floor_material = slurp(f, sizeof(int32_t) * nf);
  fclose(f);
  const int tree_only = argv[2][0] == 't';
  ptx_scene* s = ptx_ml_scene_create(&fl, tree_only ? -1 : 0);
  if (!s) {
    fprintf(stderr, "scene_create: %s\n", ptx_last_error());
    return 1;
  }
  ptx_stats st;
  ptx_scene_stats(s, &st);
  printf("leaf_size %d nodes %d depth %d leaves %d slots %d\n", ptx_leaf_size(), st.tree_nodes, st.tree_depth, st.tree_leaves, st.leaf_slots);
  if (argv[2][0] == 'r') {
    if (argc < 9) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    double* img = malloc(sizeof(double) * (size_t)w * h * 3);
    const int32_t rc = ptx_ml_render(s, w, h, atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), img, on_progress, NULL);
    if (rc != 0) {
      fprintf(stderr, "render: %s\n", ptx_last_error());
      return 1;
    }
    printf("progress_pixels %lld\n", g_pixels);
    FILE* o = fopen(argv[8], "wb");
    fwrite(img, sizeof(double), (size_t)w * h * 3, o);
    fclose(o);
    /* the error path the stub turns into caml_failwith: a negative code and a message, never a crash */
    if (ptx_ml_render(s, w, h, 0, 8, 1, img, NULL, NULL) == 0 || !ptx_last_error()[0]) return 3;
  } else if (argv[2][0] == 'p') {
    if (argc < 5) return 2;
    FILE* pf = fopen(argv[3], "rb");
    if (!pf) return 2;
    double* p6 = slurp(pf, sizeof(double) * 6);
    int32_t n_lights;
    if (fread(&n_lights, sizeof n_lights, 1, pf) != 1) return 2;
    double* l11 = slurp(pf, sizeof(double) * 11 * (size_t)n_lights);
    fclose(pf);
    const size_t n = (size_t)p6[0] * (size_t)p6[1] * 3;
    double* img = malloc(sizeof(double) * n);
    const int32_t rc = ptx_ml_ppm_render(s, p6, l11, n_lights, img, on_iteration, NULL);
    if (rc != 0) {
      fprintf(stderr, "ppm_render: %s\n", ptx_last_error());
      return 1;
    }
    printf("iterations %d\n", g_iterations);
    if (g_last_running_sum != img[0]) return 4; /* the last callback saw the final sum */
    FILE* o = fopen(argv[4], "wb");
    fwrite(img, sizeof(double), n, o);
    fclose(o);
    p6[2] = 0.0; /* iterations = 0: an error, not a crash */
    if (ptx_ml_ppm_render(s, p6, l11, n_lights, img, NULL, NULL) == 0 || !ptx_last_error()[0]) return 3;
  }
  ptx_scene_destroy(s);
  return 0;
}
