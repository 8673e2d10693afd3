/* Drives ptx_scene_create / ptx_render through bindings/ocaml/ptx_ml_marshal.h with flat arrays laid out exactly as
 * bindings/ocaml/ptx.ml's `flatten` produces them (the OCaml stub only turns OCaml values into these pointers).
 * usage: driver <flat.bin> tree                       -> host-only scene (device -1), prints tree statistics
 *        driver <flat.bin> render W H SPP BOUNCES GPUS out.bin   -> ptx_ml_render, framebuffer to out.bin */
#include <stdio.h>
#include <stdlib.h>

#include "ptx_ml_marshal.h"

static long long g_pixels;
static void on_progress(void* user, int64_t n) {
  (void)user;
  g_pixels += n;
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
  int32_t hdr[5]; /* n_spheres, n_materials, n_textures, leaf_kind, length_cutoff */
  if (fread(hdr, sizeof hdr, 1, f) != 1) return 2;
  ptx_ml_flat fl;
  fl.n_spheres = hdr[0]; fl.n_materials = hdr[1]; fl.n_textures = hdr[2]; fl.leaf_kind = hdr[3]; fl.length_cutoff = hdr[4];
  fl.xs = slurp(f, sizeof(double) * (size_t)hdr[0]);
  fl.ys = slurp(f, sizeof(double) * (size_t)hdr[0]);
  fl.zs = slurp(f, sizeof(double) * (size_t)hdr[0]);
  fl.rs = slurp(f, sizeof(double) * (size_t)hdr[0]);
  fl.sphere_material = slurp(f, sizeof(int32_t) * (size_t)hdr[0]);
  fl.materials = slurp(f, sizeof(double) * 6 * (size_t)hdr[1]);
  fl.textures = slurp(f, sizeof(double) * 9 * (size_t)hdr[2]);
  fl.camera = slurp(f, sizeof(double) * 4);
  fl.background = slurp(f, sizeof(double) * 7);
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
  if (!tree_only) {
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
  }
  ptx_scene_destroy(s);
  return 0;
}
