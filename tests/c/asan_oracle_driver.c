/* Sanitizer driver for the ORACLE (test infrastructure, CPU only): oracle/pt_oracle.c compiled with
 * -fsanitize=address,undefined into this executable and run over the three stock scenes -- scene generators, Shape_tree.create,
 * the tile-parallel path integrator, explicit-ray queries, the photon mapper.  Built by `make asan` in oracle/, run by
 * tests/test_sanitizers.py.  A checker with undefined behaviour in it proves nothing: this keeps it honest. */
#include "pt_oracle.c"

#include <stdio.h>

static int run(orc_desc* od, int w, int h, int spp, int depth, int ppm) {
  const ptx_scene_desc* d = orc_desc_get(od);
  orc_scene* sc = orc_scene_create(d);
  if (!sc) return 1;
  int info[5];
  double ms;
  orc_scene_info(sc, info, &ms);
  double* rgb = (double*)calloc((size_t)w * h * 3, sizeof(double));
  double* raw = (double*)calloc((size_t)w * h * 3, sizeof(double));
  int64_t ct[5];
  double rms = 0.0;
  if (orc_render(sc, w, h, spp, depth, 3, rgb, raw, ct, &rms) != 0) return 2;
  double o[6] = {0, 0, 0, 0.1, 0.2, 5.0}, dir[6] = {0.01, -0.02, -1.0, 0.0, 0.0, -1.0}, t[2];
  int32_t prim[2];
  int64_t ct2[5];
  orc_intersect_rays(sc, 2, o, dir, t, prim, ct2);
  if (ppm) {
    ptx_ppm_params p;
    memset(&p, 0, sizeof p);
    p.width = w; p.height = h; p.iterations = 2; p.max_bounces = 4; p.photon_count = 3000; p.alpha = 2.0 / 3.0;
    ptx_light lights[2];
    const int nl = ppm == 1 ? orc_lights_cornell(w, h, lights) : orc_lights_ganesha(sc, lights);
    int64_t st[4];
    double radius;
    memset(raw, 0, sizeof(double) * (size_t)w * h * 3);
    if (orc_ppm_render(sc, &p, lights, nl, raw, st, &radius) != 0) return 3;
  }
  printf("nodes %d depth %d segments %lld first %.6f\n", info[0], info[2], (long long)ct[1], rgb[0]);
  free(rgb);
  free(raw);
  orc_scene_destroy(sc);
  orc_desc_destroy(od);
  return 0;
}

int main(void) {
  int rc = 0;
  for (int math = 0; math < 2; ++math) {
    orc_set_math(math);
    rc |= run(orc_desc_shirley(48, 24, 0, 42), 48, 24, 2, 8, 0);
    rc |= run(orc_desc_shirley(48, 24, 1, 42), 48, 24, 2, 8, 0);
  }
  orc_set_math(0);
  rc |= run(orc_desc_cornell(32, 32, 12.0), 32, 32, 2, 16, 1);
  rc |= run(orc_desc_ganesha_like(48, 27, 2000, 7), 48, 27, 2, 8, 2);
  return rc;
}
