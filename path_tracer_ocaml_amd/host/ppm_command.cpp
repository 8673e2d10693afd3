// ppm_command.cpp -- the reference's photon-mapping executables (cornell-box/bin/main.ml, ganesha/bin/main.ml)
// in front of libptx_hip.so, with the Stdlib.Arg command line of Progressive_photon_map.Args.parse
// (progressive-photon-map/src/progressive_photon_map.ml:17-54):
//   -width <int> (600)  -height <int> (= width default)  -iterations <int> (10)  -photon-count <int> (75000)
//   -alpha <float> (2/3)  -o <file> (output.png)  -no-progress  -max-bounces <int> (4)
// ganesha adds -ganesha-ply <file> and -stop-after-bvh (ganesha/bin/main.ml:16-27).
// Built twice: -DPPM_SCENE_CORNELL -> cornell_box, -DPPM_SCENE_GANESHA -> ganesha.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host.h"

#ifndef PPM_NO_GAMMA_DEF
extern "C" void pth_ppm_gamma(const double* img_sum, int64_t count, int32_t n, double* out) {
  const double one_over_n = 1.0 / (double)n; // 1 // n
  for (int64_t i = 0; i < count; ++i) out[i] = std::pow(img_sum[i] * one_over_n, 1.0 / 2.2); // gamma x = x ** (1 / 2.2)
}
#endif

#if defined(PPM_SCENE_CORNELL) || defined(PPM_SCENE_GANESHA)
namespace {
struct Ctx {
  std::string output;
  int width, height;
  std::vector<double> avg;
};
double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
void on_iteration(void* user, int32_t i, double radius, int64_t length, const double* img_sum) {
  Ctx* c = (Ctx*)user;
  std::printf("#iteration = %d, radius = %.3f\n", i, radius);
  std::printf("  photon map length = %lld\n", (long long)length);
  std::fflush(stdout);
  pth_ppm_gamma(img_sum, (int64_t)c->width * c->height * 3, i + 1, c->avg.data()); // save_image after every iteration
  pth_write_png(c->output.c_str(), c->width, c->height, c->avg.data());
}
}  // namespace

int main(int argc, char** argv) {
  ptx_ppm_params p;
  std::memset(&p, 0, sizeof p);
  p.width = 600; p.height = -1; p.iterations = 10; p.photon_count = 75000; p.alpha = 2.0 / 3.0; p.max_bounces = 4;
  std::string output = "output.png", ply;
  bool stop_after_bvh = false;
  int n_tri = 150000;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s: option '%s' needs an argument.\n", argv[0], a.c_str()); std::exit(2); }
      return argv[++i];
    };
    if (a == "-width") p.width = std::atoi(val());
    else if (a == "-height") p.height = std::atoi(val());
    else if (a == "-iterations") p.iterations = std::atoi(val());
    else if (a == "-photon-count") p.photon_count = std::atoi(val());
    else if (a == "-alpha") p.alpha = std::atof(val());
    else if (a == "-o") output = val();
    else if (a == "-no-progress") {}
    else if (a == "-max-bounces") p.max_bounces = std::atoi(val());
#ifdef PPM_SCENE_GANESHA
    else if (a == "-ganesha-ply") ply = val();
    else if (a == "-stop-after-bvh") stop_after_bvh = true;
    else if (a == "-triangles") n_tri = std::atoi(val());
#endif
    else if (a == "-help" || a == "--help") {
      std::printf("Defaults: width = 600, height = 600, output = output.png\n");
      return 0;
    } else { std::fprintf(stderr, "%s: unknown option '%s'.\n", argv[0], a.c_str()); return 2; } // Stdlib.Arg exits 2
  }
  if (p.height < 0) p.height = p.width; // `let height = ref !width`
  (void)stop_after_bvh; (void)n_tri;
  pth_scene* hs;
  ptx_light lights[2];
  int n_lights;
#ifdef PPM_SCENE_CORNELL
  hs = pth_scene_cornell(p.width, p.height, 0.0); // the reference's scene has no emitter: a point light instead
  n_lights = pth_lights_cornell(p.width, p.height, lights);
#else
  hs = ply.empty() ? pth_scene_ganesha_like(p.width, p.height, n_tri, 7) : pth_scene_ganesha_ply(ply.c_str(), p.width, p.height);
  if (!hs) { std::fprintf(stderr, "%s\n", pth_last_error()); return 1; }
  n_lights = pth_lights_ganesha(hs, lights);
#endif
  ptx_scene_desc d = *pth_scene_desc(hs);
  d.background.kind = PTX_BG_BLACK; // the photon mapper has no background (progressive_photon_map.ml:326)
  const double t_build = now_ms();
  ptx_scene* scene = ptx_scene_create(&d, 0);
  if (!scene) { std::fprintf(stderr, "ptx_scene_create: %s\n", ptx_last_error()); return 1; }
#ifdef PPM_SCENE_GANESHA
  ptx_stats bs;
  ptx_scene_stats(scene, &bs);
  std::printf("dim = %d x %d;\n#triangles = %d\n", p.width, p.height, d.n_triangles);
  std::printf("tree depth = %d\nbuild time = %.3f ms\n", bs.tree_depth, now_ms() - t_build);
  if (stop_after_bvh) { std::printf("Stop after bvh build\n"); return 0; }
#else
  (void)t_build;
#endif
  std::printf("#max-bounces = %d\n#photons/iter = %d\n#iterations = %d\n-----\n", p.max_bounces, p.photon_count, p.iterations);
  std::fflush(stdout);
  Ctx ctx{output, p.width, p.height, std::vector<double>((size_t)p.width * p.height * 3)};
  std::vector<double> img((size_t)p.width * p.height * 3);
  ptx_ppm_stats st;
  const double t0 = now_ms();
  if (ptx_ppm_render(scene, &p, lights, n_lights, img.data(), &st, on_iteration, &ctx) != 0) {
    std::fprintf(stderr, "ptx_ppm_render: %s\n", ptx_last_error());
    return 1;
  }
#ifdef PPM_SCENE_CORNELL
  std::printf("render time = %.3f ms\n", now_ms() - t0);
#else
  std::printf("elapsed ms: %.3f\n", now_ms() - t0);
#endif
  ptx_scene_destroy(scene);
  pth_scene_free(hs);
  return 0;
}
#endif
