// render_command.cpp -- the reference's command line, kept flag for flag, in front of libptx_hip.so.
//
// Mirrors Render_command.Args.term (render_command/src/render_command.ml:16-47):
//   -d, --dimension=WIDTH,HEIGHT   (required)      --samples-per-pixel=INT (default 1)
//   -o, --output=PATH (default output.png)         --no-progress
//   --max-ray-bounces=INT (default 8)
// plus shirley_spheres' own --no-simd (shirley_spheres/bin/main.ml:12-23), and the prints of
// shirley_spheres/bin/main.ml:254-267 / render_command.ml:108.  Additions: --scene, --device, --gpus (SURVEY section 5
// "config / flags": the image spread over N GPUs of the node inside this process, ptx_render_params.n_gpus).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "host.h"

namespace {

struct Args {
  int width = 0, height = 0;
  int samples_per_pixel = 1;
  std::string output = "output.png";
  bool no_progress = false;
  int max_bounces = 8;
  bool no_simd = false;
  std::string scene = "shirley";
  int device = 0;
  int gpus = 1;
  int ganesha_triangles = 150000;
  double ceiling_emit = 12.0;
  std::string ganesha_ply; // -ganesha-ply <file> (ganesha/bin/main.ml:19-24); empty = synthetic stand-in mesh
};

[[noreturn]] void usage(const char* prog, const char* msg) {
  if (msg) std::fprintf(stderr, "%s: %s\n", prog, msg);
  std::fprintf(stderr,
               "Usage: %s -d WIDTH,HEIGHT [--samples-per-pixel=INT] [-o PATH] [--no-progress]\n"
               "          [--max-ray-bounces=INT] [--no-simd] [--scene=shirley|cornell|ganesha] [--device=INT] [--gpus=INT]\n"
               "          [--ganesha-ply=PATH] [--triangles=INT] [--ceiling-emit=FLOAT]\n",
               prog);
  std::exit(msg ? 124 : 0); // Cmdliner exits 124 on a CLI error
}

bool take_value(int argc, char** argv, int& i, const char* long_name, const char* short_name, std::string* out) {
  const std::string a = argv[i];
  const std::string ln = std::string("--") + long_name;
  if (a.rfind(ln + "=", 0) == 0) {
    *out = a.substr(ln.size() + 1);
    return true;
  }
  if (a == ln || (short_name && a == std::string("-") + short_name)) {
    if (i + 1 >= argc) usage(argv[0], ("option " + a + " needs an argument").c_str());
    *out = argv[++i];
    return true;
  }
  if (short_name && a.rfind(std::string("-") + short_name, 0) == 0 && a.size() > 2 && a[1] != '-') {
    *out = a.substr(2);
    return true;
  }
  return false;
}

Args parse(int argc, char** argv) {
  Args a;
  bool have_dim = false;
  for (int i = 1; i < argc; ++i) {
    std::string v;
    if (take_value(argc, argv, i, "dimension", "d", &v)) {
      if (std::sscanf(v.c_str(), "%d,%d", &a.width, &a.height) != 2) usage(argv[0], "invalid value for --dimension, expected WIDTH,HEIGHT");
      have_dim = true;
    } else if (take_value(argc, argv, i, "samples-per-pixel", nullptr, &v)) a.samples_per_pixel = std::atoi(v.c_str());
    else if (take_value(argc, argv, i, "output", "o", &v)) a.output = v;
    else if (take_value(argc, argv, i, "max-ray-bounces", nullptr, &v)) a.max_bounces = std::atoi(v.c_str());
    else if (take_value(argc, argv, i, "scene", nullptr, &v)) a.scene = v;
    else if (take_value(argc, argv, i, "device", nullptr, &v)) a.device = std::atoi(v.c_str());
    else if (take_value(argc, argv, i, "gpus", nullptr, &v)) a.gpus = std::atoi(v.c_str());
    else if (take_value(argc, argv, i, "triangles", nullptr, &v)) a.ganesha_triangles = std::atoi(v.c_str());
    else if (take_value(argc, argv, i, "ceiling-emit", nullptr, &v)) a.ceiling_emit = std::atof(v.c_str());
    else if (take_value(argc, argv, i, "ganesha-ply", nullptr, &v)) a.ganesha_ply = v;
    else if (!std::strcmp(argv[i], "-ganesha-ply") && i + 1 < argc) a.ganesha_ply = argv[++i]; // Stdlib.Arg spelling
    else if (!std::strcmp(argv[i], "--no-progress")) a.no_progress = true;
    else if (!std::strcmp(argv[i], "--no-simd")) a.no_simd = true;
    else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h")) usage(argv[0], nullptr);
    else usage(argv[0], (std::string("unknown option ") + argv[i]).c_str());
  }
  if (!have_dim) usage(argv[0], "required option --dimension is missing");
  // checked here, before anything is allocated from them (Cmdliner would also refuse a malformed WIDTH,HEIGHT)
  if (a.width <= 0 || a.height <= 0) usage(argv[0], "invalid value for --dimension, WIDTH and HEIGHT must be positive");
  if ((long long)a.width * a.height > (1ll << 31)) usage(argv[0], "invalid value for --dimension, image too large");
  if (a.samples_per_pixel < 1) usage(argv[0], "invalid value for --samples-per-pixel, must be >= 1");
  if (a.max_bounces < 0) usage(argv[0], "invalid value for --max-ray-bounces, must be >= 0");
  if (a.gpus < 1) usage(argv[0], "invalid value for --gpus, must be >= 1");
  return a;
}

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct Progress {
  long long total = 0, done = 0;
  double t0 = 0, last = 0;
};
void on_progress(void* user, int64_t pixels) { // the ASCII bar of render_command.ml:86-103
  Progress* p = (Progress*)user;
  p->done += pixels;
  const double t = now_ms();
  if (t - p->last < 200.0 && p->done < p->total) return; // min_interval 0.2 s
  p->last = t;
  const int width = 40, fill = (int)(width * (double)p->done / (double)p->total);
  std::fprintf(stderr, "\r%6.1fs [", (t - p->t0) * 1e-3);
  for (int i = 0; i < width; ++i) std::fputc(i < fill ? '#' : '-', stderr);
  std::fprintf(stderr, "] %3.0f%%    ", 100.0 * (double)p->done / (double)p->total);
  if (p->done >= p->total) std::fputc('\n', stderr);
}

}  // namespace

int main(int argc, char** argv) {
  const Args a = parse(argc, argv);
  pth_scene* hs = nullptr;
  if (a.scene == "shirley") hs = pth_scene_shirley(a.width, a.height, a.no_simd ? 1 : 0, 42); // Random.init 42
  else if (a.scene == "cornell") hs = pth_scene_cornell(a.width, a.height, a.ceiling_emit);
  else if (a.scene == "ganesha")
    hs = a.ganesha_ply.empty() ? pth_scene_ganesha_like(a.width, a.height, a.ganesha_triangles, 7) : pth_scene_ganesha_ply(a.ganesha_ply.c_str(), a.width, a.height);
  else usage(argv[0], "unknown --scene");
  if (!hs) {
    std::fprintf(stderr, "%s: cannot build scene %s: %s\n", argv[0], a.scene.c_str(), pth_last_error());
    return 1;
  }
  const ptx_scene_desc* d = pth_scene_desc(hs);
  std::printf("dim = %d x %d;\n", a.width, a.height);
  if (d->n_spheres) std::printf("#spheres = %d\n", d->n_spheres);
  if (d->n_triangles) std::printf("#triangles = %d\n", d->n_triangles);
  ptx_scene* scene = ptx_scene_create(d, a.device);
  if (!scene) {
    std::fprintf(stderr, "ptx_scene_create: %s\n", ptx_last_error());
    return 1;
  }
  ptx_stats st;
  ptx_scene_stats(scene, &st);
  std::printf("tree depth = %d\n", st.tree_depth);
  std::printf("build time = %.3f ms\n", st.build_ms);
  { // leaf lengths = ((size n) (count m)) ..., like Leaf_lengths (main.ml:233-248,265-267)
    std::vector<int32_t> info((size_t)st.tree_nodes * 4);
    ptx_scene_tree(scene, nullptr, info.data(), st.tree_nodes, nullptr, 0);
    std::map<int, int> hist;
    for (int i = 0; i < st.tree_nodes; ++i)
      if (info[(size_t)4 * i]) hist[info[(size_t)4 * i + 3]]++;
    std::printf("leaf lengths =\n(");
    bool first = true;
    for (auto& kv : hist) {
      std::printf("%s((size %d) (count %d))", first ? "" : " ", kv.first, kv.second);
      first = false;
    }
    std::printf(")\n");
    std::fflush(stdout);
  }
  ptx_render_params p;
  std::memset(&p, 0, sizeof p);
  p.width = a.width; p.height = a.height; p.samples_per_pixel = a.samples_per_pixel; p.max_bounces = a.max_bounces;
  p.n_gpus = a.gpus;
  std::vector<double> rgb((size_t)a.width * a.height * 3);
  Progress prog;
  prog.total = (long long)a.width * a.height;
  prog.t0 = prog.last = now_ms();
  const double t0 = now_ms();
  /* the image lives until the PNG is written, like the reference's Bimage (render_command.ml:64-70): pin it, so the frame comes
   * back with one DMA (optional: a failure only means the staged copy) */
  (void)ptx_image_pin(scene, rgb.data(), (int64_t)rgb.size());
  const int rc = ptx_render(scene, &p, rgb.data(), &st, a.no_progress ? nullptr : on_progress, &prog);
  (void)ptx_image_unpin(scene);
  const double elapsed = now_ms() - t0;
  if (rc != 0) {
    std::fprintf(stderr, "ptx_render: %s\n", ptx_last_error());
    return 1;
  }
  if (pth_write_png(a.output.c_str(), a.width, a.height, rgb.data()) != 0) {
    std::fprintf(stderr, "cannot write %s\n", a.output.c_str());
    return 1;
  }
  std::printf("rendered in: %.3f ms\n", elapsed);
  std::printf("throughput: %.3f Msamples/s\n", (double)st.samples / elapsed * 1e-3);
  ptx_scene_destroy(scene);
  pth_scene_free(hs);
  return 0;
}
