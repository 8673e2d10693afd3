// scenes.cpp -- host-side mirror of the reference's scene executables, in C++ (the reference's host
// language, OCaml, has no toolchain in this image).  Builds the declarative ptx_scene_desc that
// libptx_hip.so consumes, for the three scenes BASELINE.json names:
//
//   pth_scene_shirley      <- shirley_spheres/bin/main.ml:26-102,250-260
//   pth_scene_cornell      <- cornell-box/bin/main.ml:43-91,172-218 (+ documented ceiling emitter)
//   pth_scene_ganesha_like <- ganesha/bin/main.ml:30-35,50-119,205-260 over a synthetic mesh
//
// plus Camera.create / Camera.transform (path_tracer/src/camera.ml:14-27,39-43,58-83).
// Everything here is double precision with the reference's operation order (-ffp-contract=off).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "../csrc/pt_vec.h"
#include "host.h"

namespace {

constexpr double kPi = 3.14159265358979323846; // Float.pi

// ---------------------------------------------------------------- Camera (camera.ml)
struct Camera {
  double m[4][4]; // Mat4.look_at rows
  ptx_camera view;
};

// Mat4.dot4 (camera.ml:9-12): plain left-associated sum of products
double dot4(const double a[4], const double b[4]) { return (a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2]) + (a[3] * b[3]); }

Camera camera_create(V3 eye, V3 target, V3 up, double aspect, double vertical_fov_deg) {
  Camera c;
  const double half_height = std::tan(0.5 * (vertical_fov_deg * kPi / 180.0));
  const double half_width = aspect * half_height;
  c.view.lower_left_x = -half_width;
  c.view.lower_left_y = -half_height;
  c.view.view_x = 2.0 * half_width;
  c.view.view_y = 2.0 * half_height;
  const V3 zp = v3_normalize(v3_sub(target, eye));
  const V3 xp = v3_normalize(v3_cross(zp, v3_normalize(up)));
  const V3 yp = v3_normalize(v3_cross(xp, zp));
  const double rows[4][4] = {{xp.x, xp.y, xp.z, -v3_dot(eye, xp)},
                             {yp.x, yp.y, yp.z, -v3_dot(eye, yp)},
                             {-zp.x, -zp.y, -zp.z, v3_dot(eye, zp)},
                             {0.0, 0.0, 0.0, 1.0}};
  std::memcpy(c.m, rows, sizeof rows);
  return c;
}

// Camera.transform = Mat4.transform look_at (camera.ml:39-43,91)
V3 camera_transform(const Camera& c, V3 p) {
  const double v[4] = {p.x, p.y, p.z, 1.0};
  const double x = dot4(v, c.m[0]), y = dot4(v, c.m[1]), z = dot4(v, c.m[2]), w = dot4(v, c.m[3]);
  return v3_scale(v3(x, y, z), 1.0 / w);
}

// ---------------------------------------------------------------- Base.Random over OCaml 5's LXM
// Base.Random.init 42 seeds Stdlib.Random (LXM L64X128, MD5-based seeding); Base.Random.float draws TWO
// 30-bit `bits` per float: ((r1 * 2^-30) + r2) * 2^-30, retried if it rounds to 1.0 (third-party:
// base/src/random.ml, stdlib/random.ml, runtime/prng.c -- pinned by the golden PNG through the oracle).
class Md5 {
 public:
  static void digest(const uint8_t* msg, size_t len, uint8_t out[16]) {
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    std::vector<uint8_t> buf(msg, msg + len);
    buf.push_back(0x80);
    while (buf.size() % 64 != 56) buf.push_back(0);
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) buf.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t off = 0; off < buf.size(); off += 64) block(h, &buf[off]);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (8 * j));
  }

 private:
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  static void block(uint32_t h[4], const uint8_t* p) {
    static const int shift[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
    uint32_t w[16];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
    for (int i = 0; i < 64; ++i) {
      const int round = i / 16;
      uint32_t f;
      int g;
      switch (round) {
        case 0: f = (b & c) | (~b & d); g = i; break;
        case 1: f = (d & b) | (~d & c); g = (5 * i + 1) & 15; break;
        case 2: f = b ^ c ^ d; g = (3 * i + 5) & 15; break;
        default: f = c ^ (b | ~d); g = (7 * i) & 15; break;
      }
      // K[i] = floor(2^32 * |sin(i + 1)|)
      const uint32_t k = (uint32_t)(long long)std::floor(std::fabs(std::sin((double)(i + 1))) * 4294967296.0);
      const uint32_t tmp = d;
      d = c;
      c = b;
      b = b + rol(a + f + k + w[g], shift[round][i & 3]);
      a = tmp;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
  }
};

class BaseRandom {
 public:
  explicit BaseRandom(int64_t seed) {
    uint8_t b[9];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)((uint64_t)seed >> (8 * i));
    uint8_t d1[16], d2[16];
    b[8] = 1; Md5::digest(b, 9, d1);
    b[8] = 2; Md5::digest(b, 9, d2);
    a_ = le64(d1) | 1ull;
    s_ = le64(d1 + 8);
    x0_ = le64(d2);
    x1_ = le64(d2 + 8);
    if (x0_ == 0) x0_ = 1;
    if (x1_ == 0) x1_ = 2;
  }
  // Base.Random.float bound
  double next_float(double bound) {
    for (;;) {
      const double r1 = (double)bits30();
      const double r2 = (double)bits30();
      const double result = ((r1 * 0x1p-30) + r2) * 0x1p-30;
      if (result < 1.0) return result * bound;
    }
  }

 private:
  static uint64_t le64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() { // caml_lxm_next
    uint64_t z = s_ + x0_;
    z = (z ^ (z >> 32)) * 0xdaba0b6eb09322e3ull;
    z = (z ^ (z >> 32)) * 0xdaba0b6eb09322e3ull;
    z ^= z >> 32;
    s_ = s_ * 0xd1342543de82ef95ull + a_;
    x1_ ^= x0_;
    x0_ = rotl(x0_, 24) ^ x1_ ^ (x1_ << 16);
    x1_ = rotl(x1_, 37);
    return z;
  }
  int64_t bits30() { return (int64_t)(next() & 0x3fffffffull); } // Random.State.bits
  uint64_t a_, s_, x0_, x1_;
};

}  // namespace

// ---------------------------------------------------------------- owned scene description
struct pth_scene {
  ptx_scene_desc d{};
  std::vector<double> sx, sy, sz, sr;
  std::vector<int32_t> sm;
  std::vector<double> vx, vy, vz;
  std::vector<int32_t> ti, tm;
  std::vector<double> tuv;
  std::vector<double> floor_v, floor_uv;
  std::vector<int32_t> floor_m;
  std::vector<ptx_material> mats;
  std::vector<ptx_texture> texs;

  int solid(double r, double g, double b) {
    ptx_texture t{};
    t.kind = PTX_TEX_SOLID;
    t.even[0] = r; t.even[1] = g; t.even[2] = b;
    texs.push_back(t);
    return (int)texs.size() - 1;
  }
  int checker(int w, int h, const double even[3], const double odd[3]) {
    ptx_texture t{};
    t.kind = PTX_TEX_CHECKER;
    t.width = w; t.height = h;
    std::memcpy(t.even, even, sizeof t.even);
    std::memcpy(t.odd, odd, sizeof t.odd);
    texs.push_back(t);
    return (int)texs.size() - 1;
  }
  int material(int kind, int tex, double index = 0.0) {
    ptx_material m{};
    m.kind = kind; m.texture = tex; m.index = index;
    mats.push_back(m);
    return (int)mats.size() - 1;
  }
  void sphere(V3 c, double r, int m) {
    sx.push_back(c.x); sy.push_back(c.y); sz.push_back(c.z); sr.push_back(r); sm.push_back(m);
  }
  int vertex(V3 p) {
    vx.push_back(p.x); vy.push_back(p.y); vz.push_back(p.z);
    return (int)vx.size() - 1;
  }
  void triangle(int a, int b, int c, const double uv[6], int m) {
    ti.push_back(a); ti.push_back(b); ti.push_back(c);
    tuv.insert(tuv.end(), uv, uv + 6);
    tm.push_back(m);
  }
  void sync() {
    d.n_spheres = (int)sx.size();
    d.sphere_x = sx.data(); d.sphere_y = sy.data(); d.sphere_z = sz.data(); d.sphere_r = sr.data(); d.sphere_material = sm.data();
    d.n_vertices = (int)vx.size();
    d.vertex_x = vx.data(); d.vertex_y = vy.data(); d.vertex_z = vz.data();
    d.n_triangles = (int)tm.size();
    d.tri_indices = ti.data(); d.tri_uv = tuv.data(); d.tri_material = tm.data();
    d.n_floor_triangles = (int)floor_m.size();
    d.floor_vertices = floor_v.data(); d.floor_uv = floor_uv.data(); d.floor_material = floor_m.data();
    d.n_materials = (int)mats.size(); d.materials = mats.data();
    d.n_textures = (int)texs.size(); d.textures = texs.data();
  }
};

namespace {

void sky(ptx_background& bg) { // shirley_spheres/bin/main.ml:104-110
  bg = ptx_background{};
  bg.kind = PTX_BG_SKY;
  bg.horizon[0] = bg.horizon[1] = bg.horizon[2] = 1.0; // Color.white
  bg.zenith[0] = 0.5; bg.zenith[1] = 0.7; bg.zenith[2] = 1.0; // escape_color
}

const double kT00[2] = {0.0, 0.0}, kT01[2] = {0.0, 1.0}, kT10[2] = {1.0, 0.0}, kT11[2] = {1.0, 1.0};
void uv3(double out[6], const double a[2], const double b[2], const double c[2]) {
  out[0] = a[0]; out[1] = a[1]; out[2] = b[0]; out[3] = b[1]; out[4] = c[0]; out[5] = c[1];
}

struct Tri {
  V3 a, b, c;
  double uv[6];
  int material;
};

// quad ~material a u v = triangle_fan [a,t00; b,t10; c,t11; d,t01] with b = a+v, c = b+u, d = a+u
// (cornell-box/bin/main.ml:30-48).  triangle_fan accumulates by consing, so the list is [a c d; a b c].
std::vector<Tri> quad(int material, V3 a, V3 u, V3 v) {
  const V3 b = v3_add(a, v), c = v3_add(b, u), d = v3_add(a, u);
  Tri t1{a, c, d, {}, material}, t2{a, b, c, {}, material};
  uv3(t1.uv, kT00, kT11, kT01);
  uv3(t2.uv, kT00, kT10, kT11);
  return {t1, t2};
}

// Base List.concat_no_order = fold ~init:[] ~f:(fun acc l -> rev_append l acc): last list first, each reversed
std::vector<Tri> concat_no_order(const std::vector<std::vector<Tri>>& lists) {
  std::vector<Tri> out;
  for (auto it = lists.rbegin(); it != lists.rend(); ++it)
    for (auto jt = it->rbegin(); jt != it->rend(); ++jt) out.push_back(*jt);
  return out;
}

}  // namespace

extern "C" {

const ptx_scene_desc* pth_scene_desc(pth_scene* s) {
  if (!s) return nullptr;
  s->sync();
  return &s->d;
}
void pth_scene_free(pth_scene* s) { delete s; }

void pth_camera_create(const double eye[3], const double target[3], const double up[3], double aspect, double fov_deg,
                       ptx_camera* view_out, double look_at_out[16]) {
  const Camera c = camera_create(v3(eye[0], eye[1], eye[2]), v3(target[0], target[1], target[2]), v3(up[0], up[1], up[2]), aspect, fov_deg);
  if (view_out) *view_out = c.view;
  if (look_at_out) std::memcpy(look_at_out, c.m, sizeof c.m);
}

// shirley_spheres/bin/main.ml: Shirley_spheres.spheres () under Random.init seed, camera (width // height),
// every sphere moved to camera space; no_simd picks Array_leaf (cutoff 4) over Simd_leaf (cutoff 16).
pth_scene* pth_scene_shirley(int32_t width, int32_t height, int32_t no_simd, int64_t seed) {
  pth_scene* s = new pth_scene();
  BaseRandom rng(seed);
  const Camera cam = camera_create(v3(13.0, 2.0, 4.5), v3(0.0, 0.0, 0.0), v3(0.0, 1.0, 0.0), (double)width / (double)height, 20.0);
  const double ga[3] = {0.2, 0.3, 0.1}, gb[3] = {0.9, 0.9, 0.9};
  s->sphere(v3(0.0, -1000.0, 0.0), 1000.0, s->material(PTX_MAT_LAMBERTIAN, s->checker(1000, 2000, ga, gb))); // ground
  const int glass = s->material(PTX_MAT_DIELECTRIC, 0, 1.5); // Material.glass
  const int metal = s->material(PTX_MAT_METAL, s->solid(0.7, 0.6, 0.5));
  const int blue = s->material(PTX_MAT_LAMBERTIAN, s->solid(0.1, 0.1, 0.7));
  s->sphere(v3(-4.0, 1.0, 0.0), 1.0, glass);
  s->sphere(v3(0.0, 1.0, 0.0), 1.0, metal);
  s->sphere(v3(4.0, 1.0, 0.0), 1.0, blue);
  for (int a = -11; a <= 11; ++a) {
    for (int b = -11; b <= 11; ++b) {
      const double x = (double)a + (0.9 * rng.next_float(1.0)); // perturb
      const double z = (double)b + (0.9 * rng.next_float(1.0));
      const double radius = 0.2;
      const V3 center = v3(x, radius, z);
      if (!(v3_quadrance(v3_sub(v3(4.0, radius, 0.0), center)) > 0.81)) continue;
      const double roll = rng.next_float(1.0);
      int material;
      if (roll < 0.8) { // random_lambertian: random_v3 () * random_v3 ()
        double p[3], q[3];
        for (double& v : p) v = rng.next_float(1.0);
        for (double& v : q) v = rng.next_float(1.0);
        material = s->material(PTX_MAT_LAMBERTIAN, s->solid(q[0] * p[0], q[1] * p[1], q[2] * p[2]));
      } else if (roll < 0.95) {
        const double g = (0.5 * rng.next_float(1.0)) + 0.5;
        material = s->material(PTX_MAT_METAL, s->solid(g, g, g));
      } else {
        material = glass;
      }
      s->sphere(center, radius, material);
    }
  }
  for (size_t i = 0; i < s->sx.size(); ++i) { // Sphere.transform ~f:(Camera.transform camera)
    const V3 c = camera_transform(cam, v3(s->sx[i], s->sy[i], s->sz[i]));
    s->sx[i] = c.x; s->sy[i] = c.y; s->sz[i] = c.z;
  }
  s->d.camera = cam.view;
  sky(s->d.background);
  s->d.leaf_kind = no_simd ? PTX_LEAF_ARRAY : PTX_LEAF_SIMD;
  s->d.length_cutoff = no_simd ? 4 : 16; // leaf_size () = 16 (lib.rs:13)
  s->d.num_bins = 32;
  s->sync();
  return s;
}

// cornell-box/bin/main.ml geometry for the path integrator; ceiling_emit is the documented emitter
// extension (the reference lights this scene with a photon-map point light the path integrator ignores).
pth_scene* pth_scene_cornell(int32_t width, int32_t height, double ceiling_emit) {
  pth_scene* s = new pth_scene();
  const double fov = (2.0 * std::atan(0.5)) * 180.0 / kPi;
  const Camera cam = camera_create(v3(0.5, 0.5, -1.0), v3(0.5, 0.5, 0.0), v3(0.0, 1.0, 0.0), (double)width / (double)height, fov);
  const V3 ux = v3(1.0, 0.0, 0.0), uy = v3(0.0, 1.0, 0.0), uz = v3(0.0, 0.0, 1.0), org = v3(0.0, 0.0, 0.0);
  auto emit = [&](const std::vector<Tri>& tris) {
    for (const Tri& t : tris) {
      const int a = s->vertex(camera_transform(cam, t.a)), b = s->vertex(camera_transform(cam, t.b)), c = s->vertex(camera_transform(cam, t.c));
      s->triangle(a, b, c, t.uv, t.material);
    }
  };
  { // light_enclosure' (main.ml:190-210)
    const int encl = s->material(PTX_MAT_METAL, s->solid(0.30, 0.999, 0.30));
    const double r = 0.05;
    const V3 rx = v3_scale(ux, r), ry = v3_scale(uy, r), rz = v3_scale(uz, r), lc = v3(0.5, 0.82, 0.5);
    const V3 a = v3_sub(v3_sub(v3_sub(lc, rx), ry), rz);
    const V3 b = v3_add(v3_sub(v3_add(lc, rx), ry), rz);
    auto q2 = [&](V3 p, V3 u, V3 v) { return quad(encl, p, v3_scale(u, 2.0), v3_scale(v, 2.0)); };
    emit(concat_no_order({q2(a, rz, ry), q2(a, ry, rx), q2(b, v3_neg(rz), ry), q2(b, rx, ry)}));
  }
  { // empty_box (main.ml:52-68)
    const int red = s->material(PTX_MAT_LAMBERTIAN, s->solid(0.7, 0.0, 0.0));
    const int blue = s->material(PTX_MAT_LAMBERTIAN, s->solid(0.0, 0.0, 0.7));
    const int grey = s->material(PTX_MAT_LAMBERTIAN, s->solid(0.7, 0.7, 0.7));
    const double ca[3] = {0.2, 0.3, 0.1}, cb[3] = {0.9, 0.9, 0.9};
    const int checks = s->material(PTX_MAT_LAMBERTIAN, s->checker(10, 10, ca, cb));
    const int ceiling = s->material(PTX_MAT_LAMBERTIAN, s->solid(0.7, 0.7, 0.7));
    for (double& e : s->mats[(size_t)ceiling].emit) e = ceiling_emit;
    emit(concat_no_order({quad(red, org, uz, uy), quad(blue, ux, uz, uy), quad(checks, org, ux, uz), quad(ceiling, uy, ux, uz), quad(grey, uz, ux, uy)}));
  }
  { // spheres (main.ml:70-91)
    const double radius = 0.20;
    const int m_metal = s->material(PTX_MAT_METAL, s->solid(1.0, 1.0, 1.0));
    const int m_glass = s->material(PTX_MAT_DIELECTRIC, 0, 1.5);
    const int m_back = s->material(PTX_MAT_LAMBERTIAN, s->solid(0.75, 0.75, 0.75));
    s->sphere(camera_transform(cam, v3(1.0 - 0.1 - radius, radius, 1.0 - 0.2 - radius)), radius, m_metal);
    s->sphere(camera_transform(cam, v3(0.1 + radius, 0.1 + radius, 0.2 + radius)), radius, m_glass);
    const double big = 10.0;
    s->sphere(camera_transform(cam, v3(0.5, 0.5, -2.0 - big)), big, m_back);
  }
  s->d.camera = cam.view;
  s->d.background = ptx_background{};
  s->d.background.kind = PTX_BG_BLACK;
  s->d.leaf_kind = PTX_LEAF_ARRAY;
  s->d.length_cutoff = 2;
  s->d.num_bins = 32;
  s->sync();
  return s;
}

// Synthetic "ganesha-like" mesh (the real ganesha.ply is not in the reference repository): a closed
// lat-long surface of about n_target triangles displaced by seeded lobes; world-space vertices rounded to
// float32 like a PLY `float` property.
static void ganesha_like_mesh(int32_t n_target, uint64_t seed, std::vector<double>& X, std::vector<double>& Y,
                              std::vector<double>& Z, std::vector<int32_t>& tri) {
  int nv = (int)std::floor(std::sqrt((double)n_target / 4.0));
  if (nv < 4) nv = 4;
  const int nu = 2 * nv;
  uint64_t st = seed;
  auto splitmix = [&]() {
    uint64_t z = (st += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  auto unit = [&]() { return (double)(splitmix() >> 11) * 0x1p-53; };
  struct Lobe { double x, y, z, amp, sharp; };
  std::vector<Lobe> lobes(24);
  for (Lobe& l : lobes) {
    const double u1 = unit(), u2 = unit(), u3 = unit(), u4 = unit();
    const double zc = 2.0 * u1 - 1.0, ph = 2.0 * kPi * u2, rr = std::sqrt(1.0 - zc * zc);
    l = Lobe{rr * std::cos(ph), zc, rr * std::sin(ph), 0.10 + 0.35 * u3, 4.0 + 28.0 * u4};
  }
  const V3 centre = v3(328.0, 42.0, 20.0), radii = v3(30.0, 44.0, 26.0);
  std::vector<int> vid((size_t)nu * (size_t)(nv + 1));
  for (int j = 0; j <= nv; ++j) {
    for (int i = 0; i < nu; ++i) {
      const double th = kPi * (double)j / (double)nv, ph = 2.0 * kPi * (double)i / (double)nu;
      V3 n = v3(std::sin(th) * std::cos(ph), std::cos(th), std::sin(th) * std::sin(ph));
      if (j == 0) n = v3(0.0, 1.0, 0.0);
      if (j == nv) n = v3(0.0, -1.0, 0.0);
      double disp = 1.0;
      for (const Lobe& l : lobes) disp += l.amp * std::exp(l.sharp * ((n.x * l.x + n.y * l.y + n.z * l.z) - 1.0));
      disp += 0.02 * std::sin(37.0 * ph) * std::sin(29.0 * th);
      const V3 p = v3(centre.x + radii.x * disp * n.x, centre.y + radii.y * disp * n.y, centre.z + radii.z * disp * n.z);
      vid[(size_t)j * nu + i] = (int)X.size();
      X.push_back((double)(float)p.x); // a PLY `float` property
      Y.push_back((double)(float)p.y);
      Z.push_back((double)(float)p.z);
    }
  }
  for (int j = 0; j < nv; ++j)
    for (int i = 0; i < nu; ++i) {
      const int i1 = (i + 1) % nu;
      const int a = vid[(size_t)j * nu + i], b = vid[(size_t)j * nu + i1], c = vid[(size_t)(j + 1) * nu + i1], d = vid[(size_t)(j + 1) * nu + i];
      if (j != 0) { tri.push_back(a); tri.push_back(b); tri.push_back(c); }
      if (j != nv - 1) { tri.push_back(a); tri.push_back(c); tri.push_back(d); }
    }
}

} // extern "C"

// ganesha/bin/main.ml over a world-space mesh: camera (:30-35), Mesh.create (vertices to camera space, :74-79),
// Lambertian (.1,.7,.2) with tex_coords (t00,t01,t11) (:111-115), leaf cutoff 8 (:158), the 500x500 checker floor
// tested before the tree (:205-260).  sky: Shirley's sky background (documented extension -- the reference
// lights this scene with photon-map spot lights the path integrator ignores).
pth_scene* pth_scene_ganesha_from_mesh(int32_t width, int32_t height, const std::vector<double>& X, const std::vector<double>& Y,
                                       const std::vector<double>& Z, const std::vector<int32_t>& tri, bool sky_bg) {
  pth_scene* s = new pth_scene();
  const Camera cam = camera_create(v3(328.0, 70.282, 345.0), v3(328.0, 10.0, 0.0), v3(-0.00212272, 0.998201, -0.0599264), (double)width / (double)height, 30.0);
  const int mat = s->material(PTX_MAT_LAMBERTIAN, s->solid(0.1, 0.7, 0.2));
  for (size_t i = 0; i < X.size(); ++i) s->vertex(camera_transform(cam, v3(X[i], Y[i], Z[i])));
  double uv[6];
  uv3(uv, kT00, kT01, kT11);
  for (size_t t = 0; t + 2 < tri.size(); t += 3) s->triangle(tri[t], tri[t + 1], tri[t + 2], uv, mat);
  { // Floor (main.ml:205-245) under the tree's bbox = union of the triangle bboxes
    Box bb{};
    for (size_t t = 0; t < s->tm.size(); ++t) {
      Box tb{};
      for (int k = 0; k < 3; ++k) {
        const int vi = s->ti[3 * t + k];
        Box pb;
        pb.mn = pb.mx = v3(s->vx[(size_t)vi], s->vy[(size_t)vi], s->vz[(size_t)vi]);
        tb = k == 0 ? pb : box_union(tb, pb);
      }
      bb = t == 0 ? tb : box_union(bb, tb);
    }
    const V3 ctr = box_center(bb);
    const V3 center = v3(ctr.x, bb.mn.y, ctr.z);
    const double size = 5000.0;
    const V3 xp = v3_scale(v3(1.0, 0.0, 0.0), size), zp = v3_scale(v3(0.0, 0.0, 1.0), size);
    const V3 pa = v3_add(center, v3_neg(v3_add(xp, zp)));
    const V3 pb = v3_add(pa, v3_scale(xp, 2.0));
    const V3 pc = v3_add(pb, v3_scale(zp, 2.0));
    const V3 pd = v3_add(pa, v3_scale(zp, 2.0));
    const double ea[3] = {0.2, 0.3, 0.1}, eb[3] = {0.9, 0.9, 0.9};
    const int fm = s->material(PTX_MAT_LAMBERTIAN, s->checker(500, 500, ea, eb));
    const V3 f1[3] = {pa, pb, pc}, f2[3] = {pa, pc, pd};
    for (const V3* f : {f1, f2})
      for (int k = 0; k < 3; ++k) {
        s->floor_v.push_back(f[k].x); s->floor_v.push_back(f[k].y); s->floor_v.push_back(f[k].z);
      }
    double u1[6], u2[6];
    uv3(u1, kT00, kT01, kT11);
    uv3(u2, kT00, kT11, kT10);
    s->floor_uv.insert(s->floor_uv.end(), u1, u1 + 6);
    s->floor_uv.insert(s->floor_uv.end(), u2, u2 + 6);
    s->floor_m = {fm, fm};
  }
  s->d.camera = cam.view;
  if (sky_bg) sky(s->d.background);
  else {
    s->d.background = ptx_background{};
    s->d.background.kind = PTX_BG_BLACK;
  }
  s->d.leaf_kind = PTX_LEAF_ARRAY;
  s->d.length_cutoff = 8; // ganesha/bin/main.ml:158
  s->d.num_bins = 32;
  s->sync();
  return s;
}

extern "C" {

pth_scene* pth_scene_ganesha_like(int32_t width, int32_t height, int32_t n_target, uint64_t seed) {
  std::vector<double> X, Y, Z;
  std::vector<int32_t> tri;
  ganesha_like_mesh(n_target, seed, X, Y, Z, tri);
  return pth_scene_ganesha_from_mesh(width, height, X, Y, Z, tri, true);
}

// writes the synthetic mesh as a binary little-endian PLY with the real model's layout
// (element vertex: float x y z; element face: list uint8 int vertex_indices)
int32_t pth_write_ganesha_like_ply(const char* path, int32_t n_target, uint64_t seed) {
  std::vector<double> X, Y, Z;
  std::vector<int32_t> tri;
  ganesha_like_mesh(n_target, seed, X, Y, Z, tri);
  FILE* f = std::fopen(path, "wb");
  if (!f) return -1;
  std::fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment synthetic ganesha-like mesh\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\nelement face %zu\nproperty list uint8 int vertex_indices\nend_header\n", X.size(), tri.size() / 3);
  for (size_t i = 0; i < X.size(); ++i) {
    const float v[3] = {(float)X[i], (float)Y[i], (float)Z[i]};
    std::fwrite(v, 4, 3, f);
  }
  for (size_t t = 0; t + 2 < tri.size(); t += 3) {
    const uint8_t n = 3;
    std::fwrite(&n, 1, 1, f);
    std::fwrite(&tri[t], 4, 3, f);
  }
  std::fclose(f);
  return 0;
}


// Lights of the two photon-mapped reference scenes, in camera space.
// cornell-box/bin/main.ml:183,225-228: one point light at Camera.transform camera (0.5, 0.82, 0.5), power 2, white
int32_t pth_lights_cornell(int32_t width, int32_t height, ptx_light* out) {
  const double fov = (2.0 * std::atan(0.5)) * 180.0 / kPi;
  const Camera cam = camera_create(v3(0.5, 0.5, -1.0), v3(0.5, 0.5, 0.0), v3(0.0, 1.0, 0.0), (double)width / (double)height, fov);
  const V3 pos = camera_transform(cam, v3(0.5, 0.82, 0.5));
  *out = ptx_light{};
  out->kind = PTX_LIGHT_POINT;
  out->position[0] = pos.x; out->position[1] = pos.y; out->position[2] = pos.z;
  out->color[0] = out->color[1] = out->color[2] = 1.0;
  out->power = 2.0;
  return 1;
}
// ganesha/bin/main.ml:267-282: two spot lights placed from the mesh's camera-space bounding box
int32_t pth_lights_ganesha(pth_scene* s, ptx_light* out) {
  if (!s || s->tm.empty()) return 0;
  Box bb{};
  for (size_t t = 0; t < s->tm.size(); ++t)
    for (int k = 0; k < 3; ++k) {
      const int vi = s->ti[3 * t + k];
      Box pb;
      pb.mn = pb.mx = v3(s->vx[(size_t)vi], s->vy[(size_t)vi], s->vz[(size_t)vi]);
      bb = (t == 0 && k == 0) ? pb : box_union(bb, pb);
    }
  const V3 center = box_center(bb);
  const V3 v = v3_sub(bb.mx, center);
  const V3 position = v3_add(bb.mx, v3_add(v3_scale(v, 3.0), v3_scale(v3(0.0, 0.0, 1.0), -400.0)));
  const V3 direction = v3_sub(center, position);
  out[0] = ptx_light{};
  out[0].kind = PTX_LIGHT_SPOT;
  out[0].power = 10000.0;
  out[0].position[0] = position.x; out[0].position[1] = position.y; out[0].position[2] = position.z;
  out[0].direction[0] = direction.x; out[0].direction[1] = direction.y; out[0].direction[2] = direction.z;
  out[0].color[0] = out[0].color[1] = out[0].color[2] = 1.0;
  out[1] = ptx_light{};
  out[1].kind = PTX_LIGHT_SPOT;
  out[1].power = 3000.0;
  out[1].position[2] = 1.0;
  out[1].direction[0] = -0.0; out[1].direction[1] = -0.0; out[1].direction[2] = -1.0; // ~-V3.unit_z
  out[1].color[0] = out[1].color[1] = out[1].color[2] = 1.0;
  return 2;
}

} // extern "C"
