/* pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A literal C restatement of the per-pixel sampling path of dalev/path-tracer-ocaml
 * (reference checked out read-only at /root/reference; every function below cites
 * the file:line it follows).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libptx_hip.so) never does.
 *
 * PINNING: the reference cannot be built here (no ocaml/dune/opam/cargo/rustc in the
 * image), so this restatement is pinned by the reference's own
 * fixtures only: every alcotest / core_bench assertion re-expressed in
 * tests/test_oracle_kat.py and the golden image shirley-spheres.png (README.md:3,7)
 * compared in tests/test_oracle_golden.py.  Third-party arithmetic outside
 * /root/reference (OCaml 5 Random = LXM L64X128 + MD5 seeding; Base Float.min/max/
 * to_int/clamp; libm) is restated from its published algorithm and pinned by that
 * image to 8 bits only.
 *
 * Math modes (orc_set_math): 0 = pt_math.h (the shared host/device functions the GPU
 * uses; default, used for bit-parity), 1 = the platform libm the OCaml runtime would
 * call (Float.hypot / sin / cos / acos / atan2 / ( ** )), used to measure the ulp gap.
 *
 * Build: gcc -O2 -march=x86-64-v3 -ffp-contract=off (see oracle/Makefile).  OCaml
 * never contracts a*b+c; FMAs appear only where the reference writes Float.fma.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/ptx.h"
#include "../path_tracer_ocaml_amd/csrc/pt_math.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ math dispatch */
static int g_math_mode = 0;
ORC_API void orc_set_math(int mode) { g_math_mode = mode; }
ORC_API int orc_get_math(void) { return g_math_mode; }

static inline double m_hypot(double x, double y) { return g_math_mode ? hypot(x, y) : pt_hypot(x, y); }
static inline double m_sin(double x) { return g_math_mode ? sin(x) : pt_sin(x); }
static inline double m_cos(double x) { return g_math_mode ? cos(x) : pt_cos(x); }
static inline double m_acos(double x) { return g_math_mode ? acos(x) : pt_acos(x); }
static inline double m_atan2(double y, double x) { return g_math_mode ? atan2(y, x) : pt_atan2(y, x); }
/* ( ** ) 5.0 -- material.ml:19,37 */
static inline double m_pow5(double x) { return g_math_mode ? pow(x, 5.0) : pt_pow5(x); }

ORC_API double orc_math(int fn, double a, double b) {
  switch (fn) {
    case 0: return m_hypot(a, b);
    case 1: return m_sin(a);
    case 2: return m_cos(a);
    case 3: return m_acos(a);
    case 4: return m_atan2(a, b);
    case 5: return m_pow5(a);
    case 6: return sqrt(a);
    case 7: return a / b;
    case 8: return fma(a, b, b);
    /* the product's fused normalisation scalars (pt_rnorm3 / pt_rnorm_frame) and bare sqrt / reciprocal / division sequences
     * against the literal expressions they stand for */
    case 9: return 1.0 / m_hypot(a, m_hypot(b, a - b));                    /* V3.normalize, affine.ml:65-68 */
    case 10: return 1.0 / m_hypot(m_hypot(1.0 + a, b), m_hypot(a - b, 0.0)); /* Quaternion.normalize, quaternion.ml:11-15 */
    case 11: return sqrt(a);
    case 12: return 1.0 / a;
    case 13: return a / b;
    case 14: return sqrt(a);
    /* the fused forms themselves, host build (tests/test_math.py compares them with cases 9 / 10 without a GPU) */
    case 19: return pt_rnorm3(a, b, a - b);
    case 20: return pt_rnorm_frame(1.0 + a, b, a - b);
  }
  return NAN;
}

ORC_API void orc_math_vec(int fn, int64_t n, const double* a, const double* b, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = orc_math(fn, a[i], b ? b[i] : 0.0);
}

/* ------------------------------------------------------------------ V3 (affine.ml) */
typedef struct { double x, y, z; } v3;

static inline v3 v3_make(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }  /* affine.ml:45 */
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }  /* :46 */
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }  /* :47 */
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }                       /* :49 */
/* V3.fma u v w = u*v+w fused per component, affine.ml:53 */
static inline v3 v3_fma(v3 u, v3 v, v3 w) { return v3_make(fma(u.x, v.x, w.x), fma(u.y, v.y, w.y), fma(u.z, v.z, w.z)); }
/* V3.dot, affine.ml:60 */
static inline double v3_dot(v3 v, v3 w) { return fma(v.x, w.x, fma(v.y, w.y, v.z * w.z)); }
/* V3.scale v s = map (( *. ) s), affine.ml:61 */
static inline v3 v3_scale(v3 v, double s) { return v3_make(s * v.x, s * v.y, s * v.z); }
static inline double v3_quadrance(v3 v) { return v3_dot(v, v); }                           /* :62 */
/* V3.lerp, affine.ml:63 */
static inline v3 v3_lerp(double t, v3 v, v3 w) { return v3_add(v3_scale(v, 1.0 - t), v3_scale(w, t)); }
/* V3.normalize, affine.ml:65-68 */
static inline v3 v3_normalize(v3 v) {
  double scalar = 1.0 / m_hypot(v.x, m_hypot(v.y, v.z));
  return v3_scale(v, scalar);
}
/* V3.cross, affine.ml:70-73 : h w x y z = fma w x (-(y*z)) */
static inline double cross_h(double w, double x, double y, double z) { return fma(w, x, -(y * z)); }
static inline v3 v3_cross(v3 p, v3 q) {
  double a = p.x, b = p.y, c = p.z, d = q.x, e = q.y, f = q.z;
  return v3_make(cross_h(b, f, c, e), cross_h(c, d, a, f), cross_h(a, e, b, d));
}
static inline double v3_axis(v3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
/* V3.min_coord / max_coord, affine.ml:56-57 (Base Float.min/max: NaN-propagating) */
static inline double v3_min_coord(v3 v) { return pt_base_min(v.x, pt_base_min(v.y, v.z)); }
static inline double v3_max_coord(v3 v) { return pt_base_max(v.x, pt_base_max(v.y, v.z)); }

/* ------------------------------------------------------------------ Ray (ray.ml) */
typedef struct { v3 origin, direction, direction_inv; } ray_t;
/* Ray.create, ray.ml:7-10 */
static inline ray_t ray_create(v3 origin, v3 direction) {
  ray_t r;
  r.origin = origin;
  r.direction = direction;
  r.direction_inv = v3_make(1.0 / direction.x, 1.0 / direction.y, 1.0 / direction.z);
  return r;
}
/* Ray.point_at, ray.ml:15 */
static inline v3 ray_point_at(const ray_t* r, double t) { return v3_add(r->origin, v3_scale(r->direction, t)); }

/* ------------------------------------------------------------------ sampler (low_discrepancy_sequence.ml) */
/* phi_approx, low_discrepancy_sequence.ml:8-17 */
static double lds_phi_approx(int d) {
  double dp = 1.0 / ((double)d + 1.0);
  double x = 2.0;
  for (int it = 0; it < 100000; ++it) {
    double xp = pow(1.0 + x, dp);
    if (x == xp) return x;
    x = xp;
  }
  return x;
}
/* alpha, :22-25 */
ORC_API void orc_lds_alpha(int dimension, double* out) {
  double phi = lds_phi_approx(dimension);
  for (int i = 0; i < dimension; ++i) out[i] = 1.0 / pow(phi, (double)(i + 1));
}
ORC_API double orc_lds_phi(int dimension) { return lds_phi_approx(dimension); }
/* get, :33-36 with fractional/clamp :19-20 */
static inline double lds_get(const double* alpha, int offset, int dimension) {
  double a = alpha[dimension];
  double x = 0.5 + (a * (double)(1 + offset));
  return x - trunc(x);
}
ORC_API double orc_lds_get(int n_dim, int offset, int dimension) {
  double alpha[256];
  if (n_dim > 256) return NAN;
  orc_lds_alpha(n_dim, alpha);
  return lds_get(alpha, offset, dimension);
}

/* ------------------------------------------------------------------ filter kernel (filter_kernel.ml) */
/* exact rationals (the reference uses `num`); numerators stay tiny */
typedef struct { int64_t n, d; } rat;
static int64_t gcd64(int64_t a, int64_t b) { if (a < 0) a = -a; if (b < 0) b = -b; while (b) { int64_t t = a % b; a = b; b = t; } return a ? a : 1; }
static rat rat_make(int64_t n, int64_t d) { if (d < 0) { n = -n; d = -d; } int64_t g = gcd64(n, d); rat r = {n / g, d / g}; return r; }
static rat rat_add(rat a, rat b) { return rat_make(a.n * b.d + b.n * a.d, a.d * b.d); }
static rat rat_sub(rat a, rat b) { return rat_make(a.n * b.d - b.n * a.d, a.d * b.d); }
static rat rat_mul(rat a, rat b) { return rat_make(a.n * b.n, a.d * b.d); }
static int64_t floor_div(int64_t n, int64_t d) { int64_t q = n / d; if ((n % d != 0) && ((n < 0) != (d < 0))) --q; return q; }
static int64_t rat_floor(rat a) { return floor_div(a.n, a.d); }
static int64_t rat_ceil(rat a) { return -floor_div(-a.n, a.d); }
/* mod_num n one : n - floor(n) */
static rat rat_frac(rat a) { return rat_sub(a, rat_make(rat_floor(a), 1)); }

static int64_t pow_falling(int64_t n, int64_t k) { return k == 0 ? 1 : n * pow_falling(n - 1, k - 1); }  /* :27 */
static int64_t binomial(int64_t n, int64_t k) { return pow_falling(n, k) / pow_falling(k, k); }           /* :28-29 */

/* Binomial.create ~order ~pixel_radius, filter_kernel.ml:49-85.  data has (2r+1)^2 weights,
 * row-major; returns the dim. */
ORC_API int orc_filter_binomial(int order, int pixel_radius, double* data /* dim*dim */, double* w1d /* dim, optional */) {
  int f_width = 1 + 2 * pixel_radius;
  rat ratio = rat_make(order, f_width);
  int64_t coeffs[64];
  if (order > 64 || f_width > 64) return -1;
  for (int k = 0; k < order; ++k) coeffs[k] = binomial(order - 1, k);
  double w[64];
  rat one = rat_make(1, 1);
  for (int i = 0; i < f_width; ++i) {
    rat ip = rat_mul(rat_make(i, 1), ratio);
    rat jp = rat_add(ip, ratio);
    int64_t beg = rat_floor(ip);
    int64_t end_ = rat_ceil(jp);
    int64_t len = end_ - beg;
    rat sum = rat_make(0, 1);
    for (int64_t k = 0; k < len; ++k) {
      rat weight;
      if (k == 0) weight = rat_sub(one, rat_frac(ip));
      else if (k == len - 1) weight = rat_sub(one, rat_sub(rat_make(end_, 1), jp));
      else weight = one;
      sum = rat_add(sum, rat_mul(weight, rat_make(coeffs[k + beg], 1)));
    }
    w[i] = (double)sum.n / (double)sum.d; /* float_of_num: nearest */
  }
  double total = 0.0;
  for (int i = 0; i < f_width; ++i) total = total + w[i]; /* fold_left (+.) 0.0, :82 */
  for (int i = 0; i < f_width; ++i) w[i] = w[i] / total;  /* :83 */
  if (w1d) for (int i = 0; i < f_width; ++i) w1d[i] = w[i];
  /* outer_product, :40-47 */
  for (int j = 0; j < f_width * f_width; ++j) data[j] = w[j / f_width] * w[j % f_width];
  return f_width;
}

/* ------------------------------------------------------------------ Tile (tile.ml) */
typedef struct { int row, col, width, height; } tile_t;
static inline int tile_area(tile_t t) { return t.width * t.height; }
/* Tile.split, tile.ml:14-39 (order: loop lhs @ loop rhs) */
static void tile_split_rec(tile_t t, int max_area, tile_t** out, int* n, int* cap) {
  if (tile_area(t) <= max_area) {
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 64; *out = (tile_t*)realloc(*out, sizeof(tile_t) * (size_t)*cap); }
    (*out)[(*n)++] = t;
    return;
  }
  tile_t lhs = t, rhs = t;
  if (t.width > t.height) { /* split_once, :28 */
    int half_w = t.width / 2;
    lhs.width = half_w;
    rhs.col = t.col + half_w;
    rhs.width = t.width - half_w;
  } else {
    int half_h = t.height / 2;
    lhs.height = half_h;
    rhs.row = t.row + half_h;
    rhs.height = t.height - half_h;
  }
  tile_split_rec(lhs, max_area, out, n, cap);
  tile_split_rec(rhs, max_area, out, n, cap);
}
/* returns number of tiles; out gets 4 ints per tile: row, col, width, height */
ORC_API int orc_tile_split(int width, int height, int max_area, int* out, int capacity) {
  tile_t root = {0, 0, width, height};
  tile_t* tiles = NULL;
  int n = 0, cap = 0;
  tile_split_rec(root, max_area, &tiles, &n, &cap);
  for (int i = 0; i < n && i < capacity; ++i) {
    out[4 * i] = tiles[i].row; out[4 * i + 1] = tiles[i].col; out[4 * i + 2] = tiles[i].width; out[4 * i + 3] = tiles[i].height;
  }
  free(tiles);
  return n;
}

/* ------------------------------------------------------------------ Film_tile (film_tile.ml) */
typedef struct {
  tile_t tile;
  int border, width, height; /* pixels image dims */
  double* pixels;            /* (width*height*3), index (y*width+x)*3+c */
  const double* kernel;      /* dim*dim */
  int kdim;
} film_tile_t;

/* Film_tile.create, film_tile.ml:15-21 */
static film_tile_t film_tile_create(tile_t tile, const double* kernel, int pixel_radius) {
  film_tile_t ft;
  ft.tile = tile;
  ft.border = pixel_radius;
  ft.width = tile.width + 2 * pixel_radius;
  ft.height = tile.height + 2 * pixel_radius;
  ft.pixels = (double*)calloc((size_t)ft.width * ft.height * 3, sizeof(double));
  ft.kernel = kernel;
  ft.kdim = 2 * pixel_radius + 1;
  return ft;
}
/* Film_tile.write_pixel, film_tile.ml:23-38 + Filter_kernel.iter, filter_kernel.ml:14-24 */
static void film_tile_write_pixel(film_tile_t* t, int x, int y, v3 color) {
  int border = t->border;
  x = x + border;
  y = y + border;
  int r = t->border, i = 0;
  for (int dy = -r; dy <= r; ++dy) {
    for (int dx = -r; dx <= r; ++dx) {
      double weight = t->kernel[i++];
      int px = x + dx, py = y + dy;
      double* p = &t->pixels[((size_t)py * t->width + px) * 3];
      p[0] = fma(weight, color.x, p[0]);
      p[1] = fma(weight, color.y, p[1]);
      p[2] = fma(weight, color.z, p[2]);
    }
  }
}
/* Film_tile.write_sample, film_tile.ml:40-45 (Float.to_int truncates) */
static void film_tile_write_sample(film_tile_t* t, double x, double y, v3 color) {
  film_tile_write_pixel(t, (int)x, (int)y, color);
}

/* KAT helper for the reference's Film_tile tests (path_tracer_test.ml:72-119):
 * creates a film tile, write_pixel (x,y) white, returns width*height*3 pixels and the
 * global coordinate of local (0,0). */
ORC_API int orc_film_tile_kat(int row, int col, int width, int height, int pixel_radius, int wx, int wy,
                              double* pixels_out, int* dims_out /* w,h,gx0,gy0 */) {
  double kern[64 * 64];
  orc_filter_binomial(5, pixel_radius, kern, NULL);
  tile_t t = {row, col, width, height};
  film_tile_t ft = film_tile_create(t, kern, pixel_radius);
  film_tile_write_pixel(&ft, wx, wy, v3_make(1.0, 1.0, 1.0));
  memcpy(pixels_out, ft.pixels, sizeof(double) * (size_t)ft.width * ft.height * 3);
  dims_out[0] = ft.width; dims_out[1] = ft.height;
  dims_out[2] = 0 + t.col - ft.border; /* Film_tile.iter, film_tile.ml:47-61 */
  dims_out[3] = 0 + t.row - ft.border;
  free(ft.pixels);
  return 0;
}

/* ------------------------------------------------------------------ Bbox (bbox.ml) */
typedef struct { v3 min, max; } bbox_t;
static inline bbox_t bbox_union(bbox_t a, bbox_t b) { /* bbox.ml:14-18 */
  bbox_t r;
  r.min = v3_make(pt_base_min(a.min.x, b.min.x), pt_base_min(a.min.y, b.min.y), pt_base_min(a.min.z, b.min.z));
  r.max = v3_make(pt_base_max(a.max.x, b.max.x), pt_base_max(a.max.y, b.max.y), pt_base_max(a.max.z, b.max.z));
  return r;
}
static inline v3 bbox_center(bbox_t b) { return v3_scale(v3_add(b.min, b.max), 0.5); } /* bbox.ml:12 */
static inline double bbox_surface_area(bbox_t b) {                                      /* bbox.ml:33-38 */
  v3 d = v3_sub(b.max, b.min);
  double a = fma(d.x, d.y, fma(d.y, d.z, d.z * d.x));
  return 2.0 * a;
}
/* Bbox.hit_range / is_hit, bbox.ml:40-56 */
static inline int bbox_is_hit(const bbox_t* t, const ray_t* ray, double t_min, double t_max) {
  v3 invd = ray->direction_inv;
  v3 o = ray->origin;
  v3 t0 = v3_mul(v3_sub(t->min, o), invd);
  v3 t1 = v3_mul(v3_sub(t->max, o), invd);
  v3 mn = v3_make(pt_base_min(t0.x, t1.x), pt_base_min(t0.y, t1.y), pt_base_min(t0.z, t1.z));
  v3 mx = v3_make(pt_base_max(t0.x, t1.x), pt_base_max(t0.y, t1.y), pt_base_max(t0.z, t1.z));
  double a = v3_max_coord(mn);
  double b = v3_min_coord(mx);
  double lo = pt_base_max(t_min, a);
  double hi = pt_base_min(t_max, b);
  return lo <= hi;
}
static inline int bbox_mem(const bbox_t* t, v3 p) { /* bbox.ml:58-64 */
  return t->min.x <= p.x && p.x <= t->max.x && t->min.y <= p.y && p.y <= t->max.y && t->min.z <= p.z && p.z <= t->max.z;
}
ORC_API int orc_bbox_is_hit(const double* bb /*6*/, const double* o, const double* d, double t_min, double t_max) {
  bbox_t b = {v3_make(bb[0], bb[1], bb[2]), v3_make(bb[3], bb[4], bb[5])};
  ray_t r = ray_create(v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
  return bbox_is_hit(&b, &r, t_min, t_max);
}
ORC_API int orc_bbox_mem(const double* bb, const double* p) {
  bbox_t b = {v3_make(bb[0], bb[1], bb[2]), v3_make(bb[3], bb[4], bb[5])};
  return bbox_mem(&b, v3_make(p[0], p[1], p[2]));
}

/* ------------------------------------------------------------------ Quaternion (quaternion.ml) */
typedef struct { double r; v3 v; } quat_t;
/* Quaternion.normalize, quaternion.ml:11-15 */
static inline quat_t quat_normalize(quat_t q) {
  double s = 1.0 / m_hypot(m_hypot(q.r, q.v.x), m_hypot(q.v.y, q.v.z));
  quat_t o;
  o.r = q.r * s;
  o.v = v3_scale(q.v, s);
  return o;
}
/* Quaternion.mul, quaternion.ml:25-32 */
static inline quat_t quat_mul(quat_t a, quat_t b) {
  quat_t o;
  o.r = (a.r * b.r) - v3_dot(a.v, b.v);
  o.v = v3_add(v3_add(v3_cross(a.v, b.v), v3_scale(b.v, a.r)), v3_scale(a.v, b.r));
  return o;
}
static inline quat_t quat_conj(quat_t q) { quat_t o = {q.r, v3_neg(q.v)}; return o; } /* :34-37 */
/* Quaternion.transform, quaternion.ml:39-42 */
static inline v3 quat_transform(quat_t t, v3 v) {
  quat_t p = {0.0, v};
  quat_t q = quat_mul(quat_mul(t, p), quat_conj(t));
  return q.v;
}

/* ------------------------------------------------------------------ Shader_space (shader_space.ml) */
typedef struct { quat_t rotation; v3 origin, normal; } sspace_t;
/* Shader_space.create, shader_space.ml:11-23 */
static inline sspace_t sspace_create(v3 normal, v3 origin) {
  const double epsilon = 1e-9;
  sspace_t s;
  double x = normal.x, y = normal.y, z = normal.z;
  if (z > 1.0 - epsilon) {
    s.rotation.r = 1.0; s.rotation.v = v3_make(0.0, 0.0, 0.0);
  } else if (z < epsilon - 1.0) {
    s.rotation.r = 0.0; s.rotation.v = v3_make(0.0, 1.0, 0.0);
  } else {
    quat_t q = {1.0 + z, v3_make(y, -x, 0.0)};
    s.rotation = quat_normalize(q);
  }
  s.origin = origin;
  s.normal = normal;
  return s;
}
static inline v3 sspace_rotate(const sspace_t* t, v3 v) { return quat_transform(t->rotation, v); }                  /* :27 */
static inline v3 sspace_rotate_inv(const sspace_t* t, v3 v) { return quat_transform(quat_conj(t->rotation), v); }  /* :29-32 */
static inline v3 sspace_reflect(v3 v) { return v3_make(-v.x, -v.y, v.z); }                                          /* :34-39 */
/* Shader_space.refract, :41-49 */
static inline v3 sspace_refract(v3 wi, double index) {
  double c = pt_base_min(wi.z, 1.0);
  v3 perp = v3_scale(v3_sub(v3_make(0.0, 0.0, c), wi), index);
  v3 para = v3_make(0.0, 0.0, -sqrt(fabs(1.0 - v3_quadrance(perp))));
  return v3_add(perp, para);
}
/* Shader_space.world_ray, :51-54 */
static inline ray_t sspace_world_ray(const sspace_t* t, v3 dir_ss) {
  v3 dir = sspace_rotate_inv(t, dir_ss);
  return ray_create(v3_add(t->origin, v3_scale(dir, 1e-3)), dir);
}
/* Shader_space.unit_square_to_hemisphere, :56-64 */
static inline v3 unit_square_to_hemisphere(double u, double v) {
  const double pi = 3.14159265358979323846; /* Float.pi */
  double r = sqrt(u);
  double theta = v * 2.0 * pi;
  double x = r * m_cos(theta);
  double y = r * m_sin(theta);
  double z = sqrt(1.0 - u);
  return v3_make(x, y, z);
}
ORC_API void orc_unit_square_to_hemisphere(double u, double v, double* out) {
  v3 w = unit_square_to_hemisphere(u, v);
  out[0] = w.x; out[1] = w.y; out[2] = w.z;
}
/* Shader_space.omega_i, :66-69 */
static inline v3 sspace_omega_i(const sspace_t* t, const ray_t* ray) { return sspace_rotate(t, v3_neg(ray->direction)); }

/* ------------------------------------------------------------------ Texture / Material / Scatter */
typedef struct { double u, v; } texcoord_t;

typedef struct {
  int n_materials, n_textures;
  ptx_material* materials;
  ptx_texture* textures;
} mattable_t;

/* Texture.eval / solid / checker, texture.ml:16-31 */
static inline v3 texture_eval(const mattable_t* mt, int tex, texcoord_t coord) {
  const ptx_texture* t = &mt->textures[tex];
  if (t->kind == PTX_TEX_SOLID) return v3_make(t->even[0], t->even[1], t->even[2]);
  double width = (double)(t->width - 1), height = (double)(t->height - 1);
  double xp = coord.u * width, yp = coord.v * height;
  int64_t px = ((int64_t)xp) & 1, py = ((int64_t)yp) & 1; /* Float.to_int a land 1 */
  if (px == py) return v3_make(t->even[0], t->even[1], t->even[2]);
  return v3_make(t->odd[0], t->odd[1], t->odd[2]);
}

enum { SC_ABSORB = 0, SC_SPECULAR = 1, SC_DIFFUSE = 2 };
typedef struct { int kind; ray_t ray; v3 attenuation; } scatter_t;

/* Hit.t, hit.ml:3-7: do_scatter is a closure over the values captured below */
typedef struct {
  sspace_t shader_space;
  v3 emit;
  /* captured by Material.scatter's partial application, material.ml:22-57 */
  const ptx_material* m;
  texcoord_t tex_coord;
  v3 omega_i;
  int hit_front;
} hit_t;

/* schlick_reflectance, material.ml:16-20 */
static inline double schlick_reflectance(double cos_theta, double index) {
  double q = (1.0 - index) / (1.0 + index);
  double r0 = q * q; /* Float.square */
  return r0 + ((1.0 - r0) * m_pow5(1.0 - cos_theta));
}
/* Base Float.clamp_exn, NaN-propagating form */
static inline double base_clamp(double t, double mn, double mx) { return t < mn ? mn : (mx < t ? mx : t); }

/* Material.scatter applied to u, material.ml:22-57 */
static scatter_t hit_scatter(const mattable_t* mt, const hit_t* h, double u) {
  scatter_t s;
  memset(&s, 0, sizeof s);
  const ptx_material* m = h->m;
  const sspace_t* ss = &h->shader_space;
  if (m->kind == PTX_MAT_LAMBERTIAN) {
    s.kind = SC_DIFFUSE;
    s.attenuation = texture_eval(mt, m->texture, h->tex_coord);
    return s;
  }
  if (m->kind == PTX_MAT_METAL) {
    v3 omega_r = sspace_reflect(h->omega_i);
    double z = omega_r.z;
    if (z <= 0.0) { s.kind = SC_ABSORB; return s; }
    v3 a = texture_eval(mt, m->texture, h->tex_coord);
    double sp = m_pow5(1.0 - h->omega_i.z);
    v3 c = v3_scale(v3_sub(v3_make(1.0, 1.0, 1.0), a), sp);
    s.kind = SC_SPECULAR;
    s.attenuation = v3_add(a, c);
    s.ray = sspace_world_ray(ss, omega_r);
    return s;
  }
  /* Dielectric */
  double index = m->index, index_inv = 1.0 / m->index; /* material.ml:13 */
  double wi_z = h->omega_i.z;
  double c = base_clamp(wi_z, 0.0, 1.0);
  double sn = sqrt(1.0 - c * c);
  double refract_ratio = h->hit_front ? index_inv : index;
  v3 wo;
  if (refract_ratio * sn > 1.0 || schlick_reflectance(c, refract_ratio) > u) wo = sspace_reflect(h->omega_i);
  else wo = sspace_refract(h->omega_i, refract_ratio);
  s.kind = SC_SPECULAR;
  s.ray = sspace_world_ray(ss, wo);
  s.attenuation = v3_make(1.0, 1.0, 1.0);
  return s;
}

/* ------------------------------------------------------------------ primitives */
enum { PRIM_SPHERE = 0, PRIM_TRIANGLE = 1 };
typedef struct {
  int kind;
  int material;
  int id; /* index in the build list: [triangles] @ [spheres] */
  /* sphere */
  v3 center; double radius;
  /* triangle */
  v3 a, b, c; texcoord_t ta, tb, tc;
} prim_t;

/* Sphere.bbox, sphere.ml:16-19 */
static inline bbox_t sphere_bbox(const prim_t* s) {
  v3 r = v3_make(s->radius, s->radius, s->radius);
  bbox_t b; b.min = v3_add(s->center, v3_neg(r)); b.max = v3_add(s->center, r);
  return b;
}
/* Sphere.intersect (scalar, --no-simd path), sphere.ml:35-54 */
static inline int sphere_intersect(const prim_t* s, const ray_t* ray, double t_min, double t_max, double* t_out) {
  v3 d = ray->direction;
  double r2 = s->radius * s->radius;
  v3 f = v3_sub(s->center, ray->origin); /* of_points ~src:origin ~tgt:center */
  double bp = v3_dot(f, d);
  double a = v3_quadrance(d);
  double discrim = r2 - v3_quadrance(v3_sub(v3_scale(d, bp / a), f));
  if (discrim < 0.0) return 0;
  double sign_bp = (bp >= 0.0) ? 1.0 : -1.0;
  double q = fma(sign_bp, sqrt(a * discrim), bp);
  double c = v3_quadrance(f) - r2;
  double t_hit = (c > 0.0) ? c / q : q / a;
  if (t_min <= t_hit && t_hit <= t_max) { *t_out = t_hit; return 1; }
  return 0;
}
ORC_API int orc_sphere_intersect(const double* center, double radius, const double* o, const double* d, double t_min, double t_max, double* t_out) {
  prim_t s; memset(&s, 0, sizeof s);
  s.center = v3_make(center[0], center[1], center[2]); s.radius = radius;
  ray_t r = ray_create(v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
  return sphere_intersect(&s, &r, t_min, t_max, t_out);
}

/* spheres_intersect_aux, x86 AVX2+FMA body, sphere-intersect-rs/src/lib.rs:102-178.
 * xs..rs: SoA packet, len a multiple of 4 (NaN padded, main.ml:177-193), <= 16 lanes used.
 * The AVX lanes are independent, so a scalar loop with the same per-lane operations is
 * bit-identical.  Returns index (or -1), t in *t_found. */
static inline int spheres_intersect_packet(const double* xs, const double* ys, const double* zs, const double* rs, int len,
                                           v3 o, v3 d, double t_min, double t_max, double* t_found_out) {
  double t_hits[16];
  for (int i = 0; i < 16; ++i) t_hits[i] = 0.0; /* lib.rs:114 */
  /* V3::dot is UNFUSED scalar, lib.rs:38-40: x*x + y*y + z*z, left-assoc */
  double d_quadrance = d.x * d.x + d.y * d.y + d.z * d.z;
  double a = d_quadrance;
  double one_over_a = 1.0 / a;
  int chunks = len / 4; if (chunks > 4) chunks = 4; /* zip with t_hits.chunks_exact_mut(4) */
  for (int i = 0; i < chunks * 4; ++i) {
    double fx = xs[i] - o.x, fy = ys[i] - o.y, fz = zs[i] - o.z;
    double r2 = rs[i] * rs[i];
    double c = fma(fx, fx, fma(fy, fy, fz * fz)) - r2;          /* lib.rs:140 */
    double bp = fma(fx, d.x, fma(fy, d.y, fz * d.z));           /* :142 */
    double bp_over_a = bp * one_over_a;                          /* :143 */
    double wx = fma(d.x, bp_over_a, -fx);                        /* fmsub :145-147 */
    double wy = fma(d.y, bp_over_a, -fy);
    double wz = fma(d.z, bp_over_a, -fz);
    double wq = fma(wx, wx, fma(wy, wy, wz * wz));
    double discriminant = r2 - wq;
    double q_rhs = sqrt(a * discriminant);                       /* :152 */
    double q = pt_signbit(bp) ? (bp - q_rhs) : (bp + q_rhs);     /* blendv on sign(bp) :153 */
    double c_div_q = c / q;
    double q_div_a = q * one_over_a;
    double t_hit = pt_signbit(c) ? q_div_a : c_div_q;            /* blendv on sign(c) :157 */
    int outside = (t_hit < t_min) || (t_hit > t_max);            /* _CMP_LT_OQ / _CMP_GT_OQ: false on NaN */
    /* blendv on the sign bit of (discriminant | outside_range), :162-166 */
    if (pt_signbit(discriminant) || outside) t_hit = NAN;
    t_hits[i] = t_hit;
  }
  double t_found = t_max;
  int found = -1;
  int n = len < 16 ? len : 16; /* .take(xs.len()) over a 16-array */
  for (int i = 0; i < n; ++i) {
    if (t_hits[i] <= t_found) { t_found = t_hits[i]; found = i; }
  }
  *t_found_out = t_found;
  return found;
}
ORC_API int orc_spheres_intersect_packet(const double* xs, const double* ys, const double* zs, const double* rs, int len,
                                         const double* o, const double* d, double t_min, double t_max, double* t_out) {
  return spheres_intersect_packet(xs, ys, zs, rs, len, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]), t_min, t_max, t_out);
}

/* Triangle.bbox, triangle.ml:67-72 */
static inline bbox_t triangle_bbox(const prim_t* t) {
  bbox_t r;
  v3 lo = v3_make(pt_base_min(t->a.x, t->b.x), pt_base_min(t->a.y, t->b.y), pt_base_min(t->a.z, t->b.z));
  v3 hi = v3_make(pt_base_max(t->a.x, t->b.x), pt_base_max(t->a.y, t->b.y), pt_base_max(t->a.z, t->b.z));
  r.min = v3_make(pt_base_min(lo.x, t->c.x), pt_base_min(lo.y, t->c.y), pt_base_min(lo.z, t->c.z));
  r.max = v3_make(pt_base_max(hi.x, t->c.x), pt_base_max(hi.y, t->c.y), pt_base_max(hi.z, t->c.z));
  return r;
}
typedef struct { double t_hit, u, v; } trihit_t;
/* Triangle.intersect, triangle.ml:74-98 */
static inline int triangle_intersect(const prim_t* t, const ray_t* r, double t_min, double t_max, trihit_t* out) {
  const double epsilon = 1e-6;
  v3 e1 = v3_sub(t->b, t->a);
  v3 e2 = v3_sub(t->c, t->a);
  v3 dir = r->direction;
  v3 pvec = v3_cross(dir, e2);
  double det = v3_dot(e1, pvec);
  if (fabs(det) < epsilon) return 0;
  double det_inv = 1.0 / det;
  v3 tvec = v3_sub(r->origin, t->a);
  double u = det_inv * v3_dot(tvec, pvec);
  v3 qvec = v3_cross(tvec, e1);
  double v = det_inv * v3_dot(dir, qvec);
  if (0.0 <= u && u <= 1.0 && 0.0 <= v && u + v <= 1.0) {
    double t_hit = det_inv * v3_dot(e2, qvec);
    if (t_min <= t_hit && t_hit <= t_max) { out->t_hit = t_hit; out->u = u; out->v = v; return 1; }
  }
  return 0;
}
ORC_API int orc_triangle_intersect(const double* abc /*9*/, const double* o, const double* d, double t_min, double t_max, double* tuv_out) {
  prim_t t; memset(&t, 0, sizeof t);
  t.a = v3_make(abc[0], abc[1], abc[2]); t.b = v3_make(abc[3], abc[4], abc[5]); t.c = v3_make(abc[6], abc[7], abc[8]);
  ray_t r = ray_create(v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
  trihit_t h;
  int ok = triangle_intersect(&t, &r, t_min, t_max, &h);
  if (ok) { tuv_out[0] = h.t_hit; tuv_out[1] = h.u; tuv_out[2] = h.v; }
  return ok;
}

/* Sphere.hit, sphere.ml:56-69 (+ normal :21, tex_coord :25-33) */
static hit_t sphere_hit(const mattable_t* mt, const prim_t* s, double t_hit, const ray_t* ray) {
  const double pi = 3.14159265358979323846;
  const double one_over_pi = 1.0 / pi, one_over_two_pi = 1.0 / (2.0 * pi);
  hit_t h;
  v3 point = ray_point_at(ray, t_hit);
  v3 normal = v3_normalize(v3_sub(point, s->center));
  int hit_front = v3_dot(ray->direction, normal) < 0.0;
  if (!hit_front) normal = v3_neg(normal);
  double theta = m_acos(-normal.y);
  double phi = pi + m_atan2(-normal.z, normal.x);
  h.tex_coord.u = phi * one_over_two_pi;
  h.tex_coord.v = theta * one_over_pi;
  h.shader_space = sspace_create(normal, point);
  h.m = &mt->materials[s->material];
  h.emit = v3_make(h.m->emit[0], h.m->emit[1], h.m->emit[2]); /* Material.emit = black in the reference, material.ml:59 */
  h.omega_i = sspace_omega_i(&h.shader_space, ray);
  h.hit_front = hit_front;
  return h;
}
/* Triangle.Hit.to_hit, triangle.ml:43-64 (+ g_normal :24-29, point :31-38) */
static hit_t triangle_hit(const mattable_t* mt, const prim_t* t, const trihit_t* th, const ray_t* r) {
  hit_t h;
  v3 e1 = v3_sub(t->b, t->a), e2 = v3_sub(t->c, t->a);
  v3 g_normal = v3_normalize(v3_cross(e1, e2));
  double u = th->u, v = th->v;
  double w = 1.0 - u - v;
  v3 pt = v3_add(v3_add(v3_scale(t->a, w), v3_scale(t->b, u)), v3_scale(t->c, v));
  double w2 = 1.0 - u - v;
  h.tex_coord.u = (t->ta.u * w2) + (t->tb.u * u) + (t->tc.u * v);
  h.tex_coord.v = (t->ta.v * w2) + (t->tb.v * u) + (t->tc.v * v);
  int hit_front = v3_dot(r->direction, g_normal) < 0.0;
  v3 normal = hit_front ? g_normal : v3_neg(g_normal);
  h.shader_space = sspace_create(normal, pt);
  h.omega_i = sspace_omega_i(&h.shader_space, r);
  h.m = &mt->materials[t->material];
  /* triangle.ml:63 hard-codes emit = black; the emitter extension reads the material slot */
  h.emit = v3_make(h.m->emit[0], h.m->emit[1], h.m->emit[2]);
  h.hit_front = hit_front;
  return h;
}

/* ------------------------------------------------------------------ Shape_tree (shape_tree.ml) */
typedef struct { prim_t shape; bbox_t bbox; v3 centroid; } bshape_t; /* Bshape.t, :4-19 */

typedef struct node_s {
  bbox_t bbox;
  int is_leaf;
  /* branch */
  int axis;
  struct node_s *lhs, *rhs;
  /* leaf */
  int n_elts;   /* real elements */
  int length;   /* Leaf.length: padded length for Simd_leaf, n for Array_leaf */
  prim_t* elts; /* n_elts */
  double *xs, *ys, *zs, *rs; /* Simd_leaf coords, length entries (NaN padded) */
} node_t;

typedef struct {
  int64_t segments, nodes_tested, prims_tested, floor_tested, samples;
} counters_t;

typedef struct orc_scene {
  mattable_t mt;
  int n_prims;
  prim_t* prims; /* build list order */
  int n_floor;
  prim_t* floor;
  ptx_camera camera;
  ptx_background background;
  int leaf_kind, length_cutoff, num_bins;
  node_t* root;
  int tree_nodes, tree_leaves, tree_depth, leaf_slots;
  double build_ms;
} orc_scene;

static bbox_t prim_bbox(const prim_t* p) { return p->kind == PRIM_SPHERE ? sphere_bbox(p) : triangle_bbox(p); }

typedef struct { double cost; int split_index; int axis; bbox_t lhs_box, rhs_box; double scale, cb_min; int valid; } proposal_t;

/* OCaml polymorphic/Float compare: NaN equal to itself and less than any other float */
static int ocaml_float_compare(double a, double b) {
  if (a != a) return (b != b) ? 0 : -1;
  if (b != b) return 1;
  return a < b ? -1 : (a > b ? 1 : 0);
}

/* Proposal.propose_split_one_axis, shape_tree.ml:123-139, with Bin.* :27-70 and candidates :91-119 */
static proposal_t propose_split_one_axis(int num_bins, bshape_t* shapes, int n, int axis, bbox_t cbbox) {
  proposal_t best; memset(&best, 0, sizeof best); best.valid = 0;
  const double epsilon = 1e-6;
  double cb_min = v3_axis(cbbox.min, axis), cb_max = v3_axis(cbbox.max, axis);
  double scale = (double)num_bins * (1.0 - epsilon) / (cb_max - cb_min);
  if (!pt_isfinite(scale)) return best;
  int* count = (int*)calloc((size_t)num_bins, sizeof(int));
  char* has = (char*)calloc((size_t)num_bins * 3, 1); /* bounds / bbox_l / bbox_r present */
  bbox_t* bounds = (bbox_t*)malloc(sizeof(bbox_t) * (size_t)num_bins * 3);
  bbox_t* bl = bounds + num_bins; bbox_t* br = bounds + 2 * num_bins;
  char *hb = has, *hl = has + num_bins, *hr = has + 2 * num_bins;
  for (int i = 0; i < n; ++i) { /* Slice.iter shapes ~f:(Bin.insert bins.(to_bin s)) */
    int b = (int)(scale * (v3_axis(shapes[i].centroid, axis) - cb_min));
    if (hb[b]) bounds[b] = bbox_union(bounds[b], shapes[i].bbox); else { bounds[b] = shapes[i].bbox; hb[b] = 1; }
    count[b]++;
  }
  /* populate_bbox_r, :53-60 : union_opt (bbox bin_j) (bbox_r bins.(j+1)) */
  hr[num_bins - 1] = hb[num_bins - 1]; br[num_bins - 1] = bounds[num_bins - 1];
  for (int j = num_bins - 2; j >= 0; --j) {
    if (hb[j] && hr[j + 1]) { br[j] = bbox_union(bounds[j], br[j + 1]); hr[j] = 1; }
    else if (hb[j]) { br[j] = bounds[j]; hr[j] = 1; }
    else if (hr[j + 1]) { br[j] = br[j + 1]; hr[j] = 1; }
    else hr[j] = 0;
  }
  /* populate_bbox_l, :62-69 */
  hl[0] = hb[0]; bl[0] = bounds[0];
  for (int j = 1; j < num_bins; ++j) {
    if (hb[j] && hl[j - 1]) { bl[j] = bbox_union(bounds[j], bl[j - 1]); hl[j] = 1; }
    else if (hb[j]) { bl[j] = bounds[j]; hl[j] = 1; }
    else if (hl[j - 1]) { bl[j] = bl[j - 1]; hl[j] = 1; }
    else hl[j] = 0;
  }
  /* candidates, :91-119.  The list is consed (highest p first) and List.min_elt keeps the FIRST
   * minimum, so among equal costs the highest p wins: iterate p descending, replace on strict <. */
  double total_area = bbox_surface_area(bl[num_bins - 1]);
  int total_count = 0;
  for (int j = 0; j < num_bins; ++j) total_count += count[j];
  int* n_left_at = (int*)malloc(sizeof(int) * (size_t)num_bins);
  int acc = 0;
  for (int p = 0; p < num_bins - 1; ++p) { acc += count[p]; n_left_at[p] = acc; }
  for (int p = num_bins - 2; p >= 0; --p) {
    if (!hl[p] || !hr[p + 1]) continue;
    int lhs_count = n_left_at[p];
    int rhs_count = total_count - lhs_count;
    double lhs_area = (double)lhs_count * bbox_surface_area(bl[p]);
    double rhs_area = (double)rhs_count * bbox_surface_area(br[p + 1]);
    double cost = 0.25 + ((lhs_area + rhs_area) * 1.0 / total_area); /* costT + ((l + r) * costI / total) */
    if (!best.valid || ocaml_float_compare(best.cost, cost) > 0) {
      best.valid = 1; best.cost = cost; best.split_index = p; best.axis = axis;
      best.lhs_box = bl[p]; best.rhs_box = br[p + 1]; best.scale = scale; best.cb_min = cb_min;
    }
  }
  free(n_left_at); free(count); free(has); free(bounds);
  return best;
}

/* Proposal.create, shape_tree.ml:141-146: axes X,Y,Z; List.min_elt keeps the first minimum */
static proposal_t proposal_create(int num_bins, bshape_t* shapes, int n) {
  bbox_t cbbox; cbbox.min = shapes[0].centroid; cbbox.max = shapes[0].centroid; /* Bshape.centroid_bbox, :21-24 */
  for (int i = 1; i < n; ++i) { bbox_t c = {shapes[i].centroid, shapes[i].centroid}; cbbox = bbox_union(cbbox, c); }
  proposal_t best; memset(&best, 0, sizeof best);
  for (int axis = 0; axis < 3; ++axis) {
    proposal_t p = propose_split_one_axis(num_bins, shapes, n, axis, cbbox);
    if (!p.valid) continue;
    if (!best.valid || ocaml_float_compare(best.cost, p.cost) > 0) best = p;
  }
  return best;
}

static inline int on_lhs(const proposal_t* p, const bshape_t* b) { /* to_bin b <= p, shape_tree.ml:113,133 */
  int bin = (int)(p->scale * (v3_axis(b->centroid, p->axis) - p->cb_min));
  return bin <= p->split_index;
}

/* Slice.partition_in_place, slice.ml:67-80; returns the split index i */
static int partition_in_place(bshape_t* t, int length, const proposal_t* p) {
  int i = 0, j = length - 1;
  while (i < j) {
    while (on_lhs(p, &t[i]) && i < j) ++i;
    while (j >= 0 && !on_lhs(p, &t[j])) --j; /* (the reference evaluates get before the j>=0 guard; never reached with j<0) */
    if (i < j) { bshape_t tmp = t[i]; t[i] = t[j]; t[j] = tmp; }
  }
  return i;
}

static node_t* make_leaf(orc_scene* sc, bbox_t bbox, bshape_t* shapes, int n) { /* Tree.make_leaf :173-175 */
  node_t* nd = (node_t*)calloc(1, sizeof(node_t));
  nd->bbox = bbox; nd->is_leaf = 1; nd->axis = -1; nd->n_elts = n;
  nd->elts = (prim_t*)malloc(sizeof(prim_t) * (size_t)n);
  for (int i = 0; i < n; ++i) nd->elts[i] = shapes[i].shape;
  if (sc->leaf_kind == PTX_LEAF_SIMD) { /* Simd_leaf.of_elts, shirley_spheres/bin/main.ml:177-193 */
    int rem = n % 4, pad = rem == 0 ? 0 : 4 - rem, len = n + pad;
    nd->length = len;
    nd->xs = (double*)malloc(sizeof(double) * (size_t)len * 4);
    nd->ys = nd->xs + len; nd->zs = nd->ys + len; nd->rs = nd->zs + len;
    for (int i = 0; i < len; ++i) {
      if (i < n) { nd->xs[i] = shapes[i].shape.center.x; nd->ys[i] = shapes[i].shape.center.y; nd->zs[i] = shapes[i].shape.center.z; nd->rs[i] = shapes[i].shape.radius; }
      else { nd->xs[i] = nd->ys[i] = nd->zs[i] = nd->rs[i] = NAN; }
    }
  } else {
    nd->length = n;
  }
  sc->tree_nodes++; sc->tree_leaves++; sc->leaf_slots += nd->length;
  return nd;
}

/* Tree.create loop, shape_tree.ml:177-196 */
static node_t* tree_build(orc_scene* sc, bbox_t bbox, bshape_t* shapes, int n) {
  proposal_t p = proposal_create(sc->num_bins, shapes, n);
  if (!p.valid) return make_leaf(sc, bbox, shapes, n);
  double leaf_cost = 1.0 * (double)n; /* Proposal.leaf_cost :84 */
  if ((p.cost >= leaf_cost && n <= sc->length_cutoff) || n <= 4) return make_leaf(sc, bbox, shapes, n);
  int i = partition_in_place(shapes, n, &p);
  node_t* nd = (node_t*)calloc(1, sizeof(node_t));
  nd->bbox = bbox; nd->is_leaf = 0; nd->axis = p.axis;
  sc->tree_nodes++;
  nd->lhs = tree_build(sc, p.lhs_box, shapes, i);
  nd->rhs = tree_build(sc, p.rhs_box, shapes + i, n - i);
  return nd;
}
static int tree_depth(const node_t* n) { /* depth = cata (1 + max l r) ~leaf:L.depth (= 0) :239 */
  if (n->is_leaf) return 0;
  int l = tree_depth(n->lhs), r = tree_depth(n->rhs);
  return 1 + (l > r ? l : r);
}
static void tree_free(node_t* n) {
  if (!n) return;
  if (n->is_leaf) { free(n->elts); free(n->xs); }
  else { tree_free(n->lhs); tree_free(n->rhs); }
  free(n);
}

typedef struct { double t_hit; int found; const prim_t* prim; prim_t simd_sphere; trihit_t tri; } elthit_t;

/* Leaf.intersect: Simd_leaf.intersect (main.ml:206-217) or Array_leaf.intersect (shape_tree.ml:299-311) */
static int leaf_intersect(const orc_scene* sc, const node_t* l, const ray_t* ray, double t_min, double t_max, elthit_t* out, counters_t* ct) {
  if (sc->leaf_kind == PTX_LEAF_SIMD) {
    double t_found;
    if (ct) ct->prims_tested += l->length;
    int idx = spheres_intersect_packet(l->xs, l->ys, l->zs, l->rs, l->length, ray->origin, ray->direction, t_min, t_max, &t_found);
    if (idx < 0) return 0;
    out->t_hit = t_found; out->found = 1;
    /* rebuilds a Sphere.t from the coords, main.ml:213-215 */
    out->simd_sphere = l->elts[idx];
    out->simd_sphere.center = v3_make(l->xs[idx], l->ys[idx], l->zs[idx]);
    out->simd_sphere.radius = l->rs[idx];
    out->prim = &out->simd_sphere;
    return 1;
  }
  int any = 0;
  double tm = t_max;
  for (int i = 0; i < l->n_elts; ++i) {
    const prim_t* s = &l->elts[i];
    if (ct) ct->prims_tested += 1;
    if (s->kind == PRIM_SPHERE) {
      double t;
      if (sphere_intersect(s, ray, t_min, tm, &t)) { any = 1; out->t_hit = t; out->prim = s; tm = t; }
    } else {
      trihit_t th;
      if (triangle_intersect(s, ray, t_min, tm, &th)) { any = 1; out->t_hit = th.t_hit; out->prim = s; out->tri = th; tm = th.t_hit; }
    }
  }
  out->found = any;
  return any;
}

/* Tree.intersect, shape_tree.ml:198-220 (recursive, ordered, far child searched with t_max = near hit) */
static int tree_intersect_rec(const orc_scene* sc, const node_t* t, const ray_t* ray, const int dirs[3], double t_min, double t_max, elthit_t* out, counters_t* ct) {
  if (ct) ct->nodes_tested += 1;
  if (!bbox_is_hit(&t->bbox, ray, t_min, t_max)) return 0;
  if (t->is_leaf) return leaf_intersect(sc, t, ray, t_min, t_max, out, ct);
  const node_t *t1, *t2;
  if (dirs[t->axis]) { t1 = t->lhs; t2 = t->rhs; } else { t1 = t->rhs; t2 = t->lhs; }
  elthit_t h1;
  if (!tree_intersect_rec(sc, t1, ray, dirs, t_min, t_max, &h1, ct)) return tree_intersect_rec(sc, t2, ray, dirs, t_min, t_max, out, ct);
  elthit_t h2;
  if (tree_intersect_rec(sc, t2, ray, dirs, t_min, h1.t_hit, &h2, ct)) { *out = h2; if (h2.prim == &h2.simd_sphere) out->prim = &out->simd_sphere; return 1; }
  *out = h1; if (h1.prim == &h1.simd_sphere) out->prim = &out->simd_sphere;
  return 1;
}
static int tree_intersect(const orc_scene* sc, const ray_t* ray, double t_min, double t_max, elthit_t* out, counters_t* ct) {
  v3 dir = ray->direction;
  int dirs[3] = {dir.x >= 0.0, dir.y >= 0.0, dir.z >= 0.0};
  return tree_intersect_rec(sc, sc->root, ray, dirs, t_min, t_max, out, ct);
}

#define MAX_FINITE 1.7976931348623157e308 /* Float.max_finite_value */

/* Scene.intersect: shirley (main.ml:273-277), cornell (cornell-box/bin/main.ml:230-234),
 * ganesha with the floor tested first (ganesha/bin/main.ml:247-256,286-298).
 * Returns 1 and fills h; prim_id = build-list index (floor: n_prims + i). */
static int scene_intersect(const orc_scene* sc, const ray_t* r, hit_t* h, double* t_out, int* prim_id, counters_t* ct) {
  if (ct) ct->segments += 1;
  elthit_t eh;
  if (sc->n_floor > 0) {
    /* Floor.intersect: f1 then f2, first Some wins */
    trihit_t fh; int which = -1;
    for (int i = 0; i < sc->n_floor; ++i) {
      if (ct) ct->floor_tested += 1;
      if (triangle_intersect(&sc->floor[i], r, 0.0, MAX_FINITE, &fh)) { which = i; break; }
    }
    if (which >= 0) {
      double t_max = fh.t_hit;
      if (sc->root && tree_intersect(sc, r, 0.0, t_max, &eh, ct)) goto tree_hit;
      *h = triangle_hit(&sc->mt, &sc->floor[which], &fh, r);
      if (t_out) *t_out = fh.t_hit;
      if (prim_id) *prim_id = sc->n_prims + which;
      return 1;
    }
  }
  if (!sc->root || !tree_intersect(sc, r, 0.0, MAX_FINITE, &eh, ct)) return 0;
tree_hit:
  if (eh.prim->kind == PRIM_SPHERE) *h = sphere_hit(&sc->mt, eh.prim, eh.t_hit, r);
  else *h = triangle_hit(&sc->mt, eh.prim, &eh.tri, r);
  if (t_out) *t_out = eh.t_hit;
  if (prim_id) *prim_id = eh.prim->id;
  return 1;
}

/* Scene.background: shirley_spheres/bin/main.ml:104-110 */
static v3 scene_background(const orc_scene* sc, const ray_t* ray) {
  if (sc->background.kind == PTX_BG_BLACK) return v3_make(0.0, 0.0, 0.0);
  v3 d = v3_normalize(ray->direction);
  double t = 0.5 * (v3_dot(d, v3_make(0.0, 1.0, 0.0)) + 1.0);
  const double* hz = sc->background.horizon; const double* zn = sc->background.zenith;
  return v3_lerp(t, v3_make(hz[0], hz[1], hz[2]), v3_make(zn[0], zn[1], zn[2]));
}

/* ------------------------------------------------------------------ scene construction from a declarative description */
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

ORC_API orc_scene* orc_scene_create(const ptx_scene_desc* d) {
  orc_scene* sc = (orc_scene*)calloc(1, sizeof(orc_scene));
  sc->mt.n_materials = d->n_materials; sc->mt.n_textures = d->n_textures;
  sc->mt.materials = (ptx_material*)malloc(sizeof(ptx_material) * (size_t)(d->n_materials + 1));
  sc->mt.textures = (ptx_texture*)malloc(sizeof(ptx_texture) * (size_t)(d->n_textures + 1));
  memcpy(sc->mt.materials, d->materials, sizeof(ptx_material) * (size_t)d->n_materials);
  memcpy(sc->mt.textures, d->textures, sizeof(ptx_texture) * (size_t)d->n_textures);
  sc->camera = d->camera; sc->background = d->background;
  sc->leaf_kind = d->leaf_kind; sc->length_cutoff = d->length_cutoff; sc->num_bins = d->num_bins > 0 ? d->num_bins : 32;
  int n = d->n_triangles + d->n_spheres;
  sc->n_prims = n;
  sc->prims = (prim_t*)calloc((size_t)(n + 1), sizeof(prim_t));
  for (int i = 0; i < d->n_triangles; ++i) {
    prim_t* p = &sc->prims[i];
    p->kind = PRIM_TRIANGLE; p->id = i; p->material = d->tri_material[i];
    int ia = d->tri_indices[3 * i], ib = d->tri_indices[3 * i + 1], ic = d->tri_indices[3 * i + 2];
    p->a = v3_make(d->vertex_x[ia], d->vertex_y[ia], d->vertex_z[ia]);
    p->b = v3_make(d->vertex_x[ib], d->vertex_y[ib], d->vertex_z[ib]);
    p->c = v3_make(d->vertex_x[ic], d->vertex_y[ic], d->vertex_z[ic]);
    const double* uv = &d->tri_uv[6 * i];
    p->ta.u = uv[0]; p->ta.v = uv[1]; p->tb.u = uv[2]; p->tb.v = uv[3]; p->tc.u = uv[4]; p->tc.v = uv[5];
  }
  for (int i = 0; i < d->n_spheres; ++i) {
    prim_t* p = &sc->prims[d->n_triangles + i];
    p->kind = PRIM_SPHERE; p->id = d->n_triangles + i; p->material = d->sphere_material[i];
    p->center = v3_make(d->sphere_x[i], d->sphere_y[i], d->sphere_z[i]); p->radius = d->sphere_r[i];
  }
  sc->n_floor = d->n_floor_triangles;
  sc->floor = (prim_t*)calloc((size_t)(sc->n_floor + 1), sizeof(prim_t));
  for (int i = 0; i < sc->n_floor; ++i) {
    prim_t* p = &sc->floor[i];
    const double* v = &d->floor_vertices[9 * i]; const double* uv = &d->floor_uv[6 * i];
    p->kind = PRIM_TRIANGLE; p->id = n + i; p->material = d->floor_material[i];
    p->a = v3_make(v[0], v[1], v[2]); p->b = v3_make(v[3], v[4], v[5]); p->c = v3_make(v[6], v[7], v[8]);
    p->ta.u = uv[0]; p->ta.v = uv[1]; p->tb.u = uv[2]; p->tb.v = uv[3]; p->tc.u = uv[4]; p->tc.v = uv[5];
  }
  if (n > 0) {
    /* Shape_tree.create, shape_tree.ml:252-263 */
    double t0 = now_ms();
    bshape_t* bs = (bshape_t*)malloc(sizeof(bshape_t) * (size_t)n);
    for (int i = 0; i < n; ++i) { bs[i].shape = sc->prims[i]; bs[i].bbox = prim_bbox(&sc->prims[i]); bs[i].centroid = bbox_center(bs[i].bbox); }
    bbox_t bbox = bs[0].bbox;
    for (int i = 1; i < n; ++i) bbox = bbox_union(bbox, bs[i].bbox);
    sc->root = tree_build(sc, bbox, bs, n);
    sc->tree_depth = tree_depth(sc->root);
    free(bs);
    sc->build_ms = now_ms() - t0;
  }
  return sc;
}
ORC_API void orc_scene_destroy(orc_scene* sc) {
  if (!sc) return;
  tree_free(sc->root); free(sc->prims); free(sc->floor); free(sc->mt.materials); free(sc->mt.textures); free(sc);
}
ORC_API void orc_scene_info(const orc_scene* sc, int* out /* nodes, leaves, depth, slots, n_prims */, double* build_ms) {
  out[0] = sc->tree_nodes; out[1] = sc->tree_leaves; out[2] = sc->tree_depth; out[3] = sc->leaf_slots; out[4] = sc->n_prims;
  if (build_ms) *build_ms = sc->build_ms;
}

/* pre-order flattening with the same record shape as ptx_scene_tree (include/ptx.h) */
static void flatten_rec(const node_t* n, double* bbox_out, int* info_out, int* prim_order, int* n_nodes, int* n_slots) {
  int me = (*n_nodes)++;
  if (bbox_out) { double* b = &bbox_out[6 * me]; b[0] = n->bbox.min.x; b[1] = n->bbox.min.y; b[2] = n->bbox.min.z; b[3] = n->bbox.max.x; b[4] = n->bbox.max.y; b[5] = n->bbox.max.z; }
  if (n->is_leaf) {
    if (info_out) { info_out[4 * me] = 1; info_out[4 * me + 1] = -1; info_out[4 * me + 2] = *n_slots; info_out[4 * me + 3] = n->length; }
    for (int i = 0; i < n->length; ++i) { if (prim_order) prim_order[*n_slots] = i < n->n_elts ? n->elts[i].id : -1; (*n_slots)++; }
    return;
  }
  int lhs_index = *n_nodes;
  flatten_rec(n->lhs, bbox_out, info_out, prim_order, n_nodes, n_slots);
  int rhs_index = *n_nodes;
  flatten_rec(n->rhs, bbox_out, info_out, prim_order, n_nodes, n_slots);
  if (info_out) { info_out[4 * me] = 0; info_out[4 * me + 1] = n->axis; info_out[4 * me + 2] = lhs_index; info_out[4 * me + 3] = rhs_index; }
}
ORC_API int orc_scene_tree(const orc_scene* sc, double* bbox_out, int* info_out, int* prim_order_out) {
  int n_nodes = 0, n_slots = 0;
  if (sc->root) flatten_rec(sc->root, bbox_out, info_out, prim_order_out, &n_nodes, &n_slots);
  return n_nodes;
}

/* ------------------------------------------------------------------ Camera (camera.ml) */
/* Mat4.dot4, camera.ml:9-12: unfused, left-assoc */
static inline double dot4(const double* a, const double* b) { return (a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2]) + (a[3] * b[3]); }
typedef struct { double look_at[4][4]; ptx_camera cam; } camera_t;
/* Camera.create, camera.ml:58-83 + Mat4.look_at :14-27 (translate/rotate fields are dead code) */
static camera_t camera_create(v3 eye, v3 target, v3 up, double aspect, double vertical_fov_deg) {
  const double pi = 3.14159265358979323846;
  camera_t c;
  double half_height = tan(0.5 * (vertical_fov_deg * pi / 180.0));
  double half_width = aspect * half_height;
  c.cam.lower_left_x = -half_width; c.cam.lower_left_y = -half_height;
  c.cam.view_x = 2.0 * half_width; c.cam.view_y = 2.0 * half_height;
  v3 zp = v3_normalize(v3_sub(target, eye));
  v3 xp = v3_normalize(v3_cross(zp, v3_normalize(up)));
  v3 yp = v3_normalize(v3_cross(xp, zp));
  double r0[4] = {xp.x, xp.y, xp.z, -v3_dot(eye, xp)};
  double r1[4] = {yp.x, yp.y, yp.z, -v3_dot(eye, yp)};
  double r2[4] = {-zp.x, -zp.y, -zp.z, v3_dot(eye, zp)};
  double r3[4] = {0.0, 0.0, 0.0, 1.0};
  memcpy(c.look_at[0], r0, sizeof r0); memcpy(c.look_at[1], r1, sizeof r1); memcpy(c.look_at[2], r2, sizeof r2); memcpy(c.look_at[3], r3, sizeof r3);
  return c;
}
/* Camera.transform = Mat4.transform look_at, camera.ml:39-43,91 */
static inline v3 camera_transform(const camera_t* c, v3 p) {
  double v[4] = {p.x, p.y, p.z, 1.0};
  double x = dot4(v, c->look_at[0]), y = dot4(v, c->look_at[1]), z = dot4(v, c->look_at[2]), w = dot4(v, c->look_at[3]);
  return v3_scale(v3_make(x, y, z), 1.0 / w);
}
/* Camera.ray, camera.ml:93-102 */
static inline ray_t camera_ray(const ptx_camera* t, double dx, double dy) {
  v3 dir = v3_normalize(v3_make(t->lower_left_x + (t->view_x * dx), t->lower_left_y + (t->view_y * dy), -1.0));
  return ray_create(v3_make(0.0, 0.0, 0.0), dir);
}
ORC_API void orc_camera_create(const double* eye, const double* target, const double* up, double aspect, double fov_deg,
                               double* cam4_out, double* look_at16_out) {
  camera_t c = camera_create(v3_make(eye[0], eye[1], eye[2]), v3_make(target[0], target[1], target[2]), v3_make(up[0], up[1], up[2]), aspect, fov_deg);
  cam4_out[0] = c.cam.lower_left_x; cam4_out[1] = c.cam.lower_left_y; cam4_out[2] = c.cam.view_x; cam4_out[3] = c.cam.view_y;
  if (look_at16_out) memcpy(look_at16_out, c.look_at, sizeof c.look_at);
}
ORC_API void orc_camera_ray(const double* cam4, double dx, double dy, double* od_out /*6*/) {
  ptx_camera c = {cam4[0], cam4[1], cam4[2], cam4[3]};
  ray_t r = camera_ray(&c, dx, dy);
  od_out[0] = r.origin.x; od_out[1] = r.origin.y; od_out[2] = r.origin.z; od_out[3] = r.direction.x; od_out[4] = r.direction.y; od_out[5] = r.direction.z;
}

/* ------------------------------------------------------------------ OCaml 5 Random (third-party: stdlib/random.ml + runtime/prng.c, LXM L64X128) */
/* MD5 (RFC 1321) -- Digest.bytes */
static void md5(const uint8_t* msg, size_t len, uint8_t out[16]) {
  static const uint32_t K[64] = {
      0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
      0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
      0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
      0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
      0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
      0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
  static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                            4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
  uint32_t a0 = 0x67452301, b0 = 0xefcdab89, c0 = 0x98badcfe, d0 = 0x10325476;
  size_t padded = ((len + 8) / 64 + 1) * 64;
  uint8_t* buf = (uint8_t*)calloc(padded, 1);
  memcpy(buf, msg, len);
  buf[len] = 0x80;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; ++i) buf[padded - 8 + i] = (uint8_t)(bits >> (8 * i));
  for (size_t off = 0; off < padded; off += 64) {
    uint32_t M[16];
    for (int i = 0; i < 16; ++i) M[i] = (uint32_t)buf[off + 4 * i] | ((uint32_t)buf[off + 4 * i + 1] << 8) | ((uint32_t)buf[off + 4 * i + 2] << 16) | ((uint32_t)buf[off + 4 * i + 3] << 24);
    uint32_t A = a0, B = b0, C = c0, D = d0;
    for (int i = 0; i < 64; ++i) {
      uint32_t F; int g;
      if (i < 16) { F = (B & C) | (~B & D); g = i; }
      else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) % 16; }
      else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) % 16; }
      else { F = C ^ (B | ~D); g = (7 * i) % 16; }
      F = F + A + K[i] + M[g];
      A = D; D = C; C = B;
      B = B + ((F << S[i]) | (F >> (32 - S[i])));
    }
    a0 += A; b0 += B; c0 += C; d0 += D;
  }
  free(buf);
  uint32_t w[4] = {a0, b0, c0, d0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(w[i] >> (8 * j));
}
ORC_API void orc_md5(const uint8_t* msg, int len, uint8_t* out16) { md5(msg, (size_t)len, out16); }

typedef struct { uint64_t a, s, x0, x1; } lxm_t;
static uint64_t le64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }
/* Random.State.full_init / reinit with seed [| seed |] (Random.init seed), OCaml 5 stdlib/random.ml */
static lxm_t lxm_init(int64_t seed) {
  uint8_t b[9];
  for (int i = 0; i < 8; ++i) b[i] = (uint8_t)((uint64_t)seed >> (8 * i));
  uint8_t d1[16], d2[16];
  b[8] = 0x01; md5(b, 9, d1);
  b[8] = 0x02; md5(b, 9, d2);
  lxm_t s;
  s.a = le64(d1) | 1ULL;
  s.s = le64(d1 + 8);
  uint64_t i3 = le64(d2), i4 = le64(d2 + 8);
  s.x0 = i3 != 0 ? i3 : 1ULL;
  s.x1 = i4 != 0 ? i4 : 2ULL;
  return s;
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
/* caml_lxm_next, runtime/prng.c */
static uint64_t lxm_next(lxm_t* st) {
  uint64_t z = st->s + st->x0;
  z = (z ^ (z >> 32)) * 0xdaba0b6eb09322e3ULL;
  z = (z ^ (z >> 32)) * 0xdaba0b6eb09322e3ULL;
  z = (z ^ (z >> 32));
  st->s = st->s * 0xd1342543de82ef95ULL + st->a;
  uint64_t q0 = st->x0, q1 = st->x1;
  q1 ^= q0;
  q0 = rotl64(q0, 24);
  q0 = q0 ^ q1 ^ (q1 << 16);
  q1 = rotl64(q1, 37);
  st->x0 = q0; st->x1 = q1;
  return z;
}
/* Random.State.float: rawfloat (53-bit mantissa, rejects 0) * bound */
static double lxm_float(lxm_t* st, double bound) {
  for (;;) {
    uint64_t b = lxm_next(st);
    uint64_t n = b >> 11;
    if (n != 0) return (double)(int64_t)n * 0x1.p-53 * bound;
  }
}
ORC_API void orc_random_floats(int64_t seed, int n, double* out) {
  lxm_t s = lxm_init(seed);
  for (int i = 0; i < n; ++i) out[i] = lxm_float(&s, 1.0);
}

/* OCaml 4.x stdlib Random (lagged Fibonacci, 55 x 30-bit words, MD5 seeding) -- the generator a
 * pre-5.0 ("4.12+domains"/4.14) toolchain would have linked behind Base.Random. */
typedef struct { int64_t st[55]; int idx; } lf_t;
static void lf_init(lf_t* s, int64_t seed) {
  for (int i = 0; i < 55; ++i) s->st[i] = i;
  /* accu := Digest.string (accu ^ string_of_int seed.(k)) starting from "x" */
  uint8_t accu[16 + 32]; size_t alen = 1; accu[0] = 'x';
  char num[32]; int nl = snprintf(num, sizeof num, "%lld", (long long)seed);
  for (int i = 0; i <= 54 + 55; ++i) {
    int j = i % 55;
    uint8_t buf[16 + 32]; memcpy(buf, accu, alen); memcpy(buf + alen, num, (size_t)nl);
    uint8_t d[16]; md5(buf, alen + (size_t)nl, d);
    memcpy(accu, d, 16); alen = 16;
    int64_t ex = (int64_t)d[0] + ((int64_t)d[1] << 8) + ((int64_t)d[2] << 16) + ((int64_t)d[3] << 24);
    s->st[j] = (s->st[j] ^ ex) & 0x3FFFFFFF;
  }
  s->idx = 0;
}
static int64_t lf_bits(lf_t* s) {
  s->idx = (s->idx + 1) % 55;
  int64_t curval = s->st[s->idx];
  int64_t newval = s->st[(s->idx + 24) % 55] + (curval ^ ((curval >> 25) & 0x1FFFFFFF));
  int64_t newval30 = newval & 0x3FFFFFFF;
  s->st[s->idx] = newval30;
  return newval30;
}
static double lf_float(lf_t* s, double bound) {
  double scale = 1073741824.0;
  double r1 = (double)lf_bits(s);
  double r2 = (double)lf_bits(s);
  return ((r1 / scale + r2) / scale) * bound;
}
/* rng_kind: 0 = OCaml 5 LXM (the reference's declared toolchain, dune-project:15), 1 = OCaml 4.x */
typedef struct { int kind; lxm_t lxm; lf_t lf; } rng_t;
static void rng_init(rng_t* r, int kind, int64_t seed) { r->kind = kind; if ((kind & 1) == 0) r->lxm = lxm_init(seed); else lf_init(&r->lf, seed); }
static int64_t rng_bits30(rng_t* r) { return (r->kind & 1) ? lf_bits(&r->lf) : (int64_t)(lxm_next(&r->lxm) & 0x3FFFFFFF); }
/* Base.Random.State.float (third-party, base/src/random.ml): rawfloat from two 30-bit draws,
 * ((r1 * 2^-30) + r2) * 2^-30, retried if it rounds up to 1.0 */
static double base_rawfloat(rng_t* r, int swap) {
  for (;;) {
    double r1 = (double)rng_bits30(r);
    double r2 = (double)rng_bits30(r);
    if (swap) { double t = r1; r1 = r2; r2 = t; }
    double result = ((r1 * 0x1p-30) + r2) * 0x1p-30;
    if (result < 1.0) return result;
  }
}
/* kind: 0 LXM + stdlib float; 1 OCaml4 + stdlib float; 2 LXM + Base rawfloat; 3 OCaml4 + Base rawfloat; 4/5 = 2/3 with r1,r2 swapped */
static double rng_float(rng_t* r, double bound) {
  switch (r->kind) {
    case 0: return lxm_float(&r->lxm, bound);
    case 1: return lf_float(&r->lf, bound);
    case 2: case 3: return base_rawfloat(r, 0) * bound;
    default: return base_rawfloat(r, 1) * bound;
  }
}
ORC_API void orc_random_floats_kind(int kind, int64_t seed, int n, double* out) {
  rng_t r; rng_init(&r, kind, seed);
  for (int i = 0; i < n; ++i) out[i] = rng_float(&r, 1.0);
}

/* ------------------------------------------------------------------ scene description builders */
/* growable description owned by the oracle; the arrays are handed to tests / orc_scene_create */
typedef struct orc_desc {
  ptx_scene_desc d;
  int cap_s, cap_v, cap_t, cap_m, cap_x;
  double *sx, *sy, *sz, *sr; int32_t* sm;
  double *vx, *vy, *vz;
  int32_t* ti; double* tuv; int32_t* tm;
  double floor_v[18], floor_uv[12]; int32_t floor_m[2];
  ptx_material* mats; ptx_texture* texs;
} orc_desc;

static void* grow(void* p, size_t elt, int* cap, int need) {
  if (need <= *cap) return p;
  int nc = *cap ? *cap : 64;
  while (nc < need) nc *= 2;
  *cap = nc;
  return realloc(p, elt * (size_t)nc);
}
static void desc_sync(orc_desc* o) {
  o->d.sphere_x = o->sx; o->d.sphere_y = o->sy; o->d.sphere_z = o->sz; o->d.sphere_r = o->sr; o->d.sphere_material = o->sm;
  o->d.vertex_x = o->vx; o->d.vertex_y = o->vy; o->d.vertex_z = o->vz;
  o->d.tri_indices = o->ti; o->d.tri_uv = o->tuv; o->d.tri_material = o->tm;
  o->d.floor_vertices = o->floor_v; o->d.floor_uv = o->floor_uv; o->d.floor_material = o->floor_m;
  o->d.materials = o->mats; o->d.textures = o->texs;
}
static int desc_add_texture(orc_desc* o, ptx_texture t) {
  int cap = o->cap_x; o->texs = (ptx_texture*)grow(o->texs, sizeof(ptx_texture), &cap, o->d.n_textures + 1); o->cap_x = cap;
  o->texs[o->d.n_textures] = t; return o->d.n_textures++;
}
static int desc_add_material(orc_desc* o, ptx_material m) {
  int cap = o->cap_m; o->mats = (ptx_material*)grow(o->mats, sizeof(ptx_material), &cap, o->d.n_materials + 1); o->cap_m = cap;
  o->mats[o->d.n_materials] = m; return o->d.n_materials++;
}
static int desc_solid_tex(orc_desc* o, double r, double g, double b) {
  ptx_texture t; memset(&t, 0, sizeof t); t.kind = PTX_TEX_SOLID; t.even[0] = r; t.even[1] = g; t.even[2] = b; return desc_add_texture(o, t);
}
static int desc_checker_tex(orc_desc* o, int w, int h, const double* even, const double* odd) {
  ptx_texture t; memset(&t, 0, sizeof t); t.kind = PTX_TEX_CHECKER; t.width = w; t.height = h;
  memcpy(t.even, even, 24); memcpy(t.odd, odd, 24); return desc_add_texture(o, t);
}
static int desc_mat(orc_desc* o, int kind, int tex, double index) {
  ptx_material m; memset(&m, 0, sizeof m); m.kind = kind; m.texture = tex; m.index = index; return desc_add_material(o, m);
}
static void desc_add_sphere(orc_desc* o, v3 c, double r, int mat) {
  int n = o->d.n_spheres, cap = o->cap_s;
  o->sx = (double*)grow(o->sx, 8, &cap, n + 1); cap = o->cap_s;
  o->sy = (double*)grow(o->sy, 8, &cap, n + 1); cap = o->cap_s;
  o->sz = (double*)grow(o->sz, 8, &cap, n + 1); cap = o->cap_s;
  o->sr = (double*)grow(o->sr, 8, &cap, n + 1); cap = o->cap_s;
  o->sm = (int32_t*)grow(o->sm, 4, &cap, n + 1); o->cap_s = cap;
  o->sx[n] = c.x; o->sy[n] = c.y; o->sz[n] = c.z; o->sr[n] = r; o->sm[n] = mat; o->d.n_spheres = n + 1;
}
static int desc_add_vertex(orc_desc* o, v3 p) {
  int n = o->d.n_vertices, cap = o->cap_v;
  o->vx = (double*)grow(o->vx, 8, &cap, n + 1); cap = o->cap_v;
  o->vy = (double*)grow(o->vy, 8, &cap, n + 1); cap = o->cap_v;
  o->vz = (double*)grow(o->vz, 8, &cap, n + 1); o->cap_v = cap;
  o->vx[n] = p.x; o->vy[n] = p.y; o->vz[n] = p.z; o->d.n_vertices = n + 1; return n;
}
static void desc_add_tri_idx(orc_desc* o, int a, int b, int c, const double* uv6, int mat) {
  int n = o->d.n_triangles, cap = o->cap_t;
  o->ti = (int32_t*)grow(o->ti, 12, &cap, n + 1); cap = o->cap_t;
  o->tuv = (double*)grow(o->tuv, 48, &cap, n + 1); cap = o->cap_t;
  o->tm = (int32_t*)grow(o->tm, 4, &cap, n + 1); o->cap_t = cap;
  o->ti[3 * n] = a; o->ti[3 * n + 1] = b; o->ti[3 * n + 2] = c; memcpy(&o->tuv[6 * n], uv6, 48); o->tm[n] = mat; o->d.n_triangles = n + 1;
}
static void desc_add_tri(orc_desc* o, v3 a, v3 b, v3 c, const double* uv6, int mat) {
  int ia = desc_add_vertex(o, a), ib = desc_add_vertex(o, b), ic = desc_add_vertex(o, c);
  desc_add_tri_idx(o, ia, ib, ic, uv6, mat);
}
ORC_API void orc_desc_destroy(orc_desc* o) {
  if (!o) return;
  free(o->sx); free(o->sy); free(o->sz); free(o->sr); free(o->sm); free(o->vx); free(o->vy); free(o->vz); free(o->ti); free(o->tuv); free(o->tm); free(o->mats); free(o->texs); free(o);
}
ORC_API const ptx_scene_desc* orc_desc_get(orc_desc* o) { desc_sync(o); return &o->d; }

static void sky_background(ptx_background* bg) { /* shirley_spheres/bin/main.ml:104-110 */
  memset(bg, 0, sizeof *bg); bg->kind = PTX_BG_SKY;
  bg->horizon[0] = bg->horizon[1] = bg->horizon[2] = 1.0;
  bg->zenith[0] = 0.5; bg->zenith[1] = 0.7; bg->zenith[2] = 1.0;
}

/* Shirley_spheres.spheres + camera + transform to camera space, shirley_spheres/bin/main.ml:26-102,250-260.
 * no_simd selects Array_leaf (cutoff 4, main.ml:115-130) instead of Simd_leaf (cutoff leaf_size () = 16). */
ORC_API orc_desc* orc_desc_shirley_rng(int width, int height, int no_simd, int64_t seed, int rng_kind) {
  orc_desc* o = (orc_desc*)calloc(1, sizeof(orc_desc));
  rng_t rng; rng_init(&rng, rng_kind, seed); /* Random.init 42 */
  camera_t cam = camera_create(v3_make(13.0, 2.0, 4.5), v3_make(0.0, 0.0, 0.0), v3_make(0.0, 1.0, 0.0), (double)width / (double)height, 20.0);
  /* ground, :38-43 */
  double ga[3] = {0.2, 0.3, 0.1}, gb[3] = {0.9, 0.9, 0.9};
  int checks = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_checker_tex(o, 1000, 2000, ga, gb), 0.0);
  desc_add_sphere(o, v3_make(0.0, -1000.0, 0.0), 1000.0, checks);
  /* big_spheres, :45-54 */
  int glass = desc_mat(o, PTX_MAT_DIELECTRIC, 0, 1.5);
  int metal = desc_mat(o, PTX_MAT_METAL, desc_solid_tex(o, 0.7, 0.6, 0.5), 0.0);
  int blue = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, 0.1, 0.1, 0.7), 0.0);
  desc_add_sphere(o, v3_make(-4.0, 1.0, 0.0), 1.0, glass);
  desc_add_sphere(o, v3_make(0.0, 1.0, 0.0), 1.0, metal);
  desc_add_sphere(o, v3_make(4.0, 1.0, 0.0), 1.0, blue);
  /* small spheres, :82-101: a outer, b inner, both -11..11 */
  for (int a = -11; a <= 11; ++a) {
    for (int b = -11; b <= 11; ++b) {
      double x = (double)a + (0.9 * rng_float(&rng, 1.0)); /* perturb, :82 */
      double z = (double)b + (0.9 * rng_float(&rng, 1.0));
      double radius = 0.2;
      v3 center = v3_make(x, radius, z);
      v3 p = v3_make(4.0, radius, 0.0);
      if (v3_quadrance(v3_sub(p, center)) > 0.81) {
        /* random_material, :70-80 */
        double roll = rng_float(&rng, 1.0);
        int material;
        if (roll < 0.8) {
          /* random_lambertian :65-68 : V3.Infix.(random_v3 () * random_v3 ()).  OCaml evaluates the
           * right operand first, but the component products commute, so draw order only pairs
           * draw k with draw k+3. */
          double r1[3], r2[3];
          for (int k = 0; k < 3; ++k) r1[k] = rng_float(&rng, 1.0);
          for (int k = 0; k < 3; ++k) r2[k] = rng_float(&rng, 1.0);
          material = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, r2[0] * r1[0], r2[1] * r1[1], r2[2] * r1[2]), 0.0);
        } else if (roll < 0.95) {
          double zc = (0.5 * rng_float(&rng, 1.0)) + 0.5;
          material = desc_mat(o, PTX_MAT_METAL, desc_solid_tex(o, zc, zc, zc), 0.0);
        } else {
          material = glass;
        }
        desc_add_sphere(o, center, radius, material);
      }
    }
  }
  /* Sphere.transform ~f:(Camera.transform camera), main.ml:258-260 */
  for (int i = 0; i < o->d.n_spheres; ++i) {
    v3 c = camera_transform(&cam, v3_make(o->sx[i], o->sy[i], o->sz[i]));
    o->sx[i] = c.x; o->sy[i] = c.y; o->sz[i] = c.z;
  }
  o->d.camera = cam.cam;
  sky_background(&o->d.background);
  o->d.leaf_kind = no_simd ? PTX_LEAF_ARRAY : PTX_LEAF_SIMD;
  o->d.length_cutoff = no_simd ? 4 : 16;
  o->d.num_bins = 32;
  desc_sync(o);
  return o;
}

ORC_API orc_desc* orc_desc_shirley(int width, int height, int no_simd, int64_t seed) { return orc_desc_shirley_rng(width, height, no_simd, seed, 2); }

static const double T00[2] = {0.0, 0.0}, T01[2] = {0.0, 1.0}, T10[2] = {1.0, 0.0}, T11[2] = {1.0, 1.0};

/* triangle_fan / quad, cornell-box/bin/main.ml:30-48.  triangle_fan conses, so the fan comes out
 * in REVERSE order: quad = [tri o c d; tri o b c]. */
typedef struct { v3 a, b, c; double uv[6]; int material; } ctri_t;
static void cornell_quad(ctri_t out[2], int material, v3 a, v3 u, v3 v) {
  v3 b = v3_add(a, v), c = v3_add(b, u), d = v3_add(a, u);
  /* pts = [a,t00; b,t10; c,t11; d,t01]; o = a; loop [b;c;d] -> tris = [tri a c d; tri a b c] */
  double uv1[6] = {T00[0], T00[1], T11[0], T11[1], T01[0], T01[1]};
  double uv2[6] = {T00[0], T00[1], T10[0], T10[1], T11[0], T11[1]};
  out[0].a = a; out[0].b = c; out[0].c = d; memcpy(out[0].uv, uv1, sizeof uv1); out[0].material = material;
  out[1].a = a; out[1].b = b; out[1].c = c; memcpy(out[1].uv, uv2, sizeof uv2); out[1].material = material;
}
/* Base List.concat_no_order = fold ~init:[] ~f:(fun acc l -> rev_append l acc) (third-party, Base
 * list.ml): the LAST list comes first and every list is reversed.  quads: n_quads x 2 triangles. */
static void emit_concat_no_order(orc_desc* o, const camera_t* cam, ctri_t (*quads)[2], int n_quads) {
  for (int q = n_quads - 1; q >= 0; --q)
    for (int k = 1; k >= 0; --k) {
      const ctri_t* t = &quads[q][k];
      /* Triangle.transform ~f:(Camera.transform camera), cornell-box/bin/main.ml:24-27,218 */
      desc_add_tri(o, camera_transform(cam, t->a), camera_transform(cam, t->b), camera_transform(cam, t->c), t->uv, t->material);
    }
}

/* Cornell-box geometry / materials / camera, cornell-box/bin/main.ml:43-91,172-218, for the PATH integrator.
 * The reference lights this scene with a PPM point light that the path integrator cannot see
 * (SURVEY.md section 8 A20); the documented extension: the ceiling quad's material gets
 * emit = (emit, emit, emit) through the Hit.emit slot, background black. */
ORC_API orc_desc* orc_desc_cornell(int width, int height, double ceiling_emit) {
  const double pi = 3.14159265358979323846;
  orc_desc* o = (orc_desc*)calloc(1, sizeof(orc_desc));
  double fov = (2.0 * atan(0.5)) * 180.0 / pi; /* main.ml:177-180 */
  camera_t cam = camera_create(v3_make(0.5, 0.5, -1.0), v3_make(0.5, 0.5, 0.0), v3_make(0.0, 1.0, 0.0), (double)width / (double)height, fov);
  v3 ux = v3_make(1.0, 0.0, 0.0), uy = v3_make(0.0, 1.0, 0.0), uz = v3_make(0.0, 0.0, 1.0), org = v3_make(0.0, 0.0, 0.0);
  /* light_enclosure', :190-210 : metal (0.30, 0.999, 0.30) quads [r; f; l; b] */
  int encl = desc_mat(o, PTX_MAT_METAL, desc_solid_tex(o, 0.30, 0.999, 0.30), 0.0);
  {
    double r = 0.05;
    v3 rx = v3_scale(ux, r), ry = v3_scale(uy, r), rz = v3_scale(uz, r);
    v3 lc = v3_make(0.5, 0.82, 0.5);
    v3 a = v3_sub(v3_sub(v3_sub(lc, rx), ry), rz);
    v3 b = v3_add(v3_sub(v3_add(lc, rx), ry), rz);
    ctri_t q[4][2];
    /* local quad p u v = quad p (2u) (2v) */
    cornell_quad(q[0], encl, a, v3_scale(rz, 2.0), v3_scale(ry, 2.0));         /* r */
    cornell_quad(q[1], encl, a, v3_scale(ry, 2.0), v3_scale(rx, 2.0));         /* f */
    cornell_quad(q[2], encl, b, v3_scale(v3_neg(rz), 2.0), v3_scale(ry, 2.0)); /* l */
    cornell_quad(q[3], encl, b, v3_scale(rx, 2.0), v3_scale(ry, 2.0));         /* b */
    emit_concat_no_order(o, &cam, q, 4);
  }
  /* empty_box, :52-68 */
  int red = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, 0.7, 0.0, 0.0), 0.0);
  int blue = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, 0.0, 0.0, 0.7), 0.0);
  int grey = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, 0.7, 0.7, 0.7), 0.0);
  double ca[3] = {0.2, 0.3, 0.1}, cb[3] = {0.9, 0.9, 0.9};
  int checker = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_checker_tex(o, 10, 10, ca, cb), 0.0);
  int ceil_mat = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, 0.7, 0.7, 0.7), 0.0);
  o->mats[ceil_mat].emit[0] = o->mats[ceil_mat].emit[1] = o->mats[ceil_mat].emit[2] = ceiling_emit;
  {
    ctri_t q[5][2];
    cornell_quad(q[0], red, org, uz, uy);      /* right_wall */
    cornell_quad(q[1], blue, ux, uz, uy);      /* left_wall */
    cornell_quad(q[2], checker, org, ux, uz);  /* floor */
    cornell_quad(q[3], ceil_mat, uy, ux, uz);  /* ceiling (grey + emitter extension) */
    cornell_quad(q[4], grey, uz, ux, uy);      /* rear_wall */
    emit_concat_no_order(o, &cam, q, 5);
  }
  /* spheres, :70-91 */
  {
    double radius = 0.20;
    int m_metal = desc_mat(o, PTX_MAT_METAL, desc_solid_tex(o, 1.0, 1.0, 1.0), 0.0);
    int m_glass = desc_mat(o, PTX_MAT_DIELECTRIC, 0, 1.5);
    int m_back = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, 0.75, 0.75, 0.75), 0.0);
    desc_add_sphere(o, camera_transform(&cam, v3_make(1.0 - 0.1 - radius, radius, 1.0 - 0.2 - radius)), radius, m_metal);
    desc_add_sphere(o, camera_transform(&cam, v3_make(0.1 + radius, 0.1 + radius, 0.2 + radius)), radius, m_glass);
    double big = 10.0;
    desc_add_sphere(o, camera_transform(&cam, v3_make(0.5, 0.5, -2.0 - big)), big, m_back);
  }
  o->d.camera = cam.cam;
  memset(&o->d.background, 0, sizeof o->d.background); o->d.background.kind = PTX_BG_BLACK;
  o->d.leaf_kind = PTX_LEAF_ARRAY; o->d.length_cutoff = 2; o->d.num_bins = 32; /* main.ml:159-168 */
  desc_sync(o);
  return o;
}

/* splitmix64: generator for the SYNTHETIC ganesha-like mesh only (the real ganesha.ply is not in the
 * reference repository, ganesha/README.md:1) */
static uint64_t splitmix64(uint64_t* s) { uint64_t z = (*s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

/* Ganesha-like scene: ganesha/bin/main.ml:30-35 (camera), :50-119 (mesh, Lambertian (.1,.7,.2), tex_coords
 * t00,t01,t11, cutoff 8), :205-260 (checker floor 500x500 of two 10000-wide triangles under the mesh,
 * tested before the tree).  The mesh itself is synthetic: a closed lat-long surface of ~n_target triangles,
 * radially displaced by a seeded sum of lobes, placed where the real model sits (bbox ~[-?]) -- see DESIGN.md.
 * World-space vertices are stored as float32-rounded values like a PLY `float` property would give. */
ORC_API orc_desc* orc_desc_ganesha_like(int width, int height, int n_target, uint64_t seed) {
  const double pi = 3.14159265358979323846;
  orc_desc* o = (orc_desc*)calloc(1, sizeof(orc_desc));
  v3 up = v3_make(-0.00212272, 0.998201, -0.0599264);
  camera_t cam = camera_create(v3_make(328.0, 70.282, 345.0), v3_make(328.0, 10.0, 0.0), up, (double)width / (double)height, 30.0);
  int mat = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_solid_tex(o, 0.1, 0.7, 0.2), 0.0);
  /* grid: nu x nv quads -> 2*nu*nv triangles (poles collapse but stay as degenerate-free fans) */
  int nv = (int)floor(sqrt((double)n_target / 4.0)); if (nv < 4) nv = 4;
  int nu = 2 * nv;
  uint64_t st = seed;
  enum { NL = 24 };
  double lobe[NL][5];
  for (int i = 0; i < NL; ++i) {
    double u1 = (double)(splitmix64(&st) >> 11) * 0x1.p-53, u2 = (double)(splitmix64(&st) >> 11) * 0x1.p-53;
    double u3 = (double)(splitmix64(&st) >> 11) * 0x1.p-53, u4 = (double)(splitmix64(&st) >> 11) * 0x1.p-53;
    double zc = 2.0 * u1 - 1.0, ph = 2.0 * pi * u2, rr = sqrt(1.0 - zc * zc);
    lobe[i][0] = rr * cos(ph); lobe[i][1] = zc; lobe[i][2] = rr * sin(ph);
    lobe[i][3] = 0.10 + 0.35 * u3;      /* amplitude */
    lobe[i][4] = 4.0 + 28.0 * u4;       /* sharpness */
  }
  v3 centre = v3_make(328.0, 42.0, 20.0);
  v3 radii = v3_make(30.0, 44.0, 26.0);
  int* vid = (int*)malloc(sizeof(int) * (size_t)(nu * (nv + 1)));
  for (int j = 0; j <= nv; ++j) {
    for (int i = 0; i < nu; ++i) {
      double th = pi * (double)j / (double)nv, ph = 2.0 * pi * (double)i / (double)nu;
      v3 n = v3_make(sin(th) * cos(ph), cos(th), sin(th) * sin(ph));
      if (j == 0) n = v3_make(0.0, 1.0, 0.0);
      if (j == nv) n = v3_make(0.0, -1.0, 0.0);
      double disp = 1.0;
      for (int l = 0; l < NL; ++l) {
        double dt = n.x * lobe[l][0] + n.y * lobe[l][1] + n.z * lobe[l][2];
        disp += lobe[l][3] * exp(lobe[l][4] * (dt - 1.0));
      }
      disp += 0.02 * sin(37.0 * ph) * sin(29.0 * th);
      v3 p = v3_make(centre.x + radii.x * disp * n.x, centre.y + radii.y * disp * n.y, centre.z + radii.z * disp * n.z);
      p = v3_make((double)(float)p.x, (double)(float)p.y, (double)(float)p.z); /* PLY float property */
      v3 pc = camera_transform(&cam, p); /* Mesh.create, main.ml:74-79 */
      vid[j * nu + i] = desc_add_vertex(o, pc);
    }
  }
  double uv[6] = {T00[0], T00[1], T01[0], T01[1], T11[0], T11[1]}; /* tex_coords = (t00, t01, t11), :111 */
  for (int j = 0; j < nv; ++j) {
    for (int i = 0; i < nu; ++i) {
      int i1 = (i + 1) % nu;
      int a = vid[j * nu + i], b = vid[j * nu + i1], c = vid[(j + 1) * nu + i1], d = vid[(j + 1) * nu + i];
      if (j != 0) desc_add_tri_idx(o, a, b, c, uv, mat);       /* at the north pole a == b geometrically: skip the sliver */
      if (j != nv - 1) desc_add_tri_idx(o, a, c, d, uv, mat);  /* at the south pole c == d */
    }
  }
  free(vid);
  /* Floor, main.ml:205-245: needs the mesh bbox in camera space = tree bbox */
  {
    bbox_t bb;
    bb.min = v3_make(o->vx[0], o->vy[0], o->vz[0]); bb.max = bb.min;
    /* Triangles.bbox tree = union of triangle bboxes (every vertex is used) */
    for (int i = 0; i < o->d.n_triangles; ++i) {
      prim_t t; memset(&t, 0, sizeof t);
      int ia = o->ti[3 * i], ib = o->ti[3 * i + 1], ic = o->ti[3 * i + 2];
      t.a = v3_make(o->vx[ia], o->vy[ia], o->vz[ia]); t.b = v3_make(o->vx[ib], o->vy[ib], o->vz[ib]); t.c = v3_make(o->vx[ic], o->vy[ic], o->vz[ic]);
      bbox_t tb = triangle_bbox(&t);
      bb = (i == 0) ? tb : bbox_union(bb, tb);
    }
    v3 ctr = bbox_center(bb);
    v3 center = v3_make(ctr.x, bb.min.y, ctr.z);
    double s = 5000.0;
    v3 xp = v3_scale(v3_make(1.0, 0.0, 0.0), s), zp = v3_scale(v3_make(0.0, 0.0, 1.0), s);
    v3 pa = v3_add(center, v3_neg(v3_add(xp, zp)));
    v3 pb = v3_add(pa, v3_scale(xp, 2.0));
    v3 pc = v3_add(pb, v3_scale(zp, 2.0));
    v3 pd = v3_add(pa, v3_scale(zp, 2.0));
    double ea[3] = {0.2, 0.3, 0.1}, eb[3] = {0.9, 0.9, 0.9};
    int fm = desc_mat(o, PTX_MAT_LAMBERTIAN, desc_checker_tex(o, 500, 500, ea, eb), 0.0);
    /* f1 = a b c (t00,t01,t11); f2 = a c d (t00,t11,t10) */
    double f1[9] = {pa.x, pa.y, pa.z, pb.x, pb.y, pb.z, pc.x, pc.y, pc.z};
    double f2[9] = {pa.x, pa.y, pa.z, pc.x, pc.y, pc.z, pd.x, pd.y, pd.z};
    double u1[6] = {T00[0], T00[1], T01[0], T01[1], T11[0], T11[1]};
    double u2[6] = {T00[0], T00[1], T11[0], T11[1], T10[0], T10[1]};
    memcpy(o->floor_v, f1, sizeof f1); memcpy(o->floor_v + 9, f2, sizeof f2);
    memcpy(o->floor_uv, u1, sizeof u1); memcpy(o->floor_uv + 6, u2, sizeof u2);
    o->floor_m[0] = fm; o->floor_m[1] = fm;
    o->d.n_floor_triangles = 2;
  }
  o->d.camera = cam.cam;
  sky_background(&o->d.background); /* documented extension: Shirley's sky lights the scene */
  o->d.leaf_kind = PTX_LEAF_ARRAY; o->d.length_cutoff = 8; o->d.num_bins = 32; /* main.ml:158 */
  desc_sync(o);
  return o;
}

/* ------------------------------------------------------------------ Integrator (integrator.ml) */
typedef struct { const double* alpha; int offset; } sampler_t;
static inline double sample_dim(const sampler_t* s, int dimension) { return lds_get(s->alpha, s->offset, dimension); }

static inline v3 add_mul(v3 a, v3 b, v3 c) { return v3_fma(b, c, a); } /* integrator.ml:29 : Color.fma b c a = b*c+a */

/* Pdf.eval Diffuse, pdf.ml:11-15 */
static inline double pdf_eval_diffuse(v3 dir) { const double pi = 3.14159265358979323846; return (dir.z < 0.0) ? 0.0 : dir.z / pi; }

/* path_tracer, integrator.ml:16-69 */
static v3 trace_path(const orc_scene* sc, double cx, double cy, const sampler_t* smp, int max_bounces, counters_t* ct) {
  ray_t ray = camera_ray(&sc->camera, cx, cy);
  int samples_index = 2;
  v3 emit0 = v3_make(0.0, 0.0, 0.0), attn0 = v3_make(1.0, 1.0, 1.0);
  const v3 black = v3_make(0.0, 0.0, 0.0);
  for (;;) {
    if (max_bounces <= 0) return add_mul(emit0, attn0, black);
    max_bounces = max_bounces - 1;
    hit_t h;
    if (!scene_intersect(sc, &ray, &h, NULL, NULL, ct)) return add_mul(emit0, attn0, scene_background(sc, &ray));
    v3 emit = h.emit;
    int j = samples_index;
    double u = sample_dim(smp, j), v = sample_dim(smp, j + 1);
    samples_index = j + 2;
    scatter_t s = hit_scatter(&sc->mt, &h, u);
    if (s.kind == SC_ABSORB) return add_mul(emit0, attn0, emit);
    if (s.kind == SC_SPECULAR) {
      v3 ne = add_mul(emit, s.attenuation, emit0);
      attn0 = v3_mul(s.attenuation, attn0);
      emit0 = ne;
      ray = s.ray;
      continue;
    }
    /* Diffuse */
    v3 dir = unit_square_to_hemisphere(u, v);         /* Pdf.sample diffuse_plus_light = Pdf.diffuse */
    double diffuse_pd = pdf_eval_diffuse(dir);
    if (diffuse_pd == 0.0) return add_mul(emit0, attn0, emit);
    double divisor = pdf_eval_diffuse(dir);
    double pd = diffuse_pd / divisor;
    if (!pt_isfinite(pd)) return add_mul(emit0, attn0, emit);
    ray_t scattered = sspace_world_ray(&h.shader_space, dir);
    v3 attenuation = v3_scale(s.attenuation, pd);
    v3 ne = add_mul(emit, attenuation, emit0);
    attn0 = v3_mul(attenuation, attn0);
    emit0 = ne;
    ray = scattered;
  }
}

/* one sample as render_tile computes it, integrator.ml:96-109: returns colour and (dx, dy) */
static v3 sample_pixel(const orc_scene* sc, const double* alpha, int width, int height, int spp, int max_bounces,
                       int gx, int gy, int pass, double* dx_out, double* dy_out, counters_t* ct) {
  double widthf = 1.0 / (double)width, heightf = 1.0 / (double)height; /* 1 // t.width */
  sampler_t smp;
  smp.alpha = alpha;
  smp.offset = (gy * width) + gx + (pass * spp); /* sic: pass * samples_per_pixel, integrator.ml:98 */
  double xf = (double)gx, yf = (double)gy;
  double dx = sample_dim(&smp, 0), dy = sample_dim(&smp, 1);
  double cx = (xf + dx) * widthf;
  double cy = 1.0 - ((yf + dy) * heightf);
  if (dx_out) *dx_out = dx;
  if (dy_out) *dy_out = dy;
  if (ct) ct->samples += 1;
  return trace_path(sc, cx, cy, &smp, max_bounces, ct);
}

ORC_API void orc_trace_samples(const orc_scene* sc, int width, int height, int spp, int max_bounces, int64_t n,
                               const int32_t* xs, const int32_t* ys, const int32_t* passes, double* rgb_out, int64_t* counters_out) {
  int dim = 2 + 2 * max_bounces;
  double* alpha = (double*)malloc(sizeof(double) * (size_t)dim);
  orc_lds_alpha(dim, alpha);
  counters_t ct; memset(&ct, 0, sizeof ct);
  for (int64_t i = 0; i < n; ++i) {
    v3 c = sample_pixel(sc, alpha, width, height, spp, max_bounces, xs[i], ys[i], passes[i], NULL, NULL, &ct);
    rgb_out[3 * i] = c.x; rgb_out[3 * i + 1] = c.y; rgb_out[3 * i + 2] = c.z;
  }
  if (counters_out) { counters_out[0] = ct.samples; counters_out[1] = ct.segments; counters_out[2] = ct.nodes_tested; counters_out[3] = ct.prims_tested; counters_out[4] = ct.floor_tested; }
  free(alpha);
}

/* Scene.intersect for explicit rays (t, build-list primitive index or -1) */
ORC_API void orc_intersect_rays(const orc_scene* sc, int64_t n, const double* origins, const double* directions, double* t_out, int32_t* prim_out, int64_t* counters_out) {
  counters_t ct; memset(&ct, 0, sizeof ct);
  for (int64_t i = 0; i < n; ++i) {
    ray_t r = ray_create(v3_make(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), v3_make(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]));
    hit_t h; double t = 0.0; int id = -1;
    if (scene_intersect(sc, &r, &h, &t, &id, &ct)) { t_out[i] = t; prim_out[i] = id; }
    else { t_out[i] = 0.0; prim_out[i] = -1; }
  }
  if (counters_out) { counters_out[0] = 0; counters_out[1] = ct.segments; counters_out[2] = ct.nodes_tested; counters_out[3] = ct.prims_tested; counters_out[4] = ct.floor_tested; }
}

typedef struct {
  const orc_scene* sc;
  int width, height, spp, max_bounces;
  const double* kernel; int pixel_radius;
  tile_t* tiles; int n_tiles;
  film_tile_t* films;
  double* raw; /* optional per-pixel raw sums (H*W*3), pass order */
  int next; pthread_mutex_t mu;
  counters_t ct; int count;
  int row_begin, row_end; /* raw-sum mode restriction */
} render_job_t;

/* render_tile, integrator.ml:91-112 */
static void render_one_tile(render_job_t* job, int ti, counters_t* ct) {
  tile_t tile = job->tiles[ti];
  film_tile_t ft = film_tile_create(tile, job->kernel, job->pixel_radius);
  int dim = 2 + 2 * job->max_bounces; /* create_sampler, :89 -- built once per tile, :95 */
  double* alpha = (double*)malloc(sizeof(double) * (size_t)dim);
  orc_lds_alpha(dim, alpha);
  for (int pass = 0; pass < job->spp; ++pass) {
    for (int local_y = 0; local_y < tile.height; ++local_y) { /* Tile.iter, tile.ml:71-79 */
      int global_y = local_y + tile.row;
      for (int local_x = 0; local_x < tile.width; ++local_x) {
        int global_x = local_x + tile.col;
        double dx, dy;
        v3 color = sample_pixel(job->sc, alpha, job->width, job->height, job->spp, job->max_bounces, global_x, global_y, pass, &dx, &dy, ct);
        double x = (double)local_x + dx, y = (double)local_y + dy;
        film_tile_write_sample(&ft, x, y, color);
        if (job->raw) {
          double* p = &job->raw[((size_t)global_y * job->width + global_x) * 3];
          p[0] = p[0] + color.x; p[1] = p[1] + color.y; p[2] = p[2] + color.z;
        }
      }
    }
  }
  free(alpha);
  job->films[ti] = ft;
}
static void* render_worker(void* arg) {
  render_job_t* job = (render_job_t*)arg;
  counters_t ct; memset(&ct, 0, sizeof ct);
  for (;;) {
    pthread_mutex_lock(&job->mu);
    int ti = job->next++;
    pthread_mutex_unlock(&job->mu);
    if (ti >= job->n_tiles) break;
    render_one_tile(job, ti, job->count ? &ct : NULL);
  }
  pthread_mutex_lock(&job->mu);
  job->ct.samples += ct.samples; job->ct.segments += ct.segments; job->ct.nodes_tested += ct.nodes_tested; job->ct.prims_tested += ct.prims_tested; job->ct.floor_tested += ct.floor_tested;
  pthread_mutex_unlock(&job->mu);
  return NULL;
}

/* Integrator.render, integrator.ml:130-156.  The reference runs recommended_domain_count()-1 worker domains and
 * stitches on the main domain in COMPLETION order (non-deterministic last-ulp at tile seams); here the stitch is in
 * tile-list order so the oracle is reproducible.  rgb_out: W*H*3 post-gamma; raw_out (optional): per-pixel sums. */
ORC_API int orc_render(const orc_scene* sc, int width, int height, int spp, int max_bounces, int threads,
                       double* rgb_out, double* raw_out, int64_t* counters_out, double* ms_out) {
  double t0 = now_ms();
  render_job_t job; memset(&job, 0, sizeof job);
  job.sc = sc; job.width = width; job.height = height; job.spp = spp; job.max_bounces = max_bounces;
  int max_area = 32 * 32; /* Int.pow 32 2 */
  tile_t root = {0, 0, width, height};
  int cap = 0;
  tile_split_rec(root, max_area, &job.tiles, &job.n_tiles, &cap);
  int pixel_radius = 1;
  double kern[9];
  orc_filter_binomial(5, pixel_radius, kern, NULL); /* Binomial.create ~order:5 ~pixel_radius */
  job.kernel = kern; job.pixel_radius = pixel_radius;
  job.films = (film_tile_t*)calloc((size_t)job.n_tiles, sizeof(film_tile_t));
  job.raw = raw_out; if (raw_out) memset(raw_out, 0, sizeof(double) * (size_t)width * height * 3);
  job.count = counters_out != NULL;
  pthread_mutex_init(&job.mu, NULL);
  if (threads < 1) threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  for (int i = 1; i < threads; ++i) pthread_create(&th[i], NULL, render_worker, &job);
  render_worker(&job);
  for (int i = 1; i < threads; ++i) pthread_join(th[i], NULL);
  free(th);
  /* stitch_tile, integrator.ml:114-128 */
  memset(rgb_out, 0, sizeof(double) * (size_t)width * height * 3);
  for (int ti = 0; ti < job.n_tiles; ++ti) {
    film_tile_t* ft = &job.films[ti];
    for (int local_y = 0; local_y < ft->height; ++local_y) { /* Film_tile.iter, film_tile.ml:47-61 */
      int global_y = local_y + ft->tile.row - ft->border;
      for (int local_x = 0; local_x < ft->width; ++local_x) {
        int global_x = local_x + ft->tile.col - ft->border;
        if (0 <= global_x && global_x < width && 0 <= global_y && global_y < height) {
          const double* c = &ft->pixels[((size_t)local_y * ft->width + local_x) * 3];
          double* p = &rgb_out[((size_t)global_y * width + global_x) * 3];
          p[0] = c[0] + p[0]; p[1] = c[1] + p[1]; p[2] = c[2] + p[2];
        }
      }
    }
    free(ft->pixels);
  }
  /* gamma, integrator.ml:152-154 */
  double spp_inv = 1.0 / (double)spp;
  for (size_t i = 0; i < (size_t)width * height * 3; ++i) rgb_out[i] = sqrt(rgb_out[i] * spp_inv);
  if (counters_out) { counters_out[0] = job.ct.samples; counters_out[1] = job.ct.segments; counters_out[2] = job.ct.nodes_tested; counters_out[3] = job.ct.prims_tested; counters_out[4] = job.ct.floor_tested; }
  free(job.films); free(job.tiles);
  pthread_mutex_destroy(&job.mu);
  if (ms_out) *ms_out = now_ms() - t0;
  return 0;
}

/* parity aid: state after the first segment of each sample (mirror of ptx_debug_first_scatter) */
ORC_API void orc_debug_first_scatter(const orc_scene* sc, int width, int height, int spp, int max_bounces, int64_t n,
                                     const int32_t* xs, const int32_t* ys, const int32_t* passes, double* ray_out, double* attn_out, int32_t* alive_out, int32_t* info_out) {
  int dim = 2 + 2 * max_bounces;
  double* alpha = (double*)malloc(sizeof(double) * (size_t)dim);
  orc_lds_alpha(dim, alpha);
  double widthf = 1.0 / (double)width, heightf = 1.0 / (double)height;
  for (int64_t i = 0; i < n; ++i) {
    sampler_t smp; smp.alpha = alpha; smp.offset = (ys[i] * width) + xs[i] + (passes[i] * spp);
    double dx = sample_dim(&smp, 0), dy = sample_dim(&smp, 1);
    double cx = ((double)xs[i] + dx) * widthf, cy = 1.0 - (((double)ys[i] + dy) * heightf);
    ray_t ray = camera_ray(&sc->camera, cx, cy);
    alive_out[i] = 0;
    hit_t h; int prim = -1;
    if (info_out) { info_out[3 * i] = -1; info_out[3 * i + 1] = -1; info_out[3 * i + 2] = -1; }
    if (!scene_intersect(sc, &ray, &h, NULL, &prim, NULL)) continue;
    double u = sample_dim(&smp, 2), v = sample_dim(&smp, 3);
    scatter_t s = hit_scatter(&sc->mt, &h, u);
    if (info_out) { info_out[3 * i] = prim; info_out[3 * i + 1] = h.m->kind; info_out[3 * i + 2] = s.kind; }
    ray_t out; v3 att;
    if (s.kind == SC_ABSORB) continue;
    if (s.kind == SC_SPECULAR) { out = s.ray; att = s.attenuation; }
    else {
      v3 dir = unit_square_to_hemisphere(u, v);
      double pd = pdf_eval_diffuse(dir);
      if (pd == 0.0) continue;
      out = sspace_world_ray(&h.shader_space, dir);
      att = v3_scale(s.attenuation, pd / pd);
    }
    alive_out[i] = 1;
    ray_out[6 * i] = out.origin.x; ray_out[6 * i + 1] = out.origin.y; ray_out[6 * i + 2] = out.origin.z;
    ray_out[6 * i + 3] = out.direction.x; ray_out[6 * i + 4] = out.direction.y; ray_out[6 * i + 5] = out.direction.z;
    attn_out[3 * i] = att.x; attn_out[3 * i + 1] = att.y; attn_out[3 * i + 2] = att.z;
  }
  free(alpha);
}

/* ================================================================== progressive photon mapping
 * progressive-photon-map/src/progressive_photon_map.ml -- the integrator the reference uses for cornell-box and
 * ganesha.  No fixture of the reference pins this part: PARITY UNPINNED for PPM
 * (the restatement is checked only for internal consistency and against the GPU). */
typedef struct { sspace_t shader_space; v3 wi, flux; double radius; } photon_t; /* Photon.t, :112-130 */

typedef struct {
  int kind;
  v3 position, color;  /* color already scaled by power */
  sspace_t shader_space; /* spot */
} light_t;

static const double PPM_PI = 3.14159265358979323846;

static light_t light_create(const ptx_light* l) {
  light_t o; memset(&o, 0, sizeof o);
  o.kind = l->kind;
  o.position = v3_make(l->position[0], l->position[1], l->position[2]);
  o.color = v3_scale(v3_make(l->color[0], l->color[1], l->color[2]), l->power); /* Color.scale color power */
  if (l->kind == PTX_LIGHT_SPOT) /* Spot_light.create :91-95 */
    o.shader_space = sspace_create(v3_normalize(v3_make(l->direction[0], l->direction[1], l->direction[2])), o.position);
  return o;
}
static double light_power(const light_t* l) { return l->color.x + l->color.y + l->color.z; } /* :125-128 */

/* Light.random_ray :112-118 */
static ray_t light_random_ray(const light_t* l, double u, double v) {
  if (l->kind == PTX_LIGHT_POINT) { /* Point_light.random_direction :69-78 */
    double theta = 2.0 * PPM_PI * u;
    double phi = m_acos(1.0 - (2.0 * v));
    double sin_phi = m_sin(phi);
    v3 dir = v3_make(sin_phi * m_cos(theta), sin_phi * m_sin(theta), m_cos(phi));
    return ray_create(l->position, dir);
  }
  /* Spot_light.random_ray :97-105 ; angle = 0.5 * 45 * pi / 180 ; disk_radius = atan angle (:88-89) */
  double angle = 0.5 * 45.0 * PPM_PI / 180.0;
  double disk_radius = atan(angle);
  double r = disk_radius * sqrt(u);
  double theta = v * 2.0 * PPM_PI;
  double x = r * m_cos(theta), y = r * m_sin(theta), z = 1.0;
  return sspace_world_ray(&l->shader_space, v3_make(x, y, z));
}

typedef struct { photon_t* p; size_t n, cap; } photon_vec;
static void pv_push(photon_vec* v, photon_t ph) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (photon_t*)realloc(v->p, sizeof(photon_t) * v->cap); }
  v->p[v->n++] = ph;
}

/* Photon_map.trace_photon :196-233 ; deposits appended in chronological order (the reference conses: reversed later) */
static void trace_photon(const orc_scene* sc, const light_t* light, const double* alpha, int offset, int max_bounces, double radius,
                         photon_vec* out, int64_t* rays) {
  sampler_t smp; smp.alpha = alpha; smp.offset = offset;
  double u = sample_dim(&smp, 0), v = sample_dim(&smp, 1);
  ray_t ray = light_random_ray(light, u, v);
  v3 flux = light->color;
  int dim = 2;
  while (max_bounces > 0) {
    max_bounces = max_bounces - 1;
    u = sample_dim(&smp, dim); v = sample_dim(&smp, dim + 1); /* take_2d BEFORE the intersection test */
    dim = dim + 2;
    hit_t h;
    (*rays)++;
    if (!scene_intersect(sc, &ray, &h, NULL, NULL, NULL)) return;
    scatter_t s = hit_scatter(&sc->mt, &h, u);
    if (s.kind == SC_ABSORB) return;
    if (s.kind == SC_SPECULAR) { ray = s.ray; flux = v3_mul(flux, s.attenuation); continue; }
    v3 color = s.attenuation;
    flux = v3_mul(flux, color);
    photon_t ph; /* Photon.create ss ray flux ~radius :123-126 */
    ph.shader_space = h.shader_space;
    ph.wi = v3_normalize(v3_neg(ray.direction));
    ph.flux = flux;
    ph.radius = radius;
    pv_push(out, ph);
    double color_max = v3_max_coord(color);
    if (u <= color_max) {
      double cm_inv = 1.0 / color_max;
      flux = v3_scale(flux, cm_inv);
      double u2 = u * cm_inv;
      v3 dir = unit_square_to_hemisphere(u2, v);
      ray = sspace_world_ray(&h.shader_space, dir);
    } else return;
  }
}

typedef struct { node_t* root; photon_t* photons; size_t n; orc_scene stub; } photon_map_t;

/* Photon_map.create :235-256 */
static photon_map_t photon_map_create(const orc_scene* sc, const double* alpha, int base, double radius, int photon_count, int max_bounces,
                                      const light_t* lights, int n_lights, int64_t* rays) {
  double total_power = 0.0;
  for (int i = 0; i < n_lights; ++i) total_power = total_power + light_power(&lights[i]); /* List.sum: 0. + p1 + ... */
  photon_vec chron; memset(&chron, 0, sizeof chron);
  int offset = 0;
  for (int li = 0; li < n_lights; ++li) {
    double f = light_power(&lights[li]) / total_power;
    int count = (int)((double)photon_count * f); /* Int.of_float truncates */
    int start = offset;
    offset = start + count;
    for (int i = start; i < start + count; ++i) trace_photon(sc, &lights[li], alpha, i + base, max_bounces, radius, &chron, rays);
  }
  photon_map_t pm; memset(&pm, 0, sizeof pm);
  pm.n = chron.n;
  pm.photons = (photon_t*)malloc(sizeof(photon_t) * (chron.n + 1));
  for (size_t i = 0; i < chron.n; ++i) pm.photons[i] = chron.p[chron.n - 1 - i]; /* the consed list */
  free(chron.p);
  if (pm.n == 0) return pm; /* "BUG: no photons" */
  pm.stub.num_bins = 8; pm.stub.length_cutoff = 8; pm.stub.leaf_kind = PTX_LEAF_ARRAY;
  bshape_t* bs = (bshape_t*)calloc(pm.n, sizeof(bshape_t));
  for (size_t i = 0; i < pm.n; ++i) {
    v3 c = pm.photons[i].shader_space.origin, r = v3_make(radius, radius, radius);
    bs[i].shape.id = (int)i;
    bs[i].bbox.min = v3_sub(c, r); bs[i].bbox.max = v3_add(c, r);
    bs[i].centroid = bbox_center(bs[i].bbox);
  }
  bbox_t bbox = bs[0].bbox;
  for (size_t i = 1; i < pm.n; ++i) bbox = bbox_union(bbox, bs[i].bbox);
  pm.root = tree_build(&pm.stub, bbox, bs, (int)pm.n);
  free(bs);
  return pm;
}
static void photon_map_free(photon_map_t* pm) { tree_free(pm->root); free(pm->photons); }

typedef struct { int* idx; size_t n, cap; } int_vec;
/* Tree.fold_neighbors (shape_tree.ml:222-231) + Photon_map.fold_neighbors (:188-194) + the caller's filter (:341-347):
 * visit order lhs then rhs; accepted photons appended (the reference conses, so its list is this reversed) */
static void fold_neighbors_rec(const photon_map_t* pm, const node_t* t, v3 point, v3 hit_normal, int_vec* acc) {
  if (!bbox_mem(&t->bbox, point)) return;
  if (t->is_leaf) {
    for (int i = 0; i < t->n_elts; ++i) {
      const photon_t* p = &pm->photons[t->elts[i].id];
      v3 v = v3_sub(point, p->shader_space.origin);
      if (v3_quadrance(v) < p->radius * p->radius) {
        if (v3_dot(p->shader_space.normal, hit_normal) > 1e-3) {
          if (acc->n == acc->cap) { acc->cap = acc->cap ? acc->cap * 2 : 256; acc->idx = (int*)realloc(acc->idx, sizeof(int) * acc->cap); }
          acc->idx[acc->n++] = t->elts[i].id;
        }
      }
    }
    return;
  }
  fold_neighbors_rec(pm, t->lhs, point, hit_normal, acc);
  fold_neighbors_rec(pm, t->rhs, point, hit_normal, acc);
}

/* estimate_color :316-372 */
static v3 ppm_estimate(const orc_scene* sc, const photon_map_t* pm, const double* alpha, int offset, int x, int y, int width, int height,
                       int max_bounces, int_vec* scratch, int64_t* rays, int64_t* n_neighbors) {
  sampler_t smp; smp.alpha = alpha; smp.offset = offset;
  double inv_widthf = 1.0 / (double)width, inv_heightf = 1.0 / (double)height;
  double dx = sample_dim(&smp, 0), dy = sample_dim(&smp, 1);
  double cx = inv_widthf * (dx + (double)x), cy = inv_heightf * (dy + (double)y);
  ray_t ray = camera_ray(&sc->camera, cx, cy);
  v3 beta = v3_make(1.0, 1.0, 1.0);
  int dimension = 2;
  const v3 black = v3_make(0.0, 0.0, 0.0);
  for (;;) {
    if (max_bounces <= 0) return black;
    hit_t h;
    (*rays)++;
    if (!scene_intersect(sc, &ray, &h, NULL, NULL, NULL)) return black;
    max_bounces = max_bounces - 1;
    double u = sample_dim(&smp, dimension);
    scatter_t s = hit_scatter(&sc->mt, &h, u);
    if (s.kind == SC_ABSORB) return black;
    if (s.kind == SC_SPECULAR) { beta = v3_mul(s.attenuation, beta); ray = s.ray; dimension = dimension + 1; continue; }
    beta = v3_mul(s.attenuation, beta);
    v3 hit_point = h.shader_space.origin, hit_normal = h.shader_space.normal;
    scratch->n = 0;
    if (pm->root) fold_neighbors_rec(pm, pm->root, hit_point, hit_normal, scratch);
    if (scratch->n == 0) return black;
    *n_neighbors += (int64_t)scratch->n;
    const double k = 1.0;
    double normalizer = 1.0 - (2.0 / (3.0 * k));
    double radius = pm->photons[scratch->idx[scratch->n - 1]].radius; /* hd of the consed list = last accepted */
    double area = PPM_PI * (radius * radius);
    v3 flux = black;
    for (size_t j = scratch->n; j-- > 0;) { /* List.fold over the consed list = reverse visit order */
      const photon_t* p = &pm->photons[scratch->idx[j]];
      double distance = sqrt(v3_quadrance(v3_sub(p->shader_space.origin, hit_point)));
      double w = (1.0 - (distance / (k * radius))) / 1.0; /* weight / pdf, pdf = 1.0 */
      flux = v3_add(flux, v3_scale(p->flux, w));
    }
    return v3_scale(v3_mul(beta, flux), 1.0 / (area * normalizer));
  }
}

/* radius2 / radius :381-392 */
static double ppm_radius2(int i, double alpha, double init_radius2) {
  double product = 1.0;
  for (int k = 1; k <= i - 1; ++k) { double kf = (double)k; product = product * (kf + alpha) / kf; }
  return product * init_radius2 / (double)i;
}

/* Make(Scene).go :420-451 without the gamma / PNG step; img_sum_out W*H*3 (row 0 = top) */
ORC_API int orc_ppm_render(const orc_scene* sc, const ptx_ppm_params* p, const ptx_light* lights_in, int n_lights, double* img_sum_out,
                           int64_t* stats_out /* photons_stored, photon_rays, eye_rays, neighbors */, double* radius_out) {
  int width = p->width, height = p->height, max_bounces = p->max_bounces;
  if (!sc->root || n_lights <= 0) return -1;
  light_t* lights = (light_t*)malloc(sizeof(light_t) * (size_t)n_lights);
  for (int i = 0; i < n_lights; ++i) lights[i] = light_create(&lights_in[i]);
  int pdim = 2 + 2 * max_bounces, edim = 2 + max_bounces;
  double* p_alpha = (double*)malloc(sizeof(double) * (size_t)pdim);
  double* e_alpha = (double*)malloc(sizeof(double) * (size_t)edim);
  orc_lds_alpha(pdim, p_alpha);
  orc_lds_alpha(edim, e_alpha);
  /* init_radius2 :292-297 */
  bbox_t bb = sc->root->bbox;
  v3 ext = v3_sub(bb.max, bb.min);
  double a = (ext.x + ext.y + ext.z) / 3.0;
  double b = (double)(width + height) / (double)2;
  double init_radius2 = (a / b) * (a / b);
  double inv_photon_count = 1.0 / (double)p->photon_count;
  memset(img_sum_out, 0, sizeof(double) * (size_t)width * height * 3);
  int64_t st[4] = {0, 0, 0, 0};
  int_vec scratch; memset(&scratch, 0, sizeof scratch);
  double radius = 0.0;
  for (int it = 0; it < p->iterations; ++it) {
    radius = sqrt(ppm_radius2(it + 1, p->alpha, init_radius2));
    photon_map_t pm = photon_map_create(sc, p_alpha, it * p->photon_count, radius, p->photon_count, max_bounces, lights, n_lights, &st[1]);
    if (pm.n == 0) { photon_map_free(&pm); free(lights); free(p_alpha); free(e_alpha); free(scratch.idx); return -2; }
    st[0] += (int64_t)pm.n;
    int eye_sample_base = it * width * height;
    for (int pixel = 0; pixel < width * height; ++pixel) { /* render_image :374-381 */
      int x = pixel % width, y = pixel / width;
      v3 c = ppm_estimate(sc, &pm, e_alpha, pixel + eye_sample_base, x, y, width, height, max_bounces, &scratch, &st[2], &st[3]);
      v3 color = v3_scale(c, inv_photon_count);
      int yy = height - 1 - y; /* write_pixel :305-313 */
      double* px = &img_sum_out[((size_t)yy * width + x) * 3];
      px[0] = color.x + px[0]; px[1] = color.y + px[1]; px[2] = color.z + px[2];
    }
    photon_map_free(&pm);
  }
  if (stats_out) memcpy(stats_out, st, sizeof st);
  if (radius_out) *radius_out = radius;
  free(lights); free(p_alpha); free(e_alpha); free(scratch.idx);
  return 0;
}

/* lights of the two reference scenes, in camera space */
ORC_API int orc_lights_cornell(int width, int height, ptx_light* out) { /* cornell-box/bin/main.ml:183,225-228 */
  const double pi = 3.14159265358979323846;
  double fov = (2.0 * atan(0.5)) * 180.0 / pi;
  camera_t cam = camera_create(v3_make(0.5, 0.5, -1.0), v3_make(0.5, 0.5, 0.0), v3_make(0.0, 1.0, 0.0), (double)width / (double)height, fov);
  v3 pos = camera_transform(&cam, v3_make(0.5, 0.82, 0.5));
  memset(out, 0, sizeof *out);
  out->kind = PTX_LIGHT_POINT; out->position[0] = pos.x; out->position[1] = pos.y; out->position[2] = pos.z;
  out->color[0] = out->color[1] = out->color[2] = 1.0; out->power = 2.0;
  return 1;
}
ORC_API int orc_lights_ganesha(const orc_scene* sc, ptx_light* out) { /* ganesha/bin/main.ml:267-282 */
  bbox_t bbox = sc->root->bbox;
  v3 center = bbox_center(bbox);
  v3 v = v3_sub(bbox.max, center);
  v3 position = v3_add(bbox.max, v3_add(v3_scale(v, 3.0), v3_scale(v3_make(0.0, 0.0, 1.0), -400.0)));
  v3 direction = v3_sub(center, position);
  memset(out, 0, 2 * sizeof *out);
  out[0].kind = PTX_LIGHT_SPOT; out[0].power = 10000.0;
  out[0].position[0] = position.x; out[0].position[1] = position.y; out[0].position[2] = position.z;
  out[0].direction[0] = direction.x; out[0].direction[1] = direction.y; out[0].direction[2] = direction.z;
  out[0].color[0] = out[0].color[1] = out[0].color[2] = 1.0;
  out[1].kind = PTX_LIGHT_SPOT; out[1].power = 3000.0;
  out[1].position[2] = 1.0; out[1].direction[2] = -1.0; /* ~-V3.unit_z */
  out[1].direction[0] = -0.0; out[1].direction[1] = -0.0;
  out[1].color[0] = out[1].color[1] = out[1].color[2] = 1.0;
  return 2;
}

ORC_API int orc_abi_version(void) { return PTX_ABI_VERSION; }
