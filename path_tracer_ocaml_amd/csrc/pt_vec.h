/* pt_vec.h -- binary64 3-vectors with the reference's exact operation order.
 *
 * Mirrors path_tracer/src/affine.ml (V3 / P3): dot and cross use EXPLICIT fused
 * multiply-adds exactly where the reference writes Stdlib.Float.fma (affine.ml:53,60,70-73);
 * everything else is one IEEE operation per OCaml operator.  Compiled with
 * -ffp-contract=off on host and device, so nothing else is ever fused.
 */
#ifndef PT_VEC_H
#define PT_VEC_H

#include "pt_math.h"

struct V3 {
  double x, y, z;
};

PT_HD V3 v3(double x, double y, double z) {
  V3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
PT_HD V3 v3_add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }  /* affine.ml:45 */
PT_HD V3 v3_sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }  /* :46 */
PT_HD V3 v3_mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }  /* :47 */
PT_HD V3 v3_neg(V3 a) { return v3(-a.x, -a.y, -a.z); }                       /* :49 */
/* V3.fma u v w = u*v + w, fused per component (affine.ml:53) */
PT_HD V3 v3_fma(V3 u, V3 v, V3 w) {
  return v3(pt_fma(u.x, v.x, w.x), pt_fma(u.y, v.y, w.y), pt_fma(u.z, v.z, w.z));
}
/* V3.dot (affine.ml:60): fma v.x w.x (fma v.y w.y (v.z * w.z)) */
PT_HD double v3_dot(V3 v, V3 w) { return pt_fma(v.x, w.x, pt_fma(v.y, w.y, v.z * w.z)); }
/* V3.scale v s = map (( *. ) s) (affine.ml:61): s on the left */
PT_HD V3 v3_scale(V3 v, double s) { return v3(s * v.x, s * v.y, s * v.z); }
PT_HD double v3_quadrance(V3 v) { return v3_dot(v, v); }
/* V3.lerp t v w (affine.ml:63) */
PT_HD V3 v3_lerp(double t, V3 v, V3 w) { return v3_add(v3_scale(v, 1.0 - t), v3_scale(w, t)); }
/* V3.normalize (affine.ml:65-68): scale v (1 / hypot x (hypot y z)) */
PT_HD V3 v3_normalize(V3 v) {
  return v3_scale(v, pt_rnorm3(v.x, v.y, v.z));
}
/* V3.cross (affine.ml:70-73): h w x y z = fma w x (-(y*z)) */
PT_HD double v3_cross_h(double w, double x, double y, double z) { return pt_fma(w, x, -(y * z)); }
PT_HD V3 v3_cross(V3 p, V3 q) {
  return v3(v3_cross_h(p.y, q.z, p.z, q.y), v3_cross_h(p.z, q.x, p.x, q.z), v3_cross_h(p.x, q.y, p.y, q.x));
}
PT_HD double v3_axis(V3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

/* Bbox.t (bbox.ml:3-6) */
struct Box {
  V3 mn, mx;
};
/* Bbox.union (bbox.ml:14-18): Base Float.min / max per component */
PT_HD Box box_union(Box a, Box b) {
  Box r;
  r.mn = v3(pt_base_min(a.mn.x, b.mn.x), pt_base_min(a.mn.y, b.mn.y), pt_base_min(a.mn.z, b.mn.z));
  r.mx = v3(pt_base_max(a.mx.x, b.mx.x), pt_base_max(a.mx.y, b.mx.y), pt_base_max(a.mx.z, b.mx.z));
  return r;
}
/* Bbox.center (bbox.ml:12) */
PT_HD V3 box_center(Box b) { return v3_scale(v3_add(b.mn, b.mx), 0.5); }
/* Bbox.surface_area (bbox.ml:33-38) */
PT_HD double box_surface_area(Box b) {
  V3 d = v3_sub(b.mx, b.mn);
  double a = pt_fma(d.x, d.y, pt_fma(d.y, d.z, d.z * d.x));
  return 2.0 * a;
}

#endif /* PT_VEC_H */
