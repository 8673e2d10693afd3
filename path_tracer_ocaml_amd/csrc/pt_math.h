/* pt_math.h -- IEEE binary64 math shared by the HIP kernels and the host code.
 *
 * The reference (dalev/path-tracer-ocaml) is binary64 end to end and calls the
 * platform libm for hypot / sin / cos / acos / atan2 / ( ** 5.0 ):
 *   path_tracer/src/affine.ml:65-68      (V3.normalize -> Float.hypot)
 *   path_tracer/src/quaternion.ml:11-15  (Quaternion.normalize -> Float.hypot x3)
 *   path_tracer/src/shader_space.ml:56-64 (cos / sin / sqrt)
 *   sphere/src/sphere.ml:25-33           (acos / atan2)
 *   path_tracer/src/material.ml:16-20,37 (( ** ) 5.0)
 * A libm is not correctly rounded and differs between platforms by an ulp, and
 * a single ulp can flip a branch (checker parity, Schlick-vs-u) which moves a
 * pixel by 1/spp.  To make "GPU == CPU" provable rather than probable, every
 * such function is written ONCE here from +,-,*,/,sqrt,fma and integer bit
 * operations only (all correctly rounded by IEEE-754 on x86-64 and on gfx950),
 * and is compiled with contraction OFF on both sides.  The same source, the same
 * operation order => bit-identical results on host and device.
 *
 * Accuracy (measured against mpmath in tests/test_math.py): every function is
 * < 1 ulp on the domain the path tracer uses; pt_pow5 is correctly rounded
 * except in astronomically rare near-tie cases (double-double product).
 *
 * Polynomial coefficients and the pi/2 split are the public-domain fdlibm
 * constants (Sun Microsystems, 1993); the code structure is our own.
 */
#ifndef PT_MATH_H
#define PT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD static __host__ __device__ __forceinline__
#else
#define PT_HD static inline
#endif

/* ---- primitives (each maps to one IEEE operation on both targets) ---- */
PT_HD double pt_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
PT_HD double pt_sqrt(double x) { return __builtin_sqrt(x); }
PT_HD double pt_fabs(double x) { return __builtin_fabs(x); }
PT_HD double pt_trunc(double x) { return __builtin_trunc(x); }
PT_HD double pt_rint(double x) { return __builtin_rint(x); }

PT_HD uint64_t pt_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
PT_HD double pt_from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}
PT_HD int pt_signbit(double x) { return (int)(pt_bits(x) >> 63); }
PT_HD int pt_isnan(double x) { return x != x; }
PT_HD int pt_isfinite(double x) {
  return ((pt_bits(x) >> 52) & 0x7ff) != 0x7ff;
}
PT_HD double pt_nan(void) { return pt_from_bits(0x7ff8000000000000ULL); }
PT_HD double pt_inf(void) { return pt_from_bits(0x7ff0000000000000ULL); }

/* Base.Float.min / max: NaN-propagating (used by Bbox.hit_range, bbox.ml:46-49;
 * V3.min_coord/max_coord, affine.ml:56-57; Shader_space.refract, shader_space.ml:43). */
PT_HD double pt_base_min(double x, double y) {
  if (x != x || y != y) return pt_nan();
  return x < y ? x : y;
}
PT_HD double pt_base_max(double x, double y) {
  if (x != x || y != y) return pt_nan();
  return x > y ? x : y;
}

/* ---- hypot: sqrt(x^2+y^2) with one fused product, scaled against overflow ---- */
PT_HD double pt_hypot(double x, double y) {
  double ax = pt_fabs(x), ay = pt_fabs(y);
  uint64_t bx = pt_bits(ax), by = pt_bits(ay);
  /* IEEE: hypot(inf, anything) = inf, even NaN */
  if (bx == 0x7ff0000000000000ULL || by == 0x7ff0000000000000ULL) return pt_inf();
  if (ax != ax || ay != ay) return pt_nan();
  if (ax < ay) {
    double t = ax;
    ax = ay;
    ay = t;
  }
  if (ay == 0.0) return ax;
  /* ax >= ay > 0 */
  double scale = 1.0;
  if (ax > 0x1p+510) {
    ax *= 0x1p-600;
    ay *= 0x1p-600;
    scale = 0x1p+600;
  } else if (ay < 0x1p-450) {
    ax *= 0x1p+600;
    ay *= 0x1p+600;
    scale = 0x1p-600;
  }
  /* when ay is within a factor 2 of ax, (2ay)ax + (ax-ay)^2 loses fewer bits */
  double t1 = ay + ay;
  double t2 = ax - ay;
  double h;
  if (t1 >= ax)
    h = pt_sqrt(pt_fma(t1, ax, t2 * t2));
  else
    h = pt_sqrt(pt_fma(ax, ax, ay * ay));
  return h * scale;
}

/* ---- sin / cos ---- */
#define PT_S1 (-1.66666666666666324348e-01)
#define PT_S2 (8.33333333332248946124e-03)
#define PT_S3 (-1.98412698298579493134e-04)
#define PT_S4 (2.75573137070700676789e-06)
#define PT_S5 (-2.50507602534068634195e-08)
#define PT_S6 (1.58969099521155010221e-10)
#define PT_C1 (4.16666666666666019037e-02)
#define PT_C2 (-1.38888888888741095749e-03)
#define PT_C3 (2.48015872894767294178e-05)
#define PT_C4 (-2.75573143513906633035e-07)
#define PT_C5 (2.08757232129817482790e-09)
#define PT_C6 (-1.13596475577881948265e-11)

/* sin on [-pi/4, pi/4] of x + y (y = tail of the reduced argument) */
PT_HD double pt_k_sin(double x, double y) {
  double z = x * x;
  double v = z * x;
  double r = PT_S2 + z * (PT_S3 + z * (PT_S4 + z * (PT_S5 + z * PT_S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * PT_S1);
}
/* cos on [-pi/4, pi/4] of x + y */
PT_HD double pt_k_cos(double x, double y) {
  double z = x * x;
  double w = z * z;
  double r = z * (PT_C1 + z * (PT_C2 + z * PT_C3)) + (w * w) * (PT_C4 + z * (PT_C5 + z * PT_C6));
  double hz = 0.5 * z;
  w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

/* Cody-Waite reduction of x by pi/2 in three 33-bit pieces: x = n*pi/2 + y0 + y1,
 * |y0| <= pi/4 (+ slop).  Accurate for |x| < 2^20 * pi/2; beyond that the result
 * is still deterministic but loses accuracy (the tracer only passes [0, 2pi)). */
PT_HD int pt_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;
  const double pio2_2 = 6.07710050630396597660e-11;
  const double pio2_2t = 2.02226624879595063154e-21;
  const double pio2_3 = 2.02226624871116645580e-21;
  const double pio2_3t = 8.47842766036889956997e-32;
  double fn = pt_rint(x * invpio2);
  double r = x - fn * pio2_1; /* exact for |fn| < 2^20 */
  double t = r;
  double w = fn * pio2_2;
  r = t - w;
  w = fn * pio2_2t - ((t - r) - w);
  double a = r - w;
  /* heavy cancellation (x within ~2^-49 relative of a multiple of pi/2): third piece */
  int ea = (int)((pt_bits(a) >> 52) & 0x7ff);
  int ex = (int)((pt_bits(x) >> 52) & 0x7ff);
  if (ex - ea > 49) {
    t = r;
    w = fn * pio2_3;
    r = t - w;
    w = fn * pio2_3t - ((t - r) - w);
    a = r - w;
  }
  *y0 = a;
  *y1 = (r - a) - w;
  /* fn may be huge for silly inputs; keep the low two bits meaningful where it fits */
  double q = fn - 4.0 * pt_trunc(fn * 0.25);
  int n = (int)q;
  return n & 3;
}

PT_HD void pt_sincos(double x, double* s, double* c) {
  if (!pt_isfinite(x)) {
    *s = pt_nan();
    *c = pt_nan();
    return;
  }
  double y0, y1;
  int n = 0;
  if (pt_fabs(x) <= 0.78539816339744827900) {
    y0 = x;
    y1 = 0.0;
  } else {
    n = pt_rem_pio2(x, &y0, &y1);
  }
  double ks = pt_k_sin(y0, y1);
  double kc = pt_k_cos(y0, y1);
  double sv = (n & 1) ? kc : ks;
  double cv = (n & 1) ? ks : kc;
  if (n & 2) sv = -sv;
  if ((n + 1) & 2) cv = -cv;
  *s = sv;
  *c = cv;
}
PT_HD double pt_sin(double x) {
  double s, c;
  pt_sincos(x, &s, &c);
  return s;
}
PT_HD double pt_cos(double x) {
  double s, c;
  pt_sincos(x, &s, &c);
  return c;
}

/* ---- acos ---- */
PT_HD double pt_acos_r(double z) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  return p / q;
}
PT_HD double pt_acos(double x) {
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  const double pi = 3.14159265358979311600e+00;
  if (x != x) return pt_nan();
  double ax = pt_fabs(x);
  if (ax >= 1.0) {
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return pt_nan();
  }
  if (ax < 0.5) {
    if (ax <= 0x1p-57) return pio2_hi + pio2_lo;
    double z = x * x;
    double r = pt_acos_r(z);
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (x < 0.0) {
    double z = (1.0 + x) * 0.5;
    double s = pt_sqrt(z);
    double r = pt_acos_r(z);
    double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  double z = (1.0 - x) * 0.5;
  double s = pt_sqrt(z);
  double df = pt_from_bits(pt_bits(s) & 0xffffffff00000000ULL);
  double c = (z - df * df) / (s + df);
  double r = pt_acos_r(z);
  double w = r * s + c;
  return 2.0 * (df + w);
}

/* ---- atan / atan2 ---- */
PT_HD double pt_atan(double x) {
  const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01,
               hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
  const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17,
               lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
               a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
               a10 = 1.62858201153657823623e-02;
  if (x != x) return pt_nan();
  int neg = pt_signbit(x);
  double ax = pt_fabs(x);
  double hi, lo;
  int id;
  if (ax >= 0x1p+66) {
    double z = hi3 + lo3;
    return neg ? -z : z;
  }
  if (ax < 0.4375) {
    if (ax < 0x1p-27) return x;
    id = -1;
    hi = 0.0;
    lo = 0.0;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) {
      id = 0;
      hi = hi0;
      lo = lo0;
      ax = (2.0 * ax - 1.0) / (2.0 + ax);
    } else {
      id = 1;
      hi = hi1;
      lo = lo1;
      ax = (ax - 1.0) / (ax + 1.0);
    }
  } else if (ax < 2.4375) {
    id = 2;
    hi = hi2;
    lo = lo2;
    ax = (ax - 1.5) / (1.0 + 1.5 * ax);
  } else {
    id = 3;
    hi = hi3;
    lo = lo3;
    ax = -1.0 / ax;
  }
  double z = ax * ax;
  double w = z * z;
  double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0) {
    double r = ax - ax * (s1 + s2);
    return neg ? -r : r;
  }
  double r = hi - ((ax * (s1 + s2) - lo) - ax);
  return neg ? -r : r;
}

/* The constants of atan2's special cases (zeros, infinities, huge ratios) are only ever needed in branches that almost
 * never run, but as plain literals the device compiler materialises all of them ahead of the shade kernel's main loop and
 * keeps them in registers across it -- six 64-bit values the 128-VGPR kernel then spills.  Behind an empty volatile asm a
 * value is made where it is used.  Same bits either way. */
#define PT_LIT_PI 3.1415926535897931160E+00
#define PT_LIT_PI_LO 1.2246467991473531772E-16
#define PT_LIT_PI_O_2 1.5707963267948965580E+00
#define PT_LIT_PI_O_4 7.8539816339744827900E-01
#if defined(__HIP_DEVICE_COMPILE__)
template <unsigned long long BITS>
__device__ __forceinline__ double pt_rare_const() { /* the two v_mov of a 64-bit literal, pinned to the place of use */
  unsigned lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "i"((unsigned)(BITS & 0xffffffffull)), "i"((unsigned)(BITS >> 32)));
  return __hiloint2double((int)hi, (int)lo);
}
#define PT_RARE_CONST(x) pt_rare_const<__builtin_bit_cast(unsigned long long, (double)(x))>() /* x: a literal expression */
#else
#define PT_RARE_CONST(x) (x)
#endif
PT_HD double pt_atan2(double y, double x) {
  const double pi = PT_LIT_PI;
  const double pi_lo = PT_LIT_PI_LO;
  if (x != x || y != y) return pt_nan();
  if (x == 1.0) return pt_atan(y);
  int m = pt_signbit(y) | (pt_signbit(x) << 1);
  if (y == 0.0) {
    if (m == 0 || m == 1) return y;
    return (m == 2) ? PT_RARE_CONST(PT_LIT_PI) : PT_RARE_CONST(-PT_LIT_PI);
  }
  if (x == 0.0) return pt_signbit(y) ? PT_RARE_CONST(-PT_LIT_PI_O_2) : PT_RARE_CONST(PT_LIT_PI_O_2);
  int xinf = !pt_isfinite(x), yinf = !pt_isfinite(y);
  if (xinf) {
    if (yinf) {
      switch (m) {
        case 0: return PT_RARE_CONST(PT_LIT_PI_O_4);
        case 1: return PT_RARE_CONST(-PT_LIT_PI_O_4);
        case 2: return PT_RARE_CONST(3.0 * PT_LIT_PI_O_4);
        default: return PT_RARE_CONST(-3.0 * PT_LIT_PI_O_4);
      }
    }
    switch (m) {
      case 0: return 0.0;
      case 1: return -0.0;
      case 2: return PT_RARE_CONST(PT_LIT_PI);
      default: return PT_RARE_CONST(-PT_LIT_PI);
    }
  }
  if (yinf) return pt_signbit(y) ? PT_RARE_CONST(-PT_LIT_PI_O_2) : PT_RARE_CONST(PT_LIT_PI_O_2);
  int ey = (int)((pt_bits(y) >> 52) & 0x7ff);
  int ex = (int)((pt_bits(x) >> 52) & 0x7ff);
  int k = ey - ex;
  double z;
  if (k > 60)
    z = PT_RARE_CONST(PT_LIT_PI_O_2 + 0.5 * PT_LIT_PI_LO);
  else if ((m & 2) && k < -60)
    z = 0.0;
  else
    z = pt_atan(pt_fabs(y / x));
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

/* ---- x ** 5.0 : x^5 by double-double products (odd power keeps the sign) ---- */
PT_HD double pt_pow5(double x) {
  if (x != x) return pt_nan();
  if (!pt_isfinite(x)) return x;
  /* x^2 = p1 + e1 exactly */
  double p1 = x * x;
  double e1 = pt_fma(x, x, -p1);
  /* x^4 ~ p2 + e2 */
  double p2 = p1 * p1;
  double e2 = pt_fma(p1, p1, -p2);
  e2 = pt_fma(p1 + p1, e1, e2);
  /* x^5 ~ p3 + e3 */
  double p3 = p2 * x;
  if (!pt_isfinite(p3)) return p3;
  double e3 = pt_fma(p2, x, -p3);
  e3 = pt_fma(e2, x, e3);
  return p3 + e3;
}

#endif /* PT_MATH_H */
