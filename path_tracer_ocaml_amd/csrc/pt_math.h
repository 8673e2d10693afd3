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
 *
 * Shape of the code (round 4).  The shading stage issues these functions for every surface hit, and on the device a
 * divergent branch costs both of its sides plus the scalar bookkeeping around it.  So every function has ONE straight-line
 * main path that covers the whole domain the tracer can reach (finite, "mid-range" magnitudes), written with explicit
 * fused multiply-adds (Horner steps, exact residuals), and ONE rarely taken branch to a general version for everything
 * else (zeros, infinities, NaNs, magnitudes near the ends of the exponent range).  Inside a main path the device uses the
 * bare refinement sequences of sqrt / reciprocal / division (pt_sqrt_mid, pt_rcp_mid, pt_div_mid: what the compiler emits
 * for the IEEE operation minus the operand scaling and the special-value fix-up, which mid-range operands never need);
 * the host uses the IEEE operation itself.  Same bits (tests/test_gpu_parity.py::test_math_device_equals_host_bitwise).
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
#define PT_LIKELY(c) __builtin_expect(!!(c), 1)
#define PT_UNLIKELY(c) __builtin_expect(!!(c), 0)
/* A polynomial coefficient.  Device: pinned to a scalar register pair at its place of use.  A 64-bit literal cannot be an
 * operand of a binary64 instruction; left to itself the compiler copies every ADDEND of a Horner step into a vector register
 * pair first (v_mov_b32 x 2 + v_fmac_f64: three vector instructions and two live registers per step); from a scalar pair the
 * step is one v_fma_f64, and the two s_mov_b32 that make the pair issue beside other waves' vector work. */
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD double pt_sconst(double c) {
  asm volatile("" : "+s"(c));
  return c;
}
#define PT_K(c) pt_sconst(c)
#else
#define PT_K(c) (c)
#endif

/* ---- sqrt / reciprocal / division of MID-RANGE operands ----
 * Preconditions (the callers establish them with one range test per main path): every operand and the result are normal
 * numbers with exponents well inside the range -- |x| in [2^-760, 2^760] for sqrt; denominator and quotient in
 * [2^-700, 2^700] and |numerator| >= 2^-900 for the other two (a zero numerator is fine, and so is ANY numerator over a
 * denominator of exactly 1: the reciprocal and every step are then exact).  On the device these are the refinement sequences the compiler itself emits
 * for the IEEE operations (v_rsq_f64 / v_rcp_f64 seed, the same fused steps in the same order) WITHOUT the v_div_scale /
 * v_ldexp operand scaling and the v_div_fixup / v_cmp_class special-value selects around them: for mid-range operands those
 * are identities, so the result is the correctly rounded one, bit for bit what the host's sqrtsd / divsd give.
 * 10 instead of 20 vector instructions per sqrt, 7 instead of 11 per reciprocal, 8 instead of 12 per division. */
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD double pt_sqrt_mid(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
PT_HD double pt_rcp_mid(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0); /* the quotient estimate of 1 / x is r itself (1.0 * r) */
  return __builtin_fma(e, r, r);
}
PT_HD double pt_div_mid(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = n * r;
  e = __builtin_fma(-d, q, n);
  return __builtin_fma(e, r, q);
}
/* max / min of two non-negative, non-NaN values (one instruction each; the callers have excluded NaNs) */
PT_HD double pt_max_pos(double a, double b) { return __builtin_fmax(a, b); }
PT_HD double pt_min_pos(double a, double b) { return __builtin_fmin(a, b); }
#else
PT_HD double pt_sqrt_mid(double x) { return __builtin_sqrt(x); }
PT_HD double pt_rcp_mid(double x) { return 1.0 / x; }
PT_HD double pt_div_mid(double n, double d) { return n / d; }
PT_HD double pt_max_pos(double a, double b) { return a < b ? b : a; }
PT_HD double pt_min_pos(double a, double b) { return a < b ? a : b; }
#endif
/* sqrt of a value that is almost always mid-range (a sample in (0, 1), a discriminant): one test, then the bare sequence */
PT_HD double pt_sqrt_nonneg(double x) {
  if (PT_LIKELY(x >= 0x1p-700 && x <= 0x1p+700)) return pt_sqrt_mid(x);
  return __builtin_sqrt(x);
}

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

/* ---- hypot: sqrt(x^2+y^2) with one fused product ----
 * pt_hypot_core: mx >= mn >= 0, mx in [2^-380, 2^500].  When mn is within a factor 2 of mx, (2 mn) mx + (mx - mn)^2 loses
 * fewer bits than mx^2 + mn^2 (Borges, "An improved algorithm for hypot(a,b)").  mn = 0 gives sqrt(RN(mx^2)) = mx exactly. */
PT_HD double pt_hypot_core(double mx, double mn) {
  const double t1 = mn + mn;
  const double t2 = mx - mn;
  const int near = t1 >= mx;
  const double a = near ? t1 : mx;
  const double b = near ? t2 : mn;
  return pt_sqrt_mid(pt_fma(a, mx, b * b));
}
/* What the main paths below do not take: infinities, NaNs, all-zero operands, magnitudes near the ends of the exponent
 * range.  The special values are answered here (*special = the value of the hypot); anything else is finite and non-zero and
 * is brought to the middle of the range by a power of two (exact; an operand this pushes below the range was too small to
 * matter), so that the caller continues ON ITS MAIN PATH with the scaled operands and multiplies the scale back in.
 * Returns 0 when *special is the answer, else the scale (operands were multiplied by it). */
PT_HD double pt_hypot_rescale(double* a, double* b, double* c, double* special) {
  const double ia = *a, ib = *b, ic = *c; /* non-negative or NaN */
  if (pt_bits(ia) == 0x7ff0000000000000ULL || pt_bits(ib) == 0x7ff0000000000000ULL || pt_bits(ic) == 0x7ff0000000000000ULL) {
    *special = pt_inf(); /* IEEE: hypot(inf, anything) = inf, even NaN */
    return 0.0;
  }
  if (ia != ia || ib != ib || ic != ic) {
    *special = pt_nan();
    return 0.0;
  }
  double m = ia < ib ? ib : ia;
  m = m < ic ? ic : m;
  if (m == 0.0) {
    *special = 0.0;
    return 0.0;
  }
  const double k = m > 1.0 ? 0x1p-600 : 0x1p+700; /* m > 2^298 -> [2^-302, 2^424];  m < 2^-298 (denormals too) -> [2^-374, 2^402] */
  *a = ia * k;
  *b = ib * k;
  *c = ic * k;
  return k;
}
PT_HD double pt_hypot(double x, double y) {
  double ax = pt_fabs(x), ay = pt_fabs(y);
  double unscale = 1.0;
  if (PT_UNLIKELY(!((ax + ay) <= 0x1p+300 && (ax + ay) >= 0x1p-300))) { /* (a NaN fails the test) */
    double zero = 0.0, special;
    const double k = pt_hypot_rescale(&ax, &ay, &zero, &special);
    if (k == 0.0) return special;
    unscale = k > 1.0 ? 0x1p-700 : 0x1p+600;
  }
  const int sw = ax < ay;
  const double mx = sw ? ay : ax, mn = sw ? ax : ay;
  return pt_hypot_core(mx, mn) * unscale;
}
/* 1 / hypot x (hypot y z): the scalar of V3.normalize (affine.ml:65-68), one range test for the whole expression.
 * An inner hypot far below the range cannot change the outer one (x then carries the whole sum), so it is replaced by
 * its larger operand instead of leaving the main path -- axis-aligned normals (y = z = 0) stay on it.  Equal, bit for bit, to
 * the nested expression evaluated call by call through pt_hypot for every operand triple whose inner hypot is a normal number
 * (tests/test_math.py); with SUBNORMAL operands the nested form rounds the inner hypot to the subnormal grid first. */
PT_HD double pt_rnorm3(double x, double y, double z) {
  double ax = pt_fabs(x), ay = pt_fabs(y), az = pt_fabs(z);
  const double sum = (ax + ay) + az;
  double k = 1.0;
  if (PT_UNLIKELY(!(sum >= 0x1p-298 && sum <= 0x1p+298))) {
    double special;
    k = pt_hypot_rescale(&ax, &ay, &az, &special);
    if (k == 0.0) return 1.0 / special;
  }
  const double m1 = pt_max_pos(ay, az), n1 = pt_min_pos(ay, az);
  double h1 = pt_hypot_core(m1, n1);
  h1 = m1 >= 0x1p-380 ? h1 : m1;
  const double m2 = pt_max_pos(ax, h1), n2 = pt_min_pos(ax, h1);
  return pt_rcp_mid(pt_hypot_core(m2, n2)) * k; /* 1 / (h / k) */
}
/* 1 / hypot (hypot r x) (hypot y 0): the scalar of Quaternion.normalize (quaternion.ml:11-15) for the frame quaternion
 * (1 + n.z; (n.y, -n.x, 0)) of Shader_space.create (shader_space.ml:11-23); hypot y 0 = |y| */
PT_HD double pt_rnorm_frame(double r, double x, double y) {
  double ar = pt_fabs(r), ax = pt_fabs(x), ay = pt_fabs(y);
  const double sum = (ar + ax) + ay;
  double k = 1.0;
  if (PT_UNLIKELY(!(sum >= 0x1p-298 && sum <= 0x1p+298))) {
    double special;
    k = pt_hypot_rescale(&ar, &ax, &ay, &special);
    if (k == 0.0) return 1.0 / special;
  }
  const double m1 = pt_max_pos(ar, ax), n1 = pt_min_pos(ar, ax);
  double h1 = pt_hypot_core(m1, n1);
  h1 = m1 >= 0x1p-380 ? h1 : m1;
  const double m2 = pt_max_pos(h1, ay), n2 = pt_min_pos(h1, ay);
  return pt_rcp_mid(pt_hypot_core(m2, n2)) * k;
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

/* sin on [-pi/4, pi/4] of x + y (y = tail of the reduced argument); Horner steps fused */
PT_HD double pt_k_sin(double x, double y) {
  const double z = x * x;
  const double v = z * x;
  const double r = pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, PT_K(PT_S6), PT_K(PT_S5)), PT_K(PT_S4)), PT_K(PT_S3)), PT_K(PT_S2));
  /* x - ((z * (y/2 - v r) - y) - v S1) */
  const double t = pt_fma(-v, r, 0.5 * y);
  const double u = pt_fma(z, t, -y);
  return x - pt_fma(-v, PT_K(PT_S1), u);
}
/* cos on [-pi/4, pi/4] of x + y */
PT_HD double pt_k_cos(double x, double y) {
  const double z = x * x;
  const double p = pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, PT_K(PT_C6), PT_K(PT_C5)), PT_K(PT_C4)), PT_K(PT_C3)), PT_K(PT_C2)), PT_K(PT_C1));
  const double r = z * p;
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + pt_fma(z, r, -(x * y)));
}

/* Cody-Waite reduction of x by pi/2: x = n*pi/2 + y0 + y1, |y0| <= pi/4 (+ slop), for |x| < 2^20 * pi/2.
 * pio2_1 carries 33 bits, so fn * pio2_1 is exact and so is the first fused step; the second piece and its tail give
 * ~118 bits; when x lies so close to a multiple of pi/2 that this is not enough for y0's LAST bits, the second
 * subtraction was exact and a third piece is taken (fdlibm's scheme; the test here is on y0 itself). */
#define PT_INVPIO2 6.36619772367581382433e-01
#define PT_PIO2_1 1.57079632673412561417e+00
#define PT_PIO2_2 6.07710050630396597660e-11
#define PT_PIO2_2T 2.02226624879595063154e-21
#define PT_PIO2_3 2.02226624871116645580e-21
#define PT_PIO2_3T 8.47842766036889956997e-32
PT_HD int pt_rem_pio2(double x, double* y0, double* y1) {
  const double fn = pt_rint(x * PT_K(PT_INVPIO2));
  double r = pt_fma(-fn, PT_K(PT_PIO2_1), x); /* exact */
  double t = r;
  double w = fn * PT_K(PT_PIO2_2);
  r = t - w;
  w = pt_fma(fn, PT_K(PT_PIO2_2T), -((t - r) - w));
  double a = r - w;
  if (PT_UNLIKELY(pt_fabs(a) < 0x1p-45 * pt_fabs(x))) { /* (fn = 0: a = x, never taken) */
    t = r;
    w = fn * PT_K(PT_PIO2_3);
    r = t - w;
    w = pt_fma(fn, PT_K(PT_PIO2_3T), -((t - r) - w));
    a = r - w;
  }
  *y0 = a;
  *y1 = (r - a) - w;
  return (int)fn & 3; /* |fn| < 2^20: the conversion is exact, two's complement gives the residue of a negative n */
}
/* |x| >= 2^20 (never produced by the tracer, whose angles lie in [0, 2 pi)): the same reduction, no longer accurate but
 * deterministic, with the quadrant taken without an integer conversion that could overflow */
PT_HD int pt_rem_pio2_large(double x, double* y0, double* y1) {
  const double fn = pt_rint(x * PT_K(PT_INVPIO2));
  double r = x - fn * PT_K(PT_PIO2_1);
  const double t = r;
  double w = fn * PT_K(PT_PIO2_2);
  r = t - w;
  w = fn * PT_PIO2_2T - ((t - r) - w);
  const double a = r - w;
  *y0 = a;
  *y1 = (r - a) - w;
  const double q = fn - 4.0 * pt_trunc(fn * 0.25);
  return (int)q & 3;
}

PT_HD void pt_sincos(double x, double* s, double* c) {
  double y0, y1;
  int n;
  if (PT_LIKELY(pt_fabs(x) < 0x1p+20)) {
    n = pt_rem_pio2(x, &y0, &y1);
  } else {
    if (!pt_isfinite(x)) {
      *s = pt_nan();
      *c = pt_nan();
      return;
    }
    n = pt_rem_pio2_large(x, &y0, &y1);
  }
  const double ks = pt_k_sin(y0, y1);
  const double kc = pt_k_cos(y0, y1);
  const double sv = (n & 1) ? kc : ks;
  const double cv = (n & 1) ? ks : kc;
  /* sin changes sign in quadrants 2, 3; cos in 1, 2: flip the sign bit (exact, also for a zero) */
  *s = pt_from_bits(pt_bits(sv) ^ ((uint64_t)(n & 2) << 62));
  *c = pt_from_bits(pt_bits(cv) ^ ((uint64_t)((n + 1) & 2) << 62));
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
/* fdlibm's rational R(z) ~ (asin(sqrt z) - sqrt z) / (z sqrt z) on [0, 1/4]; numerator and denominator both < 2 */
PT_HD double pt_acos_r(double z) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  const double p = z * pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, PT_K(pS5), PT_K(pS4)), PT_K(pS3)), PT_K(pS2)), PT_K(pS1)), PT_K(pS0));
  const double q = pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, PT_K(qS4), PT_K(qS3)), PT_K(qS2)), PT_K(qS1)), 1.0);
  return pt_div_mid(p, q); /* q in [0.6, 1], 2^-116 <= p < 0.05 */
}
PT_HD double pt_acos(double x) {
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  const double pi = 3.14159265358979311600e+00;
  const double ax = pt_fabs(x);
  if (PT_UNLIKELY(!(ax < 1.0) || ax <= 0x1p-57)) { /* |x| >= 1, NaN, or so small that the answer is pi/2 */
    if (x != x) return pt_nan();
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    if (ax > 1.0) return pt_nan();
    return pio2_hi + pio2_lo;
  }
  if (ax < 0.5) {
    const double r = pt_acos_r(x * x);
    return pio2_hi - (x - pt_fma(-x, r, pio2_lo));
  }
  /* acos x = 2 asin sqrt((1 - |x|) / 2), reflected for x < 0; 1 - |x| is exact, z in [2^-54, 1/4] */
  const double z = (1.0 - ax) * 0.5;
  const double s = pt_sqrt_mid(z);
  const double r = pt_acos_r(z);
  if (x < 0.0) {
    const double w = pt_fma(r, s, -pio2_lo);
    return pi - 2.0 * (s + w);
  }
  /* sqrt z = s + c with c = (z - s^2) / (2 s): the residual of a correctly rounded square root is exact in one fma */
  const double c = pt_div_mid(pt_fma(-s, s, z), s + s);
  const double w = pt_fma(r, s, c);
  return 2.0 * (s + w);
}

/* ---- atan2 ---- */
/* The constants of atan2's special cases (zeros, infinities) are only ever needed in a branch that almost never runs, but as
 * plain literals the device compiler materialises all of them ahead of the shade kernel's main loop and keeps them in
 * registers across it -- 64-bit values the 128-VGPR kernel then spills.  Behind an empty volatile asm a value is made where
 * it is used.  Same bits either way. */
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
/* atan2 when an operand is a NaN, both are zero, or one is infinite (IEEE 754 / C99 F.9.1.4) */
PT_HD double pt_atan2_special(double y, double x) {
  if (x != x || y != y) return pt_nan();
  const int m = pt_signbit(y) | (pt_signbit(x) << 1);
  const int xinf = !pt_isfinite(x), yinf = !pt_isfinite(y);
  double r;
  if (xinf && yinf)
    r = (m & 2) ? PT_RARE_CONST(3.0 * PT_LIT_PI_O_4) : PT_RARE_CONST(PT_LIT_PI_O_4);
  else if (yinf)
    r = PT_RARE_CONST(PT_LIT_PI_O_2);
  else /* x infinite and y finite, or both zero */
    r = (m & 2) ? PT_RARE_CONST(PT_LIT_PI) : 0.0;
  return (m & 1) ? -r : r;
}
/* atan2 y x.  With mn <= mx the two magnitudes,
 *   atan(mn / mx) = atan c + atan((mn - c mx) / (mx + c mn)),   c in {0, 1/2, 1} by mn / mx < 7/16, < 11/16, else
 * (fdlibm's break points and its polynomial for |t| < 7/16): mn - c mx is exact (Sterbenz), so the ONE division carries
 * the only rounding of the reduction.  The octant / quadrant is restored as  m pi/2 + sigma r  with m in {0, 1, 2}:
 * (|y| <= |x|, x > 0): r;  (|y| > |x|): pi/2 -+ r for x >< 0;  (|y| <= |x|, x < 0): pi - r;  finally the sign of y.
 * pi and pi_lo are exactly twice pi/2's head and tail, so m * head and m * tail are exact.  One zero operand and tiny or
 * huge finite operands are first reduced to mid-range ones and then take the same path. */
PT_HD double pt_atan2(double y, double x) {
  const double atan_half_hi = 4.63647609000806093515e-01, atan_half_lo = 2.26987774529616870924e-17;
  const double atan_one_hi = 7.85398163397448278999e-01, atan_one_lo = 3.06161699786838301793e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
               a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
               a10 = 1.62858201153657823623e-02;
  const double ax = pt_fabs(x), ay = pt_fabs(y);
  const int sw = ax < ay;
  double mx = sw ? ay : ax, mn = sw ? ax : ay;
  if (PT_UNLIKELY(!(mx <= 0x1p+250 && mn >= 0x1p-250))) {
    if (!(mx > 0.0 && mx < pt_inf() && mn == mn)) return pt_atan2_special(y, x);
    if (mn * 0x1p+60 < mx) { /* the angle IS the ratio (to 2^-120): hand it on as (ratio, 1), whose division below is exact */
      mn = mn / mx;
      mx = 1.0;
    } else { /* comparable magnitudes at an end of the range: a power of two brings both to the middle, exactly */
      const double k = mx < 1.0 ? 0x1p+600 : 0x1p-600;
      mx *= k;
      mn *= k;
    }
  }
  const int b1 = mn >= 0.4375 * mx, b2 = mn >= 0.6875 * mx;
  const double c = b2 ? 1.0 : (b1 ? 0.5 : 0.0);
  const double hi = b2 ? atan_one_hi : (b1 ? atan_half_hi : 0.0);
  const double lo = b2 ? atan_one_lo : (b1 ? atan_half_lo : 0.0);
  const double t = pt_div_mid(pt_fma(-c, mx, mn), pt_fma(c, mn, mx)); /* |t| < 7/16 */
  const double z = t * t;
  const double p = z * pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z, pt_fma(z,
                       PT_K(a10), PT_K(a9)), PT_K(a8)), PT_K(a7)), PT_K(a6)), PT_K(a5)), PT_K(a4)), PT_K(a3)), PT_K(a2)), PT_K(a1)), PT_K(a0));
  const double r = hi - (pt_fma(t, p, -lo) - t); /* atan c + (t - t p(t^2)), fdlibm's grouping */
  const int xneg = pt_signbit(x);
  const double m = sw ? 1.0 : (xneg ? 2.0 : 0.0);
  const double sr = pt_from_bits(pt_bits(r) ^ ((uint64_t)(sw != xneg) << 63));
  const double q = pt_fma(m, PT_K(PT_LIT_PI_O_2), pt_fma(m, PT_K(0.5 * PT_LIT_PI_LO), sr));
  return pt_from_bits(pt_bits(q) | (pt_bits(y) & 0x8000000000000000ULL)); /* q >= 0 */
}

/* ---- x ** 5.0 : x^5 by double-double products (odd power keeps the sign) ---- */
PT_HD double pt_pow5(double x) {
  if (PT_UNLIKELY(!(pt_fabs(x) < 0x1p+200))) { /* NaN, infinities, and powers that overflow */
    if (x != x) return pt_nan();
    if (!pt_isfinite(x)) return x;
    const double h = (x * x) * (x * x);
    return h * x;
  }
  /* x^2 = p1 + e1 exactly */
  const double p1 = x * x;
  const double e1 = pt_fma(x, x, -p1);
  /* x^4 ~ p2 + e2 */
  const double p2 = p1 * p1;
  double e2 = pt_fma(p1, p1, -p2);
  e2 = pt_fma(p1 + p1, e1, e2);
  /* x^5 ~ p3 + e3 */
  const double p3 = p2 * x;
  double e3 = pt_fma(p2, x, -p3);
  e3 = pt_fma(e2, x, e3);
  return p3 + e3;
}

#endif /* PT_MATH_H */
