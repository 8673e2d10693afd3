/* kernels.hip -- the wavefront path integrator for gfx950 (MI355X), binary64 throughout.
 *
 * Stages (one kernel each, connected by path queues -- a 48-byte ray record + a 32-byte path-state record per entry -- that stay
 * resident in HBM):
 *   generate : sampler dims 0,1 + Camera.ray            (integrator.ml:96-105, camera.ml:93-102); fused into the bounce-0
 *              launches of trace and shade (PRIMARY), a kernel of its own only for explicit sample lists
 *   trace    : Scene.intersect = ordered BVH walk + leaf packets (shape_tree.ml:198-220, lib.rs:102-178,
 *              sphere.ml:35-54, triangle.ml:74-98, ganesha floor pre-test main.ml:286-298)
 *   shade    : Sphere.hit / Triangle.Hit.to_hit, Material.scatter, the body of Integrator's path loop
 *              (integrator.ml:30-66), background on a miss; survivors are compacted into the next queue
 *   accum    : per-pixel radiance sums in pass order
 *   film     : 3x3 binomial reconstruction + gamma (filter_kernel.ml, film_tile.ml, integrator.ml:114-128,152-154)
 * A render runs trace + shade of a bounce as ONE kernel (k_bounce: a wave walks 64 rays, files them in per-category pools, shades
 * a pool when it holds 64) -- around an LDS copy of the scene where tree and packets fit one, over the per-octant node image in
 * HBM / L2 otherwise; k_trace + k_shade_pool remain for what k_bounce does not cover (ptx_intersect_rays, scenes too large for
 * LDS beside the pools, PTX_FUSED = 0).
 * Both walks are stackless and their links TAGGED: a visit ends with one select between "what a hit leads to" and "what a miss leads
 * to", and the walk's control state -- wants a node, holds a leaf, over -- lives in the link's spare bits (PT_SWZ_TAG_* on the LDS
 * image, the top two bits of a word on the per-octant record).  On Simd_leaf scenes in LDS the node loop itself is gfx950 assembly
 * (PtTraverser::walk_asm); everything else is HIP C++.
 *
 * Compiled with -ffp-contract=off; every fused multiply-add below is written out exactly where the
 * reference writes Float.fma / _mm256_fmadd_pd.  No MFMA: there is no dense contraction on this path.
 */
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pt_scene.h"
#include "pt_vec.h"

#define PT_WAVE 64

/* ------------------------------------------------------------------ path queue (records in HBM) */
/* A queue entry is two records: the ray (what k_trace reads, densely) and the rest of the path's state (what only the shade
 * stage reads, entry by entry in the order its category pools dictate -- one 32-byte sector instead of five 8-byte fields in
 * five different sectors).  Both are multiples of 16 bytes: every access is a 16-byte load or store. */
struct PtRayRec { double ox, oy, oz, dx, dy, dz; };                  /* 48 B: ray origin, direction (padded to one 64-byte line
                                                                         per ray: shade +20 % -- partial-line writes) */
struct PtPathRec { double ar, ag, ab; uint32_t id; int32_t offset; }; /* 32 B: attn0 (integrator.ml:30); slot in the batch's
                                                                         contribution buffer; sampler offset = gy*W + gx + pass*spp
                                                                         (integrator.ml:98) */
struct PtEmitRec { double er, eg, eb, pad; };                         /* 32 B: emit0; scenes with emitters only */
/* Scenes with emitters keep an entry's path state and its carried emission in ONE 64-byte line: `path` then holds
 * {PtPathRec, PtEmitRec} pairs (entry i at path + 2 i).  The shade stage gathers an entry's records sparsely and memory moves
 * 64-byte lines: two 32-byte records in two arrays cost two lines per segment (cornell: 315 B fetched per segment against 144
 * of records), the pair costs one. */
struct PtQueue {
  PtRayRec* ray;
  PtPathRec* path;
  uint32_t* count; /* number of live entries (device) */
};

struct PtHits {
  double* t;     /* scenes without triangles: t_hit per entry */
  int32_t* slot; /* leaf slot index, -1 = miss; >= n_slots = floor triangle */
  /* scenes with triangles: ONE 32-byte record {t_hit, u, v, -} per entry (triangle barycentrics, triangle.ml:14-20), `t` is
   * then unused (NULL).  Round 4: only ptx_intersect_rays asks for these records (tuv != NULL).  A render does not keep them
   * at all: the shade step RECOMPUTES (t, u, v) from the ray and the one primitive that was hit (PT_RECOMPUTE_HIT) -- the
   * same function on the same operands, so the same bits -- because the record cost a 32-byte write per segment and, gathered
   * again one or two chunks later when the L2 no longer holds it, a whole 128-byte line of fabric traffic per segment
   * (cornell's bounce launches, TCC counters: 2.7 line requests per segment reach the L2, 2.4 miss): ~60 vector instructions
   * against a third of the kernel's memory traffic. */
  double4* tuv;
  /* k_bounce running the remaining bounces of a batch in one launch (PtSolo): its workgroups are at DIFFERENT bounces at the same
   * time, and entry i of the even queue and entry i of the odd queue would share t[i] -- bounce b keeps its distances at
   * t + (b & 1) * t_parity_stride there (0: `t` holds one array only and no launch runs solo). */
  size_t t_parity_stride;
};
#ifndef PT_RECOMPUTE_HIT
#define PT_RECOMPUTE_HIT 1
#endif
__device__ __forceinline__ void pt_hit_store(const PtHits& hits, uint32_t i, double t, int slot, double u, double v, bool with_uv) {
  hits.slot[i] = slot;
  if (with_uv) {
    if (!PT_RECOMPUTE_HIT || hits.tuv) hits.tuv[i] = make_double4(t, u, v, 0.0);
  } else hits.t[i] = t;
}

/* A queue written by k_shade_pool has HOLES (unused entries of its last blocks): direction x = a NaN whose payload no
 * arithmetic produces.  k_trace skips a hole and records PT_SLOT_HOLE for it. */
#define PT_HOLE_HI 0x7ff8dead
#define PT_SLOT_HOLE (-2)
__device__ __forceinline__ bool pt_is_hole(double dx) { return __double2hiint(dx) == (int)PT_HOLE_HI; }

__device__ __forceinline__ void pt_q_load_ray(const PtQueue& q, uint32_t i, V3& o, V3& d) {
  const double2* r = (const double2*)(q.ray + i);
  const double2 a = r[0], b = r[1], c = r[2];
  o = v3(a.x, a.y, b.x);
  d = v3(b.y, c.x, c.y);
}
template <bool EMIT>
__device__ __forceinline__ void pt_q_load_path(const PtQueue& q, uint32_t i, V3& attn, uint32_t& id, int& offset, V3& emit) {
  const double2* r = (const double2*)(q.path + (EMIT ? 2 * (size_t)i : (size_t)i));
  const double2 a = r[0], b = r[1];
  attn = v3(a.x, a.y, b.x);
  id = (uint32_t)__double2loint(b.y);
  offset = __double2hiint(b.y);
  if (EMIT) {
    const double2 c = r[2], e = r[3];
    emit = v3(c.x, c.y, e.x);
  }
}
__device__ __forceinline__ void pt_q_store_ray(const PtQueue& q, uint32_t i, V3 o, V3 d) {
  /* a real ray's NaN that happens to carry the hole payload (it can only come in through the caller's data: arithmetic
   * produces the canonical NaN or hands an operand's payload on) is stored as the canonical NaN: any NaN walks alike */
  if (pt_is_hole(d.x)) d.x = __hiloint2double(0x7ff80000, 0);
  double2* r = (double2*)(q.ray + i);
  r[0] = make_double2(o.x, o.y);
  r[1] = make_double2(o.z, d.x);
  r[2] = make_double2(d.y, d.z);
}
template <bool EMIT>
__device__ __forceinline__ void pt_q_store(const PtQueue& q, uint32_t i, V3 o, V3 d, V3 attn, V3 emit, uint32_t id, int offset) {
  pt_q_store_ray(q, i, o, d);
  double2* r = (double2*)(q.path + (EMIT ? 2 * (size_t)i : (size_t)i));
  r[0] = make_double2(attn.x, attn.y);
  r[1] = make_double2(attn.z, __hiloint2double(offset, (int)id));
  if (EMIT) {
    r[2] = make_double2(emit.x, emit.y);
    r[3] = make_double2(emit.z, 0.0);
  }
}

struct PtCounters { /* device-side work counters (count_work) */
  unsigned long long segments, nodes, prims, floor;
  unsigned long long undecided, fallback_steps; /* binary32 filter: lane tests handed to the binary64 code, wave steps that ran it */
  unsigned long long solo;                      /* k_bounce launches that ran their batch's remaining bounces by themselves (PtSolo) */
};

/* ------------------------------------------------------------------ small device helpers */
__device__ __forceinline__ int pt_lane() { return (int)(threadIdx.x & 63); }

/* popcount of a wave mask as a 32-bit scalar: `(int)__popcll(m) < K` is turned into a 64-bit comparison, for which the scalar
 * unit has no instruction -- it lands on the VECTOR pipe, once per turn of the walk */
__device__ __forceinline__ int pt_popc_mask(unsigned long long m) { return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32)); }

/* wave-aggregated append: returns the destination index for lanes with keep != 0 */
__device__ __forceinline__ uint32_t pt_wave_append(uint32_t* counter, bool keep) {
  const unsigned long long mask = __ballot(keep);
  const uint32_t total = (uint32_t)__popcll(mask);
  uint32_t base = 0;
  const int lane = pt_lane();
  const int leader = mask ? (int)__ffsll((unsigned long long)mask) - 1 : 0;
  if (lane == leader && total) base = atomicAdd(counter, total);
  base = (uint32_t)__shfl((int)base, leader, 64);
  const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  return base + rank;
}

__device__ __forceinline__ unsigned long long pt_wave_sum(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

/* Low_discrepancy_sequence.get (low_discrepancy_sequence.ml:19-20,33-36) */
__device__ __forceinline__ double pt_lds_get(const double* __restrict__ alpha, int offset, int dimension) {
  const double a = alpha[dimension];
  const double x = 0.5 + (a * (double)(1 + offset));
  return x - pt_trunc(x);
}

/* ------------------------------------------------------------------ generate */
struct PtGenParams {
  int32_t width, height;   /* full image */
  int32_t spp;
  int32_t local_rows;      /* rows this rank renders */
  int32_t band_rows, band_first, band_step;
  int32_t first_pass, n_pass; /* passes in this batch */
  int32_t tiles_x, tiles_y;   /* 8x8 pixel tiles over (width x local_rows) */
};

__device__ __forceinline__ int pt_global_row(const PtGenParams& g, int local_row) {
  if (g.band_step <= 1) return local_row;
  const int band_local = local_row / g.band_rows;
  const int within = local_row - band_local * g.band_rows;
  return (g.band_first + band_local * g.band_step) * g.band_rows + within;
}

/* Camera.ray (camera.ml:93-102): direction only; the origin is P3.origin */
__device__ __forceinline__ V3 pt_camera_dir(const PtSceneDev& sc, double cx, double cy) {
  return v3_normalize(v3(sc.cam_llx + (sc.cam_vx * cx), sc.cam_lly + (sc.cam_vy * cy), -1.0));
}

/* explicit (x, y, pass) triples: ptx_trace_samples */
__global__ __launch_bounds__(256) void k_generate_list(PtSceneDev sc, int width, int height, int spp, long long n,
                                                       const int32_t* __restrict__ xs, const int32_t* __restrict__ ys,
                                                       const int32_t* __restrict__ passes,
                                                       const double* __restrict__ alpha, PtQueue q) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  const uint32_t dst = pt_wave_append(q.count, valid);
  if (!valid) return;
  const int x = xs[i], gy = ys[i], pass = passes[i];
  const int offset = (gy * width) + x + (pass * spp);
  const double widthf = 1.0 / (double)width, heightf = 1.0 / (double)height;
  const double dxs = pt_lds_get(alpha, offset, 0), dys = pt_lds_get(alpha, offset, 1);
  const double cx = ((double)x + dxs) * widthf;
  const double cy = 1.0 - (((double)gy + dys) * heightf);
  const V3 dir = pt_camera_dir(sc, cx, cy);
  if (sc.has_emit) pt_q_store<true>(q, dst, v3(0.0, 0.0, 0.0), dir, v3(1.0, 1.0, 1.0), v3(0.0, 0.0, 0.0), (uint32_t)i, offset);
  else pt_q_store<false>(q, dst, v3(0.0, 0.0, 0.0), dir, v3(1.0, 1.0, 1.0), v3(0.0, 0.0, 0.0), (uint32_t)i, offset);
}

/* ------------------------------------------------------------------ trace */
/* Bbox.is_hit (bbox.ml:40-56) with Base's NaN-propagating min/max */
__device__ __forceinline__ bool pt_slab_hit(const PtNode& n, V3 o, V3 inv, double t_min, double t_max) {
  const double t0x = (n.mn[0] - o.x) * inv.x, t0y = (n.mn[1] - o.y) * inv.y, t0z = (n.mn[2] - o.z) * inv.z;
  const double t1x = (n.mx[0] - o.x) * inv.x, t1y = (n.mx[1] - o.y) * inv.y, t1z = (n.mx[2] - o.z) * inv.z;
  const double a = pt_base_max(pt_base_min(t0x, t1x), pt_base_max(pt_base_min(t0y, t1y), pt_base_min(t0z, t1z)));
  const double b = pt_base_min(pt_base_max(t0x, t1x), pt_base_min(pt_base_max(t0y, t1y), pt_base_max(t0z, t1z)));
  const double lo = pt_base_max(t_min, a);
  const double hi = pt_base_min(t_max, b);
  return lo <= hi;
}

/* Triangle.intersect (triangle.ml:74-98) */
__device__ __forceinline__ bool pt_triangle_intersect(V3 a, V3 b, V3 c, V3 o, V3 dir, double t_min, double t_max,
                                                      double* t_out, double* u_out, double* v_out) {
  const double epsilon = 1e-6;
  const V3 e1 = v3_sub(b, a);
  const V3 e2 = v3_sub(c, a);
  const V3 pvec = v3_cross(dir, e2);
  const double det = v3_dot(e1, pvec);
  if (pt_fabs(det) < epsilon) return false;
  const double det_inv = 1.0 / det;
  const V3 tvec = v3_sub(o, a);
  const double u = det_inv * v3_dot(tvec, pvec);
  const V3 qvec = v3_cross(tvec, e1);
  const double v = det_inv * v3_dot(dir, qvec);
  if (0.0 <= u && u <= 1.0 && 0.0 <= v && u + v <= 1.0) {
    const double t_hit = det_inv * v3_dot(e2, qvec);
    if (t_min <= t_hit && t_hit <= t_max) {
      *t_out = t_hit;
      *u_out = u;
      *v_out = v;
      return true;
    }
  }
  return false;
}

__device__ __forceinline__ V3 pt_load_v3(const double* p) { return v3(p[0], p[1], p[2]); }

/* Sphere.intersect, scalar (--no-simd / Array_leaf) form, sphere.ml:35-54 */
__device__ __forceinline__ bool pt_sphere_intersect_scalar(V3 center, double radius, V3 o, V3 d, double t_min,
                                                           double t_max, double* t_out) {
  const double r2 = radius * radius;
  const V3 f = v3_sub(center, o);
  const double bp = v3_dot(f, d);
  const double a = v3_quadrance(d);
  const double discrim = r2 - v3_quadrance(v3_sub(v3_scale(d, bp / a), f));
  if (discrim < 0.0) return false;
  const double sign_bp = (bp >= 0.0) ? 1.0 : -1.0;
  const double q = pt_fma(sign_bp, pt_sqrt(a * discrim), bp);
  const double c = v3_quadrance(f) - r2;
  const double t_hit = (c > 0.0) ? c / q : q / a;
  if (t_min <= t_hit && t_hit <= t_max) {
    *t_out = t_hit;
    return true;
  }
  return false;
}

/* Decodes entry i of a PRIMARY (bounce 0) launch: the queue is virtual -- entry i IS sample
 * (pass_in_batch, tile, lane) -- so camera rays are never written to or read from HBM. */
struct PtPrimarySample {
  int x, y, gy, pass_in_batch, offset;
  uint32_t id;
  bool valid;
};
__device__ __forceinline__ PtPrimarySample pt_primary_decode(const PtGenParams& g, uint32_t i) {
  PtPrimarySample s;
  const uint32_t per_pass = (uint32_t)(g.tiles_x * g.tiles_y) * 64u;
  const uint32_t pass_in_batch = i / per_pass;
  const uint32_t rem = i - pass_in_batch * per_pass;
  const int tile = (int)(rem >> 6), lane = (int)(rem & 63);
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  s.x = tx * 8 + (lane & 7);
  s.y = ty * 8 + (lane >> 3);
  s.pass_in_batch = (int)pass_in_batch;
  s.valid = (int)pass_in_batch < g.n_pass && s.x < g.width && s.y < g.local_rows;
  s.gy = pt_global_row(g, s.y);
  s.offset = (s.gy * g.width) + s.x + ((g.first_pass + (int)pass_in_batch) * g.spp); /* integrator.ml:98 */
  s.id = (uint32_t)((long long)pass_in_batch * g.width * g.local_rows + (long long)s.y * g.width + s.x);
  return s;
}
/* render_tile (integrator.ml:98-105) + Camera.ray (camera.ml:93-102) */
__device__ __forceinline__ V3 pt_primary_dir(const PtSceneDev& sc, const PtGenParams& g, const PtPrimarySample& s,
                                             const double* __restrict__ alpha) {
  const double widthf = 1.0 / (double)g.width, heightf = 1.0 / (double)g.height;
  const double dxs = pt_lds_get(alpha, s.offset, 0), dys = pt_lds_get(alpha, s.offset, 1);
  const double cx = ((double)s.x + dxs) * widthf;
  const double cy = 1.0 - (((double)s.gy + dys) * heightf);
  return pt_camera_dir(sc, cx, cy);
}

/* Bbox.is_hit when no 0 * inf can occur (every 1/d component finite): no NaN is ever produced, so Base's
 * NaN-propagating min/max coincide with the hardware v_min_f64 / v_max_f64 -- same boolean, 4x fewer
 * instructions.  (The values may differ in the sign of a zero, which no comparison can see.) */
template <bool ORIGIN_ZERO>
__device__ __forceinline__ bool pt_slab_hit_fast(const double* nb, V3 o, V3 inv, double t_min, double t_max) {
  /* camera rays start at P3.origin = (+0, +0, +0): x - (+0.0) == x bit for bit (also for x = -0.0), so the six
   * subtractions can be dropped for them */
  const double t0x = (ORIGIN_ZERO ? nb[0] : nb[0] - o.x) * inv.x, t0y = (ORIGIN_ZERO ? nb[1] : nb[1] - o.y) * inv.y,
               t0z = (ORIGIN_ZERO ? nb[2] : nb[2] - o.z) * inv.z;
  const double t1x = (ORIGIN_ZERO ? nb[3] : nb[3] - o.x) * inv.x, t1y = (ORIGIN_ZERO ? nb[4] : nb[4] - o.y) * inv.y,
               t1z = (ORIGIN_ZERO ? nb[5] : nb[5] - o.z) * inv.z;
  const double a = __builtin_fmax(__builtin_fmin(t0x, t1x), __builtin_fmax(__builtin_fmin(t0y, t1y), __builtin_fmin(t0z, t1z)));
  const double b = __builtin_fmin(__builtin_fmax(t0x, t1x), __builtin_fmin(__builtin_fmax(t0y, t1y), __builtin_fmax(t0z, t1z)));
  return __builtin_fmax(t_min, a) <= __builtin_fmin(t_max, b);
}
__device__ __forceinline__ bool pt_slab_hit_exact(const double* nb, V3 o, V3 inv, double t_min, double t_max) {
  PtNode n;
  n.mn[0] = nb[0]; n.mn[1] = nb[1]; n.mn[2] = nb[2];
  n.mx[0] = nb[3]; n.mx[1] = nb[4]; n.mx[2] = nb[5];
  return pt_slab_hit(n, o, inv, t_min, t_max);
}

/* LDS image of a node for LDS-resident scenes: 48 bytes -- a binary32 FILTER in front of the binary64 slab test, and the
 * tree THREADED per direction octant so that the walk needs no stack.
 *   words 0..5  mn.x mn.y mn.z mx.x mx.y mx.z rounded to binary32
 *   word 6      branch: lhs | rhs << 16 (node BYTE offsets into this image);  leaf: first slot | real slot count << 16
 *   word 7      bits of `mag` = max |bound| of the node (binary32, rounded up), low 2 bits replaced by the axis (3 = leaf)
 *   words 8..11 skip[8], u16 each: for a ray whose direction signs are octant o (shape_tree.ml:201), the node the
 *               reference's recursion visits next once this node's subtree is done (box missed, leaf taken, or both
 *               children searched) = the far sibling of the nearest ancestor entered through its near child; 0xffff = none.
 *               Same visiting order as the recursion (near subtree completely, then the far child, tested on arrival against
 *               the closest hit so far: shape_tree.ml:209-216), hence the same tests and counters, but no push, no pop, no
 *               stack pointer and no per-lane stack in LDS: ~12 fewer vector instructions and 1 LDS access less per visit.
 * Bbox.is_hit (bbox.ml:40-56) is a boolean of binary64 quantities; the image decides it in binary32 with a rigorous
 * error bound and hands the (rare) undecided lanes to the binary64 code, so the boolean -- and with it every hit, every
 * work counter and every pixel -- is the reference's.  For a ray with (root_mag + max|o|) max|1/d| < 2^100 and
 * max|1/d| > 2^-60 (else: always binary64):
 *   t~ = fma32(bound32, inv32, -(o * inv)32) differs from the reference's fl64((bound - o) * inv) by at most
 *        3.2 * 2^-24 * (|bound| + |o|) * |inv|  <=  M := 3.2 * 2^-24 * (mag + max|o|) * max|inv|
 *   (three binary32 roundings of the inputs, one of the fma, the reference's own two binary64 roundings; with PT_F32_INV,
 *   where inv32 and (o inv)32 are computed in binary32 from the start: 6 instead of 3.2, see PtTraverser::begin);
 *   max / min are 1-Lipschitz, so lo~ = max(0, a~) and hi~ = min(b~, t32) are within M (+ 2^-24 t for the rounded
 *   closest-hit distance) of the reference's lo, hi, and u = hi~ - lo~ (one more rounding) within 8.4 * 2^-24 * (..)
 *   + 2^-24 t of hi - lo.  With m2 = 2^-19 * (mag + max|o|) * max|inv| + 2^-21 * t32  (twice that bound):
 *        u >= m2  =>  lo <= hi (hit)        u <= -m2  =>  lo > hi (miss)        otherwise (|u| < m2 or unordered): binary64.
 * The undecided share is ~4 m2 / (hi - lo spread) ~ 1e-5 per test, scale-free because mag is the node's own. */
#ifndef PT_SWZ_NEAR
#define PT_SWZ_NEAR 1 /* 1: the LDS image also carries, per direction octant, the child the walk descends into (64-byte nodes) */
#endif
#ifndef PT_SWZ_SIGNSEL
#define PT_SWZ_SIGNSEL 1 /* needs PT_SWZ_NEAR.  1: 80-byte nodes whose bounds are stored as (mn, mx, mn) per axis, node references are
                            absolute LDS addresses */
#endif
#if PT_SWZ_NEAR && PT_SWZ_SIGNSEL
/* Layout 3 (this one).  The vector pipe is what the walk is bound by, so the image is arranged to take instructions out of a
 * visit:
 *   bytes  0..35  per axis a: mn_a, mx_a, mn_a (binary32).  A ray reads the PAIR at 12 a + (d_a >= 0 ? 0 : 4): (near, far)
 *                 bound of that axis for its direction sign, so t_near = fma(near, inv, n), t_far = fma(far, inv, n) need no
 *                 min / max per axis (an fma with a fixed second and third operand is monotone in the first, so these are
 *                 bit for bit the min and max the unselected form computes): 4 min / max per visit instead of 10
 *   bytes 36..39  word 6 as before: branch lhs | rhs << 16, leaf first slot | real count << 16
 *   bytes 40..43  word 7 as before: mag | axis
 *   bytes 44..59  skip[8] u16          bytes 60..75  near[8] u16          bytes 76..83  unused (see PT_SWZ_NODE_BYTES)
 * and every node reference (lhs, rhs, skip, near, the walk's `node`) is the node's ABSOLUTE LDS byte address, so a visit
 * starts reading at `node` itself (PT_SWZ_END / PT_SWZ_LEAF stay out of range: the image ends below 0xfffe). */
#ifndef PT_SWZ_NODE_BYTES
#define PT_SWZ_NODE_BYTES 92 /* 80 used + 12: 23 words, an ODD stride, so that node k starts in LDS bank 21 k mod 64 -- all 64 banks.
                                With 80 bytes (20 words) the nodes start in 16 of the 64 banks only, with 64 bytes in 4 */
#endif
#define PT_SWZ_OFF_LINKS 36
#define PT_SWZ_OFF_SKIP 44
#define PT_SWZ_OFF_NEAR 60
typedef const __attribute__((address_space(3))) unsigned char* PtLdsPtr;
/* native vectors (HIP's float2 class cannot be read through an address-space-qualified pointer), 4-byte aligned: the pairs
 * start at 12 a + {0, 4} */
typedef float pt_f2 __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned int pt_u2 __attribute__((ext_vector_type(2), aligned(4)));
#define PT_LDS_AT(addr) ((PtLdsPtr)(uintptr_t)(uint32_t)(addr))
#elif PT_SWZ_NEAR
/*   words 12..15 near[8], u16 each: for octant o the NEAR child of a branch (shape_tree.ml:209: lhs if bit `axis` of o is set,
 *               else rhs), PT_SWZ_LEAF for a leaf.  With it a visit needs no axis extraction, no bit test and no child select:
 *               next = hit an inner node ? near[o] : skip[o]. */
#define PT_SWZ_NODE_BYTES 64
#define PT_SWZ_OFF_LINKS 24
#define PT_SWZ_OFF_SKIP 32
#define PT_SWZ_OFF_NEAR 48
#undef PT_SWZ_SIGNSEL
#define PT_SWZ_SIGNSEL 0
#else
#define PT_SWZ_NODE_BYTES 48
#define PT_SWZ_OFF_LINKS 24
#define PT_SWZ_OFF_SKIP 32
#undef PT_SWZ_SIGNSEL
#define PT_SWZ_SIGNSEL 0
#endif
#define PT_SWZ_LEAF 0xfffeu
/* Layout 3: TAGGED links (round 5).  Node addresses are multiples of 4, so the two low bits of a 16-bit link are free:
 *   tag 0  a node to visit                       tag 1  "this lane holds a leaf; afterwards continue at link & ~3"
 *   tag 2  PT_SWZ_END: the walk is over          tag 3  "holds a leaf; afterwards the walk is over" (PT_SWZ_END | 1)
 * A leaf's near[o] entry is its skip[o] entry with bit 0 set, so a visit ends with ONE select, `node = hit ? near[o] : skip[o]`,
 * and the walk's whole control state is that register: wants a node step <=> (node & 3) == 0, holds a leaf <=> node & 1,
 * finished <=> node == PT_SWZ_END.  The leaf's (first slot, real count) word stays in the register the visit loaded it into
 * (`lkx`) until the leaf phase decodes it.  Before: a leaf compare, two conditional updates of (leaf_first, leaf_n), the END
 * compare, the conditional update of `node` and the (walking > leaf_n) compare -- 9 of a visit's 27 vector instructions. */
#ifndef PT_SWZ_TAGGED
#define PT_SWZ_TAGGED 1 /* (0: the untagged links of round 4, for the A/B) */
#endif
#define PT_SWZ_TAG_LEAF 1u
#define PT_SWZ_TAG_END 2u
/* doubles per sphere slot in the LDS copy: 4 used ({x, y, z, r}) + padding.  With 4 (8 words) a slot starts in 8 of the 64 banks
 * only; with 6 (12 words) in 16.  Global memory keeps 4. */
#ifndef PT_LDS_SPH_DOUBLES
#define PT_LDS_SPH_DOUBLES 4
#endif
#define PT_SPH_STRIDE(SWZ_) ((SWZ_) ? PT_LDS_SPH_DOUBLES : 4)
#if PT_SWZ_SIGNSEL && PT_SWZ_TAGGED
#define PT_SWZ_END 0xfffeu /* tag 2 (see PT_SWZ_TAG_*) */
#else
#define PT_SWZ_END 0xffffu
#endif
/* bytes of LDS a wave keeps for traversal stacks: LDS-resident scenes walk the threaded image (no per-lane stack) and
 * only the camera-ray packet walk keeps its shared (node, mask) stack there: 12 bytes per level, rounded to 16 */
#define PT_WAVE_STACK_BYTES(LDS_SCENE, StackT, depth) ((LDS_SCENE) ? (size_t)(depth) * 16u : ((std::is_same<StackT, PtThreadTag>::value || std::is_same<StackT, PtThreadOctTag>::value) ? (size_t)0 : (size_t)(depth) * PT_WAVE * sizeof(StackT)))

/* where the traversal data of this launch lives: HBM/L2 (large scenes) or an LDS copy (small scenes) */
/* StackT of a walk that needs no stack: scenes traversed from HBM / L2 follow per-octant skip links like the LDS image
 * does (PtSceneDev.node_skip32), which frees the LDS the per-lane stacks took (depth x 256 bytes per wave: it capped
 * the ganesha-like scene at 3 waves per SIMD where its registers allow 4) */
struct PtThreadTag {};
/* ... and the same walk on the PER-OCTANT node image (PtSceneDev.nodes32o): one 32-byte record per (direction octant, node) that
 * carries the six binary32 bounds, ONE link word and the skip link of that octant -- a visit is two 16-byte loads instead of
 * two + the skip table's entry.  The texture-address unit that binds the walk from HBM / L2 works per load instruction and
 * lane (profiles/r03c_ganesha_ta_tcp.txt: three per node test); the price is eight copies of the bounds (23 MB for 92 k nodes
 * instead of 6 MB of image + table).  A pre-order tree needs no lhs link (the lhs child of node k is k + 1), which is what
 * frees the word: inner nodes keep `rhs | axis << 30`, leaves pack `tag << 30 | real slots << 22 | first slot`. */
struct PtThreadOctTag {};
#define PT_OCT_LEAF_FIRST_BITS 22
#define PT_OCT_LEAF_REAL_MAX 255u
/* Round 5, second layout of the per-octant record: what the tagged links did for the LDS image, for the walk from
 * HBM / L2 -- where four fifths of the visits are answered by the L1 in ~120 clocks and the ~45 vector + ~40 scalar instructions
 * the old record cost per visit were as long a chain as the load.  An octant fixes the sign of every direction component, so
 *   words 0..2  the NEAR bound of each axis for this octant (mn where the direction component is >= 0, else mx), words 3..5 the
 *               FAR one: t_near = fma(near, inv, n), t_far = fma(far, inv, n) -- the six min / max per visit are gone (an fma with
 *               fixed second and third operand is monotone in the first: the same bits);
 *   word 6      what a HIT leads to: an inner node -> its near child for this octant (an index < 2^30: tag 00 in the top bits);
 *               a leaf -> 01 << 30 | real slots << 22 | first slot: "this lane holds a leaf", the leaf's packet in the same word;
 *   word 7      what a MISS -- or the end of the leaf -- leads to: the octant's skip link, or PT_OCT_END (1 << 31: nothing).
 * A visit ends with one select, node = hit ? word 6 : word 7, and the walk's control state is that register: wants a node step
 * <=> node < 2^30, holds a leaf <=> node >> 30 == 1, over <=> node >= 2^31; after the leaf phase the walk continues at the word 7
 * the leaf's visit loaded (oct_skip).  The node's magnitude for the filter's margin comes from the bounds themselves (three
 * v_max3 with |.| modifiers); its 1.000001 lives in the ray's k2. */
/* (The record of round 4 -- six bounds as stored, `rhs | axis << 30` / packed leaf word, skip link -- and the first tagged record with
 * un-offset links were this layout's A/B partners: profiles/r05_ab_oct_tagged.txt, r05_ab_oct_preoffset.txt; removed from the source
 * once measured; commit afabea1 is the last whose source holds both alternates.) */
#ifndef PT_FILTER_DEBUG
#define PT_FILTER_DEBUG 0 /* diagnostic builds only (tools/filter_error_study.py): the traverser keeps the u and the margin of its last box test on the
                             per-octant record, and the library gains ptx_debug_filter_error -- not in the product build */
#endif
/* Links are PRE-OFFSET: the links of octant o's records are record numbers in the whole image (o * n_nodes + k) and so is the walk's
 * `node`: a visit's address is one 32-bit shift beside the image's scalar base (the image is smaller than 4 GiB: the host builds it
 * only then) instead of add, 64-bit shift, 64-bit add at the head of its chain; the binary64 fallback subtracts the octant's base. */
#ifndef PT_OCT_BATCH_FALLBACK
#define PT_OCT_BATCH_FALLBACK 1 /* (0: evaluated there and then, as until round 5's last change) a lane whose box test the binary32 filter leaves undecided does not stop its
                                   wave for the ~100 instructions of the binary64 test there and then (22 % of the mesh's wave steps did, for a handful of
                                   lanes): it leaves the node loop like a lane that holds a leaf, its node tagged PT_OCT_PENDING, and all such lanes take
                                   the binary64 test TOGETHER after the loop (resolve_pending), before the leaf phase.  Same tests per ray, same order */
#endif
#define PT_OCT_PENDING 0xc0000000u /* top bits 11: "the test of node (low 30 bits) awaits its binary64 evaluation"; oct_link / oct_skip hold its links */
#define PT_OCT_END 0x80000000u
#define PT_OCT_LEAF_TAG 0x40000000u

struct PtSceneView {
  const PtNode* nodes;
  const uint32_t* skip32; /* threaded global walk: n_nodes x 8, 0xffffffff = none */
  const unsigned char* nodes32; /* 32-byte binary32 image of the nodes for the walk from HBM / L2 */
  const unsigned char* nodes32o; /* ... per direction octant (PtThreadOctTag), 8 x n_nodes x 32 bytes, or null */
  uint32_t n_nodes;
  const unsigned char* swz_nodes; /* LDS-resident scenes: the binary32 filter image, PT_SWZ_NODE_BYTES per node */
  uint32_t swz_root;              /* what the walk's `node` is for node 0: 0 (byte offsets into the image) or, with PT_SWZ_SIGNSEL, the
                                     image's absolute LDS address */
  const unsigned char* top;       /* scenes walked from HBM / L2: the LDS copy of PtSceneDev.top_nodes if has_top (else never dereferenced:
                                     no node reference carries PT_TOP_FLAG) */
  bool has_top;
  const double* sph;
  const double* tri;
  const uint8_t* kind;
  const double* nodes64; /* LDS-resident scenes with PtSceneDev.lds_nodes64: six binary64 bounds per node in LDS, else null */
  const uint8_t* cat; /* the slots' shading categories: the LDS copy on LDS-resident scenes (k_bounce files a finished ray by it right
                         after the walk: from global memory that one byte was a ~1 us round trip in every chunk's chain) */
  /* scenes walked from HBM / L2: the first n_floor_lds floor triangles (ganesha's Floor, tested before the tree for EVERY ray,
   * main.ml:247-256) as 10-double records in LDS.  Read from global memory their 2 x 5 loads per ray were a fifth of the
   * walk's vector-memory instructions -- all lanes at the same address, but the texture-address unit that binds this kernel
   * still processes every one of them; an LDS read of one address is a broadcast on a pipe the kernel does not use. */
  const __attribute__((address_space(3))) double* floor_lds;
  int n_floor_lds;
};
#define PT_FLOOR_LDS 4

struct PtTraceResult {
  double t, u, v;
  int slot;
};

/* Scene.intersect for ONE ray held by this lane.  The traversal stack (far children only) lives in LDS,
 * one column per lane (conflict-free ds_write_b32 / ds_read_b32).  A far child's bbox is tested when it
 * is POPPED, against the closest hit so far -- exactly the t_max the reference's recursion passes
 * (shape_tree.ml:210-216). */
template <class T> __device__ __forceinline__ void pt_stack_push(T* stk, int sp, uint32_t val) { stk[sp * PT_WAVE] = (T)val; }
template <class T> __device__ __forceinline__ uint32_t pt_stack_pop(const T* stk, int sp) { return (uint32_t)stk[sp * PT_WAVE]; }
__device__ __forceinline__ void pt_stack_push(PtThreadTag*, int, uint32_t) {}
__device__ __forceinline__ uint32_t pt_stack_pop(const PtThreadTag*, int) { return 0u; }
__device__ __forceinline__ void pt_stack_push(PtThreadOctTag*, int, uint32_t) {}
__device__ __forceinline__ uint32_t pt_stack_pop(const PtThreadOctTag*, int) { return 0u; }
#define PT_STACK_PUSH(stk, sp, val) pt_stack_push((stk), (sp), (uint32_t)(val))
#define PT_STACK_POP(stk, sp) pt_stack_pop((stk), (sp))

#ifndef PT_WALK_MIN
#define PT_WALK_MIN 8
#endif
#ifndef PT_WALK_MIN_GLOBAL
#define PT_WALK_MIN_GLOBAL 16 /* the same threshold for the walk from HBM / L2, where a node step is a round trip to the L2 and the leaf phase's
                                 first loads are the longest waits of the walk: leaves are taken up sooner (ganesha-like frame 4: 28.1, 8: 26.4, 16: 25.8 ms) */
#endif
#ifndef PT_LEAF_PREFETCH
#define PT_LEAF_PREFETCH 1 /* triangle-only scenes walked from HBM / L2: request triangle k + 1 before testing triangle k */
#endif
#ifndef PT_MARGIN_K2
#define PT_MARGIN_K2 0x1.000002p-19f /* the filter's margin m2 = PT_MARGIN_K2 (mag + max|o|) max|1/d| + PT_MARGIN_T t (header comment) */
#define PT_MARGIN_T 0x1p-21f
#endif
#ifndef PT_F32_INV
#define PT_F32_INV 1 /* the filter's reciprocals by v_rcp_f32 instead of Ray.create's three binary64 divisions (PtTraverser::begin) */
#endif
#ifndef PT_WALK_LOOP
#define PT_WALK_LOOP 0 /* 0: wave-level loop with a `want` ballot per turn; 1: the node walk as one divergent loop (fewer scalar
                          instructions per turn, but the kernel is bound by VECTOR issue: measured 3 % slower) */
#endif
/* PT_DIAG (diagnostic builds only, tools/diag_utilisation.sh): re-purposes the COUNT counters of SECONDARY launches
 * to measure lane utilisation per traversal phase: nodes = useful lane steps, floor = lane slots the wave spent.
 * 1: node walk   2: node walk if only the per-chunk tail were lost   3: packet scan   4: packet heavy part
 * 5 / 6 (k_bounce only, tools/diag_phases.py): a wave's life in 100 MHz ticks per phase -- walk / shade steps / rest, or walk / its leaf phases */
#ifndef PT_DIAG
#define PT_DIAG 0
#endif
#ifndef PT_DIAG_EXTRA_LOADS
#define PT_DIAG_EXTRA_LOADS 0
#endif
/* diagnostic builds only (tools/README.md, "what binds the node loop"): PT_DIAG_VISIT_VALU = n more vector instructions per
 * visit of the LDS walk, PT_DIAG_VISIT_LDS = n more 16-bit LDS reads (+ one vector add each), PT_DIAG_VISIT_SALU = n more scalar ones */
#ifndef PT_DIAG_VISIT_VALU
#define PT_DIAG_VISIT_VALU 0
#endif
#ifndef PT_DIAG_VISIT_LDS
#define PT_DIAG_VISIT_LDS 0
#endif
#ifndef PT_DIAG_VISIT_SALU
#define PT_DIAG_VISIT_SALU 0
#endif
#ifndef PT_WALK_ASM
#define PT_WALK_ASM 1 /* k_bounce on LDS scenes: the node loop in assembly (PtTraverser::walk_asm); 0: the compiler's loop */
#endif
#ifndef PT_SCAN_ADDC
#define PT_SCAN_ADDC 1
#endif
#ifndef PT_PACKET_DEFER
#define PT_PACKET_DEFER 1
#endif
#define PT_DIAG_WAVE_SLOTS(c) do { if (pt_lane() == __ffsll((long long)__ballot(1)) - 1) (c) += 64; } while (0)

/* One ray's traversal state.  begin() = Ray.create + the per-ray constants; node_step() = one visit of
 * Tree.intersect's recursion (shape_tree.ml:203-221); packet() = Leaf.intersect on the leaf the lane holds.
 * Driven by pt_trace_ray (one ray per lane; a chunk may stop early and its unfinished walks be resumed later, PtTailCtl) and by
 * pt_trace_packet (camera rays of LDS-resident scenes). */
template <int MODE, bool COUNT, bool ORIGIN_ZERO, typename StackT, bool SWZ>
struct PtTraverser {
  /* the binary32 filter runs wherever the walk is threaded: on the LDS image (SWZ) and on the 32-byte global image */
  static constexpr bool OCT = !SWZ && std::is_same<StackT, PtThreadOctTag>::value; /* the per-octant image (PtThreadOctTag) */
  static constexpr bool G32 = !SWZ && (std::is_same<StackT, PtThreadTag>::value || OCT);
  static constexpr bool FILT = SWZ || G32;
  V3 o, d, inv; /* SWZ: inv is not kept (the binary64 fallback recomputes 1 / d, the same three divisions) */
  uint32_t dirs;
  bool exact_slab; /* SWZ: also set when the binary32 filter does not apply to this ray (|1/d| >= 2^100) */
  /* binary32 filter constants of the ray (SWZ only): inv32, -(o * inv)32, k2 = 2^-19 max|inv|, c2 = max|o| k2 + 2^-21 t32 */
  float fix, fiy, fiz, fnx, fny, fnz, k2, c2base, c2, t32;
  uint32_t skip_off; /* SWZ: byte offset of this ray's octant entry in a node's skip table; OCT: index of the octant's node 0 */
  mutable uint32_t oct_skip; /* OCT: the visited node's skip link, out of its record (test_box) */
  mutable uint32_t oct_link; /* OTAG: word 6 of the visited node's record (what a hit leads to) */
#if PT_FILTER_DEBUG
  mutable float dbg_u, dbg_m2; /* the binary32 filter's u = hi~ - lo~ and margin m2 of the last box test (OTAG branch) */
#endif
  uint32_t sel_x, sel_y, sel_z; /* PT_SWZ_SIGNSEL: byte offsets of the ray's (near, far) bound pairs */
  mutable unsigned long long n_undecided = 0, n_wave_fallbacks = 0; /* COUNT only (ptx_stats.filter_*) */
  double qa, one_over_a;
  PtTraceResult r;
  int sp;
  /* TAGGED (the LDS image, layout 3): `node` carries the walk's control state in its two low bits (PT_SWZ_TAG_*), `walking` and
   * `leaf_n` are not used between the leaf phases; `lkx` = word 6 of the node visited last (a leaf's first slot | real count << 16) */
  static constexpr bool TAGGED = SWZ && (PT_SWZ_SIGNSEL != 0) && (PT_SWZ_TAGGED != 0);
  static constexpr bool OTAG = OCT; /* the per-octant record: tagged, pre-offset links */
  static constexpr bool TAGS = TAGGED || OTAG;
  uint32_t node;
  mutable uint32_t lkx;
  uint32_t lkx_diag = 0u, sdiag = 0u; /* (PT_DIAG_VISIT_*) */
  uint32_t walking; /* 0 / 1: an integer, so that "wants a node step" is ONE unsigned comparison (walking > leaf_n) */
  int leaf_first, leaf_n;
  __device__ __forceinline__ bool wants_node() const { return OTAG ? node < PT_OCT_LEAF_TAG : (TAGGED ? (node & 3u) == 0u : walking > (uint32_t)leaf_n); }
  /* the same as a wave mask, straight from the comparison: the ballot builtin of the very expression the branch tests lets the
   * compiler use ONE v_cmp for both (HIP's __ballot of a boolean that is also branched on costs two more vector instructions
   * per turn; __builtin_amdgcn_uicmp a second compare); 38 = signed greater than, 33 = not equal */
  __device__ __forceinline__ unsigned long long wants_node_mask() const { return __builtin_amdgcn_ballot_w64(wants_node()); }
  __device__ __forceinline__ bool holds_leaf() const { return OTAG ? (node >> 30) == 1u : (TAGGED ? (node & PT_SWZ_TAG_LEAF) != 0u : leaf_n > 0); }
  __device__ __forceinline__ unsigned long long holds_leaf_mask() const { return TAGS ? __builtin_amdgcn_ballot_w64(holds_leaf()) : __builtin_amdgcn_sicmp(leaf_n, 0, 38); }
  /* the ray's walk is not over (it wants a node step or holds a leaf) */
  __device__ __forceinline__ bool alive() const {
    return OTAG ? (PT_OCT_BATCH_FALLBACK ? (node & PT_OCT_PENDING) != PT_OCT_END : node < PT_OCT_END) : (TAGGED ? node != PT_SWZ_END : (walking != 0u || leaf_n > 0));
  }
  __device__ __forceinline__ bool pending() const { return OTAG && PT_OCT_BATCH_FALLBACK && node >= PT_OCT_PENDING; }
  /* ... as a wave mask, where no lane holds a leaf (after the leaf phase: the chunk cut) */
  __device__ __forceinline__ unsigned long long walking_mask() const { return TAGS ? __builtin_amdgcn_ballot_w64(alive()) : __builtin_amdgcn_uicmp(walking, 0u, 33); }

  __device__ __forceinline__ void begin(const PtSceneDev& sc, const PtSceneView& sv, V3 o_, V3 d_,
                                        unsigned long long& c_floor) {
    o = o_;
    d = d_;
    /* dirs, shape_tree.ml:201 */
    dirs = (d.x >= 0.0 ? 1u : 0u) | (d.y >= 0.0 ? 2u : 0u) | (d.z >= 0.0 ? 4u : 0u);
    if (!FILT || !PT_F32_INV) {
      inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z); /* Ray.create, ray.ml:7-10 */
      exact_slab = !(pt_isfinite(inv.x) && pt_isfinite(inv.y) && pt_isfinite(inv.z));
    } else {
      inv = v3(0.0, 0.0, 0.0); /* the filter takes its reciprocals in binary32 (below); the binary64 fallback divides for itself */
      exact_slab = false;
    }
    if (FILT) {
#if PT_SWZ_SIGNSEL
      skip_off = 2u * dirs;
      if (SWZ) { /* where this ray's (near, far) pair of each axis starts inside a node */
        sel_x = (dirs & 1u) ? 0u : 4u;
        sel_y = 12u + ((dirs & 2u) ? 0u : 4u);
        sel_z = 24u + ((dirs & 4u) ? 0u : 4u);
      }
#else
      skip_off = PT_SWZ_OFF_SKIP + 2u * dirs;
#endif
      const double omax = __builtin_fmax(pt_fabs(o.x), __builtin_fmax(pt_fabs(o.y), pt_fabs(o.z)));
#if PT_F32_INV
      /* The filter's constants straight in binary32: inv32 = v_rcp_f32(fl32(d)) (1 ulp) is within 3 * 2^-24 of 1 / d and
       * (o inv)32 = fl32(fl32(o) inv32) within 5 * 2^-24 of o / d, instead of one rounding each from Ray.create's binary64
       * quotients -- three binary64 divisions per ray (~40 vector instructions of a walk's ~1000) that only the filter used.
       * The bound above becomes 6 (t~) and 13 (u) units of 2^-24 (..) against m2 = 32 of them: still more than twice.  A zero,
       * subnormal or NaN component gives inf / NaN here and fails the guard below (binary64 throughout, as before); a
       * component beyond binary32 gives inv32 = 0 for a true |1 / d| < 2^-127, the flush case the guard's comment covers. */
      fix = __builtin_amdgcn_rcpf((float)d.x);
      fiy = __builtin_amdgcn_rcpf((float)d.y);
      fiz = __builtin_amdgcn_rcpf((float)d.z);
      const float fimax = __builtin_fmaxf(__builtin_fabsf(fix), __builtin_fmaxf(__builtin_fabsf(fiy), __builtin_fabsf(fiz)));
      const float fisum = __builtin_fabsf(fix) + __builtin_fabsf(fiy) + __builtin_fabsf(fiz); /* (fmax drops a NaN, a sum keeps it) */
#else
      const double ax = pt_fabs(inv.x), ay = pt_fabs(inv.y), az = pt_fabs(inv.z);
      const double imax = __builtin_fmax(ax, __builtin_fmax(ay, az));
#endif
      /* every binary32 intermediate stays far inside the format: (mag + |o|) |inv| < 2^100 for every node, because every
       * node lies inside the root box (root_mag = its largest |coordinate|, +inf when that exceeds binary32: such a scene
       * is walked in binary64 throughout), so no product, sum or margin of the filter can overflow, whatever the scene's
       * scale.  And max|1/d| is kept far above the binary32 subnormals (>= 2^-60): components of inv32 / (o inv)32 that
       * are subnormal -- or flushed to zero, whatever the f32 denormal mode of the code object -- are then wrong by
       * < 2^-126 (mag + |o|) absolute, which m2 >= 2^-19 (mag + |o|) 2^-60 + 1e-30 covers with room to spare. */
      const float fomax = (float)omax; /* (a magnitude beyond binary32 becomes +inf and fails the guard) */
#if PT_F32_INV
      if (!((sc.root_mag + fomax) * fisum < 0x1p100f) || !(fimax > 0x1p-60f)) exact_slab = true;
      fnx = ORIGIN_ZERO ? 0.0f : -((float)o.x * fix);
      fny = ORIGIN_ZERO ? 0.0f : -((float)o.y * fiy);
      fnz = ORIGIN_ZERO ? 0.0f : -((float)o.z * fiz);
#else
      const float fimax = (float)imax;
      if (!((sc.root_mag + fomax) * fimax < 0x1p100f) || !(fimax > 0x1p-60f)) exact_slab = true;
      fix = (float)inv.x; fiy = (float)inv.y; fiz = (float)inv.z;
      fnx = ORIGIN_ZERO ? 0.0f : -(float)(o.x * inv.x);
      fny = ORIGIN_ZERO ? 0.0f : -(float)(o.y * inv.y);
      fnz = ORIGIN_ZERO ? 0.0f : -(float)(o.z * inv.z);
#endif
      k2 = fimax * PT_MARGIN_K2;
      if (OTAG) k2 *= 1.000002f; /* (the node's magnitude is taken from its bounds as they are: test_box) */
      c2base = __builtin_fmaf(fomax * 1.000001f, k2, 1e-30f);
      /* exact_slab folded into the margin: with m2 = NaN neither `u >= m2` nor `u < -m2` holds, so every test of such a ray
       * is undecided and takes the binary64 code -- no per-visit instruction for the flag itself */
      if (exact_slab) c2base = __builtin_nanf("");
    }
    r.t = PT_MAX_FINITE;
    r.slot = -1;
    r.u = 0.0;
    r.v = 0.0;
    /* ganesha Floor.intersect (main.ml:247-256): f1 then f2, the first hit clips t_max for the tree */
    if (MODE == PT_MODE_ARRAY && sc.n_floor > 0) {
      for (int f = 0; f < sc.n_floor; ++f) {
        V3 fa, fb, fc;
        if (f < sv.n_floor_lds) { /* (wave-uniform) */
          const __attribute__((address_space(3))) double* tv = sv.floor_lds + f * 10;
          fa = v3(tv[0], tv[1], tv[2]);
          fb = v3(tv[3], tv[4], tv[5]);
          fc = v3(tv[6], tv[7], tv[8]);
        } else {
          const double* tv = sv.tri + (size_t)(sc.n_slots + f) * 10;
          fa = pt_load_v3(tv);
          fb = pt_load_v3(tv + 3);
          fc = pt_load_v3(tv + 6);
        }
        double t, u, v;
        if (COUNT) c_floor++;
        if (pt_triangle_intersect(fa, fb, fc, o, d, 0.0, PT_MAX_FINITE, &t, &u, &v)) {
          r.t = t;
          r.u = u;
          r.v = v;
          r.slot = sc.n_slots + f;
          break;
        }
      }
    }
    /* packet constants of spheres_intersect_aux (lib.rs:115-117): a is the UNFUSED scalar dot */
    qa = 0.0;
    one_over_a = 0.0;
    if (MODE == PT_MODE_SIMD) {
      qa = d.x * d.x + d.y * d.y + d.z * d.z;
      one_over_a = 1.0 / qa;
    }
    sp = 0;
    if (OCT) skip_off = dirs * sv.n_nodes;
    node = SWZ ? sv.swz_root : ((G32 && !OCT && sv.has_top) ? PT_TOP_FLAG : 0u); /* the root (slot 0 of the top image) */
    walking = sc.n_nodes > 0;
    if (TAGGED && sc.n_nodes <= 0) node = PT_SWZ_END;
    if (OTAG) node = skip_off; /* the root's record of this ray's octant */
    if (OTAG && sc.n_nodes <= 0) node = PT_OCT_END;
    lkx = 0u;
    leaf_first = 0;
    leaf_n = 0;
    if (FILT) update_t32();
  }
  /* the closest hit so far as the filter sees it; called whenever r.t may have changed */
  __device__ __forceinline__ void update_t32() {
    t32 = (float)r.t; /* PT_MAX_FINITE -> +inf */
    c2 = c2base + (t32 < 0x1p120f ? t32 * PT_MARGIN_T : 0.0f);
  }

#if PT_WALK_ASM
  /* The node loop of the LDS walk (TAGGED), written out in gfx950 assembly.  Measured (tools/README.md "what binds the node
   * loop", profiles/r05_visit_sensitivity.txt): three more vector instructions per visit cost the headline frame 0.5 %, six more
   * SCALAR ones 1.6 % -- with four waves per SIMD, each issuing at most one instruction per turn, the walk is bound by the number
   * of instructions a wave issues per visit, of whatever kind, and of the compiler's 56 (21 vector + 6 LDS + 4 waits + ~25 scalar
   * of exec-mask algebra for a loop with a divergent and a uniform exit and a divergent branch inside) the scalar ones are
   * bookkeeping.  Here a visit is 21 vector + 6 LDS + 4 waits + 6 ... 9 scalar instructions: the exec mask IS the set of lanes
   * that still want a node step, a lane leaves by one s_and of exec, and the filter's undecided case is a wave-uniform exit.
   * Same loads, same arithmetic, same select as node_step (read the two side by side).
   * Runs visits until no lane wants one, or -- once a lane has left -- fewer than `wmin` are still walking.  Returns nonzero if it
   * stopped BEFORE the select of a visit because the binary32 filter left a lane undecided: the caller performs that visit with
   * node_step (the binary64 arithmetic lives there) and calls again.  Temporaries live in v120..v127 (the halves of a
   * ds_read2_b32 result cannot be named through an operand): 128-VGPR kernels only (k_bounce). */
  __device__ __forceinline__ unsigned long long walk_asm(int wmin) {
    register uint32_t bx0 asm("v120"), bx1 asm("v121"), by0 asm("v122"), by1 asm("v123"), bz0 asm("v124"), bz1 asm("v125"), lk0 asm("v126"), lk1 asm("v127");
    lk0 = lkx;
    unsigned long long und, sv_, ent_, hit_;
    uint32_t a_, sk_, nr_, tg_, n_;
    float tn_, tf_, p_, q_, r_, s_, m2_;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "s_mov_b64 %[und], 0\n"
        "v_and_b32 %[tg], 3, %[node]\n"
        "v_cmp_eq_u32 vcc, 0, %[tg]\n"
        "s_and_b64 exec, exec, vcc\n"
        "s_cbranch_execz .Ldone%=\n"
        "s_mov_b64 %[ent], exec\n"
        "s_waitcnt lgkmcnt(0)\n" /* (scalar loads return out of order: nothing of the compiler's may be outstanding) */
        ".Lloop%=:\n"
        "v_add_u32 %[a], %[node], %[so]\n"
        "ds_read_u16 %[sk], %[a] offset:44\n"
        "ds_read_u16 %[nr], %[a] offset:60\n"
        "v_add_u32 %[a], %[node], %[sx]\n"
        "ds_read2_b32 v[120:121], %[a] offset1:1\n"
        "v_add_u32 %[a], %[node], %[sy]\n"
        "ds_read2_b32 v[122:123], %[a] offset1:1\n"
        "v_add_u32 %[a], %[node], %[sz]\n"
        "ds_read2_b32 v[124:125], %[a] offset1:1\n"
        "ds_read2_b32 v[126:127], %[node] offset0:9 offset1:10\n"
        "s_waitcnt lgkmcnt(3)\n"
        "v_fma_f32 %[tn], %[bx0], %[fix], %[fnx]\n"
        "v_fma_f32 %[tf], %[bx1], %[fix], %[fnx]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "v_fma_f32 %[p], %[by0], %[fiy], %[fny]\n"
        "v_fma_f32 %[q], %[by1], %[fiy], %[fny]\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_fma_f32 %[r], %[bz0], %[fiz], %[fnz]\n"
        "v_fma_f32 %[s], %[bz1], %[fiz], %[fnz]\n"
        "v_max_f32 %[tn], %[tn], %[p]\n"
        "v_min_f32 %[tf], %[tf], %[q]\n"
        "v_max3_f32 %[tn], %[tn], %[r], 0\n"
        "v_min3_f32 %[tf], %[tf], %[s], %[t32]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_fma_f32 %[m2], %[lk1], %[k2], %[c2]\n"
        "v_sub_f32 %[tf], %[tf], %[tn]\n"
        "v_cmp_ge_f32_e64 %[hit], %[tf], %[m2]\n"
        "v_cmp_nge_f32_e64 vcc, |%[tf]|, %[m2]\n"
        "s_cbranch_vccnz .Lund%=\n"
        "v_cndmask_b32_e64 %[node], %[sk], %[nr], %[hit]\n"
        "v_and_b32 %[tg], 3, %[node]\n"
        "v_cmp_eq_u32 vcc, 0, %[tg]\n"
        "s_and_b64 exec, exec, vcc\n"
        "s_cbranch_execz .Ldone%=\n"
        "s_cmp_eq_u64 exec, %[ent]\n"
        "s_cbranch_scc1 .Lloop%=\n"
        "s_bcnt1_i32_b64 %[n], exec\n"
        "s_cmp_ge_u32 %[n], %[wmin]\n"
        "s_cbranch_scc1 .Lloop%=\n"
        "s_branch .Ldone%=\n"
        ".Lund%=:\n"
        "s_mov_b64 %[und], vcc\n"
        ".Ldone%=:\n"
        "s_mov_b64 exec, %[sv]\n"
        : [node] "+v"(node), [lk0] "+v"(lk0), [lk1] "=&v"(lk1), [bx0] "=&v"(bx0), [bx1] "=&v"(bx1), [by0] "=&v"(by0), [by1] "=&v"(by1),
          [bz0] "=&v"(bz0), [bz1] "=&v"(bz1), [und] "=&s"(und), [sv] "=&s"(sv_), [ent] "=&s"(ent_), [hit] "=&s"(hit_), [n] "=&s"(n_),
          [a] "=&v"(a_), [sk] "=&v"(sk_), [nr] "=&v"(nr_), [tg] "=&v"(tg_), [tn] "=&v"(tn_), [tf] "=&v"(tf_), [p] "=&v"(p_), [q] "=&v"(q_),
          [r] "=&v"(r_), [s] "=&v"(s_), [m2] "=&v"(m2_)
        : [so] "v"(skip_off), [sx] "v"(sel_x), [sy] "v"(sel_y), [sz] "v"(sel_z), [fix] "v"(fix), [fiy] "v"(fiy), [fiz] "v"(fiz), [fnx] "v"(fnx),
          [fny] "v"(fny), [fnz] "v"(fnz), [t32] "v"(t32), [k2] "v"(k2), [c2] "v"(c2), [wmin] "s"(wmin)
        : "vcc", "scc");
    lkx = lk0;
    return und;
  }
#endif
  /* Visit `node`: bbox test against the closest hit so far, then descend / hold the leaf / pop.  The traversal
   * stack (far children only) lives in LDS, one column per lane (conflict-free ds_write / ds_read).  A far child's
   * bbox is tested when it is POPPED, against the closest hit so far -- exactly the t_max the reference's recursion
   * passes (shape_tree.ml:210-216). */
  /* the reference's own arithmetic on the binary64 node `np` (the filter could not decide): 1 / d again -- the same three
   * divisions as Ray.create, opaque to the optimiser, or it hoists them out of the walk and keeps six more registers live
   * across the hot loop */
  __device__ __forceinline__ bool slab64(const double* nb /* mn.xyz, mx.xyz */) const {
    double qx = d.x, qy = d.y, qz = d.z;
    asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));
    const V3 inv64 = v3(1.0 / qx, 1.0 / qy, 1.0 / qz);
    return (!(pt_isfinite(inv64.x) && pt_isfinite(inv64.y) && pt_isfinite(inv64.z))) ? pt_slab_hit_exact(nb, o, inv64, 0.0, r.t)
                                                                                     : pt_slab_hit_fast<ORIGIN_ZERO>(nb, o, inv64, 0.0, r.t);
  }
  __device__ __forceinline__ bool slab64(const PtNode* np) const { return slab64(np->mn); }
  /* What the generic u = hi - lo cannot decide although it is certain: ONE axis k whose slab interval [near_k, far_k] lies
   * inside the other two axes' intervals and inside [0, t_max] by the margin.  Then the reference's lo IS near_k and its hi IS
   * far_k, and near_k <= far_k holds by construction (the min and the max of the same two products): a hit, however small
   * far_k - near_k is -- and it is exactly 0 for a box that is flat along k (an axis-aligned wall's triangles: cornell's whole
   * room -- 4.7 % of its box tests were undecided and 31 % of its wave steps ran the binary64 fallback), tiny for thin boxes.
   * Each one-sided gap carries the same error bound as u, so the same margin decides it.  false = still undecided (near-ties
   * between DIFFERENT axes, NaNs): binary64. */
  __device__ __forceinline__ bool nested_hit(float tnx, float tfx, float tny, float tfy, float tnz, float tfz, float m2) const {
    const float lx = __builtin_fmaxf(__builtin_fmaxf(tny, tnz), 0.0f), hx = __builtin_fminf(__builtin_fminf(tfy, tfz), t32);
    const float ly = __builtin_fmaxf(__builtin_fmaxf(tnx, tnz), 0.0f), hy = __builtin_fminf(__builtin_fminf(tfx, tfz), t32);
    const float lz = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), 0.0f), hz = __builtin_fminf(__builtin_fminf(tfx, tfy), t32);
    return (tnx - lx >= m2 && hx - tfx >= m2) || (tny - ly >= m2 && hy - tfy >= m2) || (tnz - lz >= m2 && hz - tfz >= m2);
  }
  /* Bbox.is_hit of `nd` against the closest hit so far + the node's links (a, b, real slot count) */
  __device__ __forceinline__ bool test_box(const PtSceneView& sv, uint32_t nd, uint32_t& na, uint32_t& nb, uint32_t& n_real, bool active = true,
                                           bool* defer_undecided = nullptr /* OTAG: set instead of evaluating the binary64 test here */) const {
    const double t_min = 0.0;
    bool hit;
    if (FILT) {
      uint4 w0, w1;
      float mag;
      if (OTAG) { /* nd is the record's number in the whole image; the octant's tagged record: near xyz, far xyz, hit link, miss link */
        const uint4* p = (const uint4*)(sv.nodes32o + (uint32_t)(nd << 5)); /* (a 32-bit offset beside the scalar base: one shift ahead of the loads) */
        const uint4 r0 = p[0];
        uint4 r1 = p[1];
#if PT_DIAG_EXTRA_LOADS /* diagnostic builds only: what one / two more 16-byte loads per node visit cost (the record's own line: no new misses) */
        {
          const uint4 x0 = p[(nd & 1u) ? -1 : 2];
          if (x0.x == 0x7fc12345u && x0.w == 0x12345u) r1.x ^= 1u; /* (never true: keeps the load) */
#if PT_DIAG_EXTRA_LOADS > 1
          const uint4 x1 = p[(nd & 1u) ? -2 : 3];
          if (x1.x == 0x7fc12345u && x1.w == 0x12345u) r1.y ^= 1u;
#endif
        }
#endif
        oct_link = r1.z;
        oct_skip = r1.w;
        n_real = (r1.z >> PT_OCT_LEAF_FIRST_BITS) & PT_OCT_LEAF_REAL_MAX; /* (meaningful for a leaf; na / nb are not used by this walk) */
        na = r1.z & ((1u << PT_OCT_LEAF_FIRST_BITS) - 1u);
        nb = 0u;
        const float nx_ = __uint_as_float(r0.x), ny_ = __uint_as_float(r0.y), nz_ = __uint_as_float(r0.z);
        const float fx_ = __uint_as_float(r0.w), fy_ = __uint_as_float(r1.x), fz_ = __uint_as_float(r1.y);
        const float tnx = __builtin_fmaf(nx_, fix, fnx), tny = __builtin_fmaf(ny_, fiy, fny), tnz = __builtin_fmaf(nz_, fiz, fnz);
        const float tfx = __builtin_fmaf(fx_, fix, fnx), tfy = __builtin_fmaf(fy_, fiy, fny), tfz = __builtin_fmaf(fz_, fiz, fnz);
        const float a = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
        const float b = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
        const float u = __builtin_fminf(b, t32) - __builtin_fmaxf(a, 0.0f);
        const float mg = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(nx_), __builtin_fabsf(ny_)), __builtin_fabsf(nz_)),
                                         __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fx_), __builtin_fabsf(fy_)), __builtin_fabsf(fz_)));
        const float m2 = __builtin_fmaf(mg, k2, c2);
#if PT_FILTER_DEBUG
        dbg_u = u;
        dbg_m2 = m2;
#endif
        hit = u >= m2;
        if (defer_undecided) { /* PT_OCT_BATCH_FALLBACK: no branch here at all -- the caller tags the lane (node_step) */
          *defer_undecided = active && !(__builtin_fabsf(u) >= m2);
          if (COUNT && *defer_undecided) n_undecided++;
          return hit;
        }
        if (active && !(__builtin_fabsf(u) >= m2)) { /* (as below: one divergent branch; a ray the filter does not apply to carries m2 = NaN) */
          if (COUNT) {
            n_undecided++;
            if (pt_lane() == __ffsll((long long)__ballot(1)) - 1) n_wave_fallbacks++;
          }
          hit = slab64(sv.nodes + (nd - skip_off)); /* (the canonical node: the octant's base off again) */
        }
        return hit;
      }
#if PT_SWZ_SIGNSEL
      if (SWZ) { /* nd is the node's absolute LDS address; the ray's sign-selected (near, far) bounds: see the layout */
        const pt_f2 bx = *(const __attribute__((address_space(3))) pt_f2*)PT_LDS_AT(nd + sel_x);
        const pt_f2 by = *(const __attribute__((address_space(3))) pt_f2*)PT_LDS_AT(nd + sel_y);
        const pt_f2 bz = *(const __attribute__((address_space(3))) pt_f2*)PT_LDS_AT(nd + sel_z);
        const pt_u2 lk = *(const __attribute__((address_space(3))) pt_u2*)PT_LDS_AT(nd + PT_SWZ_OFF_LINKS);
        lkx = lk.x;
        na = lk.x & 0xffffu;
        nb = (lk.x >> 16) | ((lk.y & 3u) << 30);
        n_real = lk.x >> 16;
        const float tnx = __builtin_fmaf(bx.x, fix, fnx), tfx = __builtin_fmaf(bx.y, fix, fnx);
        const float tny = __builtin_fmaf(by.x, fiy, fny), tfy = __builtin_fmaf(by.y, fiy, fny);
        const float tnz = __builtin_fmaf(bz.x, fiz, fnz), tfz = __builtin_fmaf(bz.y, fiz, fnz);
        const float a = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
        const float b = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
        const float u = __builtin_fminf(b, t32) - __builtin_fmaxf(a, 0.0f);
        const float m2 = __builtin_fmaf(__uint_as_float(lk.y), k2, c2);
        hit = u >= m2;
        /* one divergent branch, no wave-uniform pre-check: two scalar instructions per visit instead of five (at 4 waves per
         * SIMD the scalar instructions of a visit -- 27 against 22 vector ones -- are no longer free) */
        if (active && !(__builtin_fabsf(u) >= m2)) {
          const bool nested = nested_hit(tnx, tfx, tny, tfy, tnz, tfz, m2);
          if (nested) hit = true;
          else {
            if (COUNT) n_undecided++;
            const uint32_t k64 = (nd - sv.swz_root) / PT_SWZ_NODE_BYTES;
            hit = sv.nodes64 ? slab64(sv.nodes64 + (size_t)k64 * 6u) : slab64(sv.nodes + k64); /* (wave-uniform choice) */
          }
          if (COUNT && __ballot(!nested) != 0 && pt_lane() == __ffsll((long long)__ballot(1)) - 1) n_wave_fallbacks++;
        }
        return hit;
      }
#endif
      if (SWZ) { /* nd is the node's BYTE offset in the LDS image */
        w0 = *(const uint4*)(sv.swz_nodes + nd);
        w1 = *(const uint4*)(sv.swz_nodes + nd + 16);
        na = w1.z & 0xffffu;
        nb = (w1.z >> 16) | ((w1.w & 3u) << 30);
        n_real = w1.z >> 16; /* meaningful for leaves only */
        mag = __uint_as_float(w1.w);
      } else { /* nd is the node's index; 32-byte global image: six binary32 bounds, a, b (leaf b: count | real << 15 | tag) */
        if (nd & PT_TOP_FLAG) { /* ... or PT_TOP_FLAG | byte offset into the LDS copy of the tree's top: same words */
          const uint4* p = (const uint4*)(sv.top + (nd & (PT_TOP_FLAG - 1u)));
          w0 = p[0];
          w1 = p[1];
        } else {
          const uint4* p = (const uint4*)(sv.nodes32 + (size_t)nd * 32u);
          w0 = p[0];
          w1 = p[1];
        }
        na = w1.z;
        const bool leaf = (w1.w >> 30) == PT_NODE_LEAF_AXIS;
        n_real = (w1.w >> 15) & 0x7fffu;
        nb = leaf ? ((w1.w & 0x7fffu) | (PT_NODE_LEAF_AXIS << 30)) : w1.w;
        /* the node's own magnitude, from the bounds just loaded (the LDS image stores it) */
        mag = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(__uint_as_float(w0.x)), __builtin_fabsf(__uint_as_float(w0.y))),
                                              __builtin_fmaxf(__builtin_fabsf(__uint_as_float(w0.z)), __builtin_fabsf(__uint_as_float(w0.w)))),
                              __builtin_fmaxf(__builtin_fabsf(__uint_as_float(w1.x)), __builtin_fabsf(__uint_as_float(w1.y)))) * 1.000001f;
      }
      const float t0x = __builtin_fmaf(__uint_as_float(w0.x), fix, fnx), t1x = __builtin_fmaf(__uint_as_float(w0.w), fix, fnx);
      const float t0y = __builtin_fmaf(__uint_as_float(w0.y), fiy, fny), t1y = __builtin_fmaf(__uint_as_float(w1.x), fiy, fny);
      const float t0z = __builtin_fmaf(__uint_as_float(w0.z), fiz, fnz), t1z = __builtin_fmaf(__uint_as_float(w1.y), fiz, fnz);
      const float a = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
      const float b = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
      const float u = __builtin_fminf(b, t32) - __builtin_fmaxf(a, 0.0f);
      const float m2 = __builtin_fmaf(mag, k2, c2);
      hit = u >= m2;
      /* (a ray the filter does not apply to carries m2 = NaN: neither comparison holds, every test of it is undecided.)
       * One comparison |u| >= m2 (u == -m2 exactly counts as a decided miss: it is one, m2 is twice the error bound) and one
       * divergent branch, no wave-uniform pre-check -- see the layout-3 branch above */
      if (active && !(__builtin_fabsf(u) >= m2)) {
        /* (no nested_hit here: on a mesh walked from HBM / L2 it settles 7 % of the undecided tests and its twenty
         * instructions in 22 % of the wave steps cost 3 % of the frame) */
        if (COUNT) {
          n_undecided++;
          if (pt_lane() == __ffsll((long long)__ballot(1)) - 1) n_wave_fallbacks++;
        }
        /* the reference's arithmetic, on the binary64 node (global memory: L2-resident, rarely read) */
        hit = slab64(sv.nodes + (SWZ ? nd / PT_SWZ_NODE_BYTES
                                     : ((nd & PT_TOP_FLAG) ? *(const uint32_t*)(sv.top + (nd & (PT_TOP_FLAG - 1u)) + 48) : nd)));
      }
    } else {
      const PtNode* np = sv.nodes + nd;
      hit = exact_slab ? pt_slab_hit_exact(np->mn, o, inv, t_min, r.t) : pt_slab_hit_fast<ORIGIN_ZERO>(np->mn, o, inv, t_min, r.t);
      na = np->a;
      nb = np->b;
      n_real = np->pad[0];
    }
    return hit;
  }

  __device__ __forceinline__ void node_step(const PtSceneView& sv, StackT* stack, unsigned long long& c_nodes,
                                            unsigned long long& c_prims) {
    if (COUNT && (PT_DIAG == 0 || (PT_DIAG <= 2 && !ORIGIN_ZERO))) c_nodes++;
    bool descend = false;
    uint32_t na, nb, n_real;
    if (OTAG) { /* the tagged per-octant record: a hit leads to word 6 (near child, or "holds a leaf"), a miss to word 7 */
      if (PT_OCT_BATCH_FALLBACK) { /* an undecided test: the lane steps out of the walk with its node tagged (resolve_pending) */
        bool und = false;
        const bool hit = test_box(sv, node, na, nb, n_real, true, &und);
        if (COUNT && PT_DIAG == 0 && !und && hit && (oct_link >> 30) == 1u) c_prims += (unsigned long long)(MODE == PT_MODE_SIMD ? ((n_real + 3u) & ~3u) : n_real);
        node = und ? (node | PT_OCT_PENDING) : (hit ? oct_link : oct_skip);
        return;
      }
      const bool hit = test_box(sv, node, na, nb, n_real);
      if (COUNT && PT_DIAG == 0 && hit && (oct_link >> 30) == 1u) c_prims += (unsigned long long)(MODE == PT_MODE_SIMD ? ((n_real + 3u) & ~3u) : n_real);
      node = hit ? oct_link : oct_skip;
      return;
    }
    /* threaded image: where to go once this subtree is done (issued beside the node's own reads) */
    constexpr bool THREAD32 = G32;
    uint32_t skip;
#if PT_SWZ_SIGNSEL
    if (SWZ) {
      skip = (uint32_t)*(const __attribute__((address_space(3))) uint16_t*)PT_LDS_AT(node + skip_off + PT_SWZ_OFF_SKIP);
      if (TAGGED) asm("" : "+v"(skip)); /* (a zero-extended 16-bit load the select below need not mask again) */
    }
#else
    if (SWZ) skip = (uint32_t)*(const uint16_t*)(sv.swz_nodes + node + skip_off);
#endif
    else if (!THREAD32) skip = 0u;
    else if (node & PT_TOP_FLAG) { /* top image: 16-bit byte offsets, a top node's successor is a top node */
      const uint32_t s16 = (uint32_t)*(const uint16_t*)(sv.top + (node & (PT_TOP_FLAG - 1u)) + 32u + 2u * dirs);
      skip = s16 == 0xffffu ? 0xffffffffu : (PT_TOP_FLAG | s16);
    } else skip = sv.skip32[(size_t)node * 8u + dirs];
#if PT_SWZ_NEAR
    if (SWZ) {
#if PT_SWZ_SIGNSEL
      uint32_t near_c = (uint32_t)*(const __attribute__((address_space(3))) uint16_t*)PT_LDS_AT(node + skip_off + PT_SWZ_OFF_NEAR);
      if (TAGGED) asm("" : "+v"(near_c));
#else
      const uint32_t near_c = (uint32_t)*(const uint16_t*)(sv.swz_nodes + node + skip_off + 16u);
#endif
      const bool hit = test_box(sv, node, na, nb, n_real);
      if (TAGGED) {
#if PT_DIAG_VISIT_VALU || PT_DIAG_VISIT_LDS || PT_DIAG_VISIT_SALU
        for (int k_ = 0; k_ < PT_DIAG_VISIT_VALU; ++k_) asm volatile("v_add_u32 %0, 1, %0" : "+v"(lkx_diag));
        for (int k_ = 0; k_ < PT_DIAG_VISIT_LDS; ++k_) lkx_diag += (uint32_t)*(const __attribute__((address_space(3))) uint16_t*)PT_LDS_AT(node + skip_off + PT_SWZ_OFF_SKIP + 2u * (uint32_t)(k_ + 1));
        for (int k_ = 0; k_ < PT_DIAG_VISIT_SALU; ++k_) { uint32_t t_; asm volatile("s_mov_b32 %0, 1" : "=s"(t_)); }
#endif
        /* an inner node that was hit: its near child; a leaf that was hit: what follows it, tagged "holds a leaf"; a miss: what
         * follows this subtree (PT_SWZ_END: nothing) */
        if (COUNT && PT_DIAG == 0 && hit && (near_c & PT_SWZ_TAG_LEAF)) c_prims += (unsigned long long)(MODE == PT_MODE_SIMD ? ((n_real + 3u) & ~3u) : n_real);
        node = hit ? near_c : skip;
        return;
      }
      const bool leaf_hit = hit && near_c == PT_SWZ_LEAF;
      if (leaf_hit) {
        leaf_first = (int)na;
        leaf_n = (int)n_real; /* real slots; the NaN padding (main.ml:185) can never be selected */
        if (COUNT && PT_DIAG == 0) c_prims += (unsigned long long)(MODE == PT_MODE_SIMD ? ((n_real + 3u) & ~3u) : n_real);
      }
      /* an inner node that was hit: its near child; else (miss, or a leaf taken) what follows this subtree */
      const uint32_t nx = (hit && !leaf_hit) ? near_c : skip;
      if (nx == PT_SWZ_END) walking = false;
      else node = nx;
      return;
    }
#endif
    const bool hit = test_box(sv, node, na, nb, n_real);
    if (hit) {
      const uint32_t axis = nb >> 30;
      if (axis == PT_NODE_LEAF_AXIS) {
        leaf_first = (int)na;
        leaf_n = (int)n_real; /* real slots; the NaN padding (main.ml:185) can never be selected */
        /* Leaf.length incl. padding: Simd_leaf pads to a multiple of 4 (main.ml:179-186); the filter image keeps the real count */
        if (COUNT && PT_DIAG == 0) c_prims += (unsigned long long)(SWZ ? (MODE == PT_MODE_SIMD ? ((n_real + 3u) & ~3u) : n_real) : (nb & 0x3fffffffu));
      } else {
        /* Branch: near child first (shape_tree.ml:209), far child deferred */
        const uint32_t lhs = na, rhs = nb & 0x3fffffffu;
        const bool lhs_first = (dirs >> axis) & 1u;
        if (!SWZ && !THREAD32) {
          PT_STACK_PUSH(stack, sp, lhs_first ? rhs : lhs);
          ++sp;
        }
        node = lhs_first ? lhs : rhs;
        descend = true;
      }
    }
    if (!descend) {
      /* the next node's bbox is tested on the NEXT visit, i.e. after this leaf's packet has been
       * intersected and r.t shrunk -- the t_max the reference passes to the far child */
      if (SWZ) {
        if (skip == PT_SWZ_END) walking = false;
        else node = skip;
      } else if (THREAD32) {
        if (skip == 0xffffffffu) walking = false;
        else node = skip;
      } else if (sp == 0) walking = false;
      else {
        --sp;
        node = PT_STACK_POP(stack, sp);
      }
    }
  }

  /* PT_OCT_BATCH_FALLBACK: the binary64 box test of every lane of the wave whose walk stepped out on an undecided filter test, together;
   * then the visit's select, as node_step would have made it (oct_link / oct_skip are still that visit's words) */
  __device__ __forceinline__ void resolve_pending(const PtSceneView& sv, unsigned long long& c_prims) {
    if (pending()) {
      const uint32_t nd = node & ~PT_OCT_PENDING;
      if (COUNT && pt_lane() == __ffsll((long long)__ballot(1)) - 1) n_wave_fallbacks++;
      const bool hit = slab64(sv.nodes + (nd - skip_off)); /* (the canonical node: the octant's base off again) */
      if (COUNT && PT_DIAG == 0 && hit && (oct_link >> 30) == 1u) {
        const uint32_t n_real = (oct_link >> PT_OCT_LEAF_FIRST_BITS) & PT_OCT_LEAF_REAL_MAX;
        c_prims += (unsigned long long)(MODE == PT_MODE_SIMD ? ((n_real + 3u) & ~3u) : n_real);
      }
      node = hit ? oct_link : oct_skip;
    }
  }
  /* Leaf.intersect on the held leaf (caller checks leaf_n > 0) */
  __device__ __forceinline__ void packet(const PtSceneView& sv, unsigned long long& c_nodes, unsigned long long& c_floor) {
    const double t_min = 0.0;
    if (TAGGED) { /* the leaf's word, as the visit that took it loaded it */
      leaf_first = (int)(lkx & 0xffffu);
      leaf_n = (int)(lkx >> 16);
      node &= ~PT_SWZ_TAG_LEAF; /* -> the node that follows the leaf, or PT_SWZ_END */
    }
    if (OTAG) { /* `node` IS the leaf's word; the walk goes on where the leaf's miss link points */
      leaf_first = (int)(node & ((1u << PT_OCT_LEAF_FIRST_BITS) - 1u));
      leaf_n = (int)((node >> PT_OCT_LEAF_FIRST_BITS) & PT_OCT_LEAF_REAL_MAX);
      node = oct_skip;
    }
    if (MODE == PT_MODE_SIMD) {
      /* spheres_intersect_aux, lib.rs:102-178, one packet lane per step, split in two so the wave stays
       * dense: SCAN (cheap, every lane: f, c, b', discriminant) runs until the lane meets a slot whose
       * discriminant is >= +0; only then do the lanes that found one run the HEAVY part (sqrt, divide)
       * together.  Testing slot after slot in lockstep would execute ~50 sqrt/div instructions per slot with
       * one lane in eight active.  Slots are still visited in order, so `t <= t_found` ties resolve alike. */
#if PT_PACKET_DEFER
      /* pass 1, lockstep over the slots: only the discriminant's sign; pass 2: the roots of the candidates, in slot
       * order.  The discriminant does not depend on the closest hit so far, so deferring the roots changes nothing. */
      for (int base = 0; base < leaf_n; base += 32) {
        const int m = (leaf_n - base) < 32 ? (leaf_n - base) : 32;
        uint32_t cand = 0;
        for (int k = 0; k < m; ++k) {
          if (COUNT && PT_DIAG == 3 && !ORIGIN_ZERO) {
            c_nodes++;
            PT_DIAG_WAVE_SLOTS(c_floor);
          }
          const double* s = sv.sph + (size_t)(leaf_first + base + k) * PT_SPH_STRIDE(SWZ);
          const double fx = ORIGIN_ZERO ? s[0] : s[0] - o.x, fy = ORIGIN_ZERO ? s[1] : s[1] - o.y,
                       fz = ORIGIN_ZERO ? s[2] : s[2] - o.z;
          const double bp_over_a = pt_fma(fx, d.x, pt_fma(fy, d.y, fz * d.z)) * one_over_a;
          const double wx = pt_fma(d.x, bp_over_a, -fx);
          const double wy = pt_fma(d.y, bp_over_a, -fy);
          const double wz = pt_fma(d.z, bp_over_a, -fz);
          const double disc = (s[3] * s[3]) - pt_fma(wx, wx, pt_fma(wy, wy, wz * wz));
#if PT_SCAN_ADDC
          /* "neither NaN nor sign bit set" (lib.rs:162-166) is ONE unsigned comparison of the bit pattern (+0 ... +inf), and the
           * candidate mask takes the outcome as the carry of cand + cand: two vector instructions instead of five; the mask
           * comes out in reverse order and is turned round once per leaf */
          {
            const unsigned long long fm = __builtin_amdgcn_ballot_w64((unsigned long long)__double_as_longlong(disc) <= 0x7ff0000000000000ull);
            unsigned long long co_;
            asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(cand), "=s"(co_) : "s"(fm));
          }
#else
          if ((disc == disc) && !pt_signbit(disc)) cand |= 1u << k;
#endif
        }
#if PT_SCAN_ADDC
        cand = __brev(cand) >> (32 - m); /* slot k -> bit k (m >= 1) */
#endif
        while (cand != 0) {
          if (COUNT && PT_DIAG == 4 && !ORIGIN_ZERO) {
            c_nodes++;
            PT_DIAG_WAVE_SLOTS(c_floor);
          }
          const int k = __ffs((int)cand) - 1;
          cand &= cand - 1u;
          const double* s = sv.sph + (size_t)(leaf_first + base + k) * PT_SPH_STRIDE(SWZ);
          const double fx = ORIGIN_ZERO ? s[0] : s[0] - o.x, fy = ORIGIN_ZERO ? s[1] : s[1] - o.y,
                       fz = ORIGIN_ZERO ? s[2] : s[2] - o.z;
          const double r2 = s[3] * s[3];
          const double c = pt_fma(fx, fx, pt_fma(fy, fy, fz * fz)) - r2;
          const double bp = pt_fma(fx, d.x, pt_fma(fy, d.y, fz * d.z));
          const double bp_over_a = bp * one_over_a;
          const double wx = pt_fma(d.x, bp_over_a, -fx);
          const double wy = pt_fma(d.y, bp_over_a, -fy);
          const double wz = pt_fma(d.z, bp_over_a, -fz);
          const double disc = r2 - pt_fma(wx, wx, pt_fma(wy, wy, wz * wz));
          const double q_rhs = pt_sqrt(qa * disc);
          const double qq = pt_signbit(bp) ? (bp - q_rhs) : (bp + q_rhs);
          const double t = pt_signbit(c) ? (qq * one_over_a) : (c / qq);
          if (!(t < t_min) && t <= r.t) {
            r.t = t;
            r.slot = leaf_first + base + k;
          }
        }
      }
#else
      int k = 0;
      while (k < leaf_n) {
        double c = 0.0, bp = 0.0, disc = 0.0;
        bool found = false;
        while (k < leaf_n && !found) {
          if (COUNT && PT_DIAG == 3 && !ORIGIN_ZERO) {
            c_nodes++;
            PT_DIAG_WAVE_SLOTS(c_floor);
          }
          const double* s = sv.sph + (size_t)(leaf_first + k) * PT_SPH_STRIDE(SWZ);
          const double fx = ORIGIN_ZERO ? s[0] : s[0] - o.x, fy = ORIGIN_ZERO ? s[1] : s[1] - o.y,
                       fz = ORIGIN_ZERO ? s[2] : s[2] - o.z; /* f = center - origin */
          const double r2 = s[3] * s[3];
          c = pt_fma(fx, fx, pt_fma(fy, fy, fz * fz)) - r2;
          bp = pt_fma(fx, d.x, pt_fma(fy, d.y, fz * d.z));
          const double bp_over_a = bp * one_over_a;
          const double wx = pt_fma(d.x, bp_over_a, -fx);
          const double wy = pt_fma(d.y, bp_over_a, -fy);
          const double wz = pt_fma(d.z, bp_over_a, -fz);
          const double wq = pt_fma(wx, wx, pt_fma(wy, wy, wz * wz));
          disc = r2 - wq;
          /* lanes whose discriminant has its sign bit set (or is NaN) end up NaN (lib.rs:162-166) */
          found = (disc == disc) && !pt_signbit(disc);
          ++k;
        }
        if (COUNT && PT_DIAG == 4 && !ORIGIN_ZERO) {
          if (found) c_nodes++;
          if (__ballot(found)) PT_DIAG_WAVE_SLOTS(c_floor);
        }
        if (found) {
          const double q_rhs = pt_sqrt(qa * disc);
          const double qq = pt_signbit(bp) ? (bp - q_rhs) : (bp + q_rhs);
          const double t = pt_signbit(c) ? (qq * one_over_a) : (c / qq);
          /* not (t < t_min), not (t > t_max), then `t <= t_found` (last index wins ties, lib.rs:169-177) */
          if (!(t < t_min) && t <= r.t) {
            r.t = t;
            r.slot = leaf_first + k - 1;
          }
        }
      }
#endif
    } else {
      /* Array_leaf.intersect, shape_tree.ml:299-311: shrinking t_max, later element wins ties */
#if PT_LEAF_PREFETCH
      if (!SWZ && sv.kind == nullptr) {
        /* a triangle-only scene walked from HBM / L2 (a mesh): the leaf's triangles are 80-byte records that mostly miss
         * every cache level above L2, and testing them one after the other made a leaf visit a chain of up to
         * length_cutoff dependent round trips -- the longest waits of the whole walk.  The next triangle's nine
         * coordinates are requested before the current one is tested; same tests, same order. */
        const double* tv = sv.tri + (size_t)leaf_first * 10;
        V3 a = pt_load_v3(tv), b = pt_load_v3(tv + 3), c = pt_load_v3(tv + 6);
        for (int k = 0; k < leaf_n; ++k) {
          const double* nx = sv.tri + (size_t)(leaf_first + (k + 1 < leaf_n ? k + 1 : k)) * 10;
          const V3 na = pt_load_v3(nx), nb = pt_load_v3(nx + 3), nc = pt_load_v3(nx + 6);
          double t, u, v;
          if (pt_triangle_intersect(a, b, c, o, d, t_min, r.t, &t, &u, &v)) {
            r.t = t;
            r.u = u;
            r.v = v;
            r.slot = leaf_first + k;
          }
          a = na; b = nb; c = nc;
        }
        leaf_n = 0;
        if (FILT) update_t32();
        return;
      }
#endif
      for (int k = 0; k < leaf_n; ++k) {
        if (COUNT && PT_DIAG == 3 && !ORIGIN_ZERO) {
          c_nodes++;
          PT_DIAG_WAVE_SLOTS(c_floor);
        }
        const int slot = leaf_first + k;
        if (sv.kind[slot] == PT_SLOT_SPHERE) {
          const double* s = sv.sph + (size_t)slot * PT_SPH_STRIDE(SWZ);
          double t;
          if (pt_sphere_intersect_scalar(v3(s[0], s[1], s[2]), s[3], o, d, t_min, r.t, &t)) {
            r.t = t;
            r.slot = slot;
          }
        } else {
          const double* tv = sv.tri + (size_t)slot * 10;
          double t, u, v;
          if (pt_triangle_intersect(pt_load_v3(tv), pt_load_v3(tv + 3), pt_load_v3(tv + 6), o, d, t_min, r.t, &t, &u,
                                    &v)) {
            r.t = t;
            r.u = u;
            r.v = v;
            r.slot = slot;
          }
        }
      }
    }
    leaf_n = 0;
    if (FILT) update_t32();
  }
};

/* Scene.intersect for ONE ray held by this lane, start to finish.
 * "while-while" traversal: every lane walks nodes until it reaches a leaf (or runs out of nodes); only then do the
 * lanes that hold a leaf test its packet TOGETHER.  Interleaving the two, as the recursive reference does, would
 * run the packet loop for one or two lanes at a time.  The order of node tests and packet tests of each individual
 * ray is unchanged. */
/* Tail cut (threaded LDS scenes only: the whole state of a walk in progress is the node it stands on, the closest hit so
 * far and its slot).  The wave stops a chunk once fewer than `min_active` of its rays are still walking; those rays are
 * not finished: the caller parks their state and resumes them later TOGETHER with the stragglers of its other chunks
 * (k_trace).  A resumed ray continues at `node` with `t` / `slot` restored, i.e. it performs exactly the box and packet
 * tests it had left, in the same order. */
struct PtTailCtl {
  int min_active;   /* in: stop when fewer rays than this are walking (0 = run to completion) */
  bool resume;      /* in: this lane continues a parked walk */
  uint32_t node;    /* in (resume) / out (unfinished): byte offset of the node to visit next */
  double t;         /* in (resume): closest hit so far */
  double u, v;      /* in (resume): its barycentrics (triangle scenes) */
  int slot;
  bool unfinished;  /* out: the ray is still walking */
};
template <int MODE, bool COUNT, bool ORIGIN_ZERO, typename StackT, bool SWZ = false, bool DIV_LOOP = (PT_WALK_LOOP != 0), bool ASM_WALK = false>
__device__ __forceinline__ PtTraceResult pt_trace_ray(const PtSceneDev& sc, const PtSceneView& sv, StackT* stack,
                                                      V3 o, V3 d, unsigned long long& c_nodes,
                                                      unsigned long long& c_prims, unsigned long long& c_floor,
                                                      bool valid = true, PtTailCtl* tc = nullptr,
                                                      unsigned long long* c_filter = nullptr) {
  PtTraverser<MODE, COUNT, ORIGIN_ZERO, StackT, SWZ> tr;
  constexpr int WALK_MIN = SWZ ? PT_WALK_MIN : PT_WALK_MIN_GLOBAL;
  unsigned long long no_count = 0; /* lanes without a ray run begin() on a dummy ray: keep them out of the counters */
  /* a resumed walk has had its floor pre-test (its outcome is part of the parked state): not counted again */
  tr.begin(sc, sv, o, d, (valid && !(tc && tc->resume)) ? c_floor : no_count);
  if (tc && tc->resume) {
    tr.node = tc->node;
    tr.r.t = tc->t;
    tr.r.u = tc->u;
    tr.r.v = tc->v;
    tr.r.slot = tc->slot;
    if (tr.FILT) tr.update_t32(); /* the filter's copy of t: stale, it would pass boxes beyond the restored hit */
  }
  while (valid && tr.alive()) {
    /* DIV_LOOP: the node walk as ONE divergent loop: a lane stays in it while it wants node steps, so the lanes still walking
     * are simply the loop's exec mask (no per-turn ballot of a `want` flag, no `continue`, no exit-reason bookkeeping: the
     * first version of this loop spent ~19 of its ~80 instructions per turn on that).  Same rule as below: once fewer than
     * PT_WALK_MIN lanes are walking and some lane holds a leaf, the pending packets are intersected first.  Half the SCALAR
     * instructions per turn: +3 % time in k_trace at 8 waves per SIMD (scalar issue is not what binds there), -1 % in k_bounce
     * at 4 (with half the waves, a wave busy with scalar bookkeeping is more often the one the vector pipe is waiting for). */
    if constexpr (ASM_WALK && (PT_WALK_ASM != 0) && (PT_DIAG == 0) && !COUNT && PtTraverser<MODE, COUNT, ORIGIN_ZERO, StackT, SWZ>::TAGGED) {
#if PT_WALK_ASM
      /* the node loop in assembly; a visit the binary32 filter cannot decide for some lane is performed here, in binary64 */
      while (tr.walk_asm(WALK_MIN) != 0ull)
        if (tr.wants_node()) tr.node_step(sv, stack, c_nodes, c_prims);
#endif
    } else if constexpr (DIV_LOOP) {
     if (tr.wants_node()) {
      bool leaf_waiting = false; /* wave-uniform: a lane of this wave left the walk holding a leaf */
      /* TAGGED: "a lane has left the walk" instead (scalar: the loop's exec mask against the one it was entered with) -- a lane
       * leaves with a leaf or because its walk is over; when the few that are left come out and no leaf is waiting, the chunk cut
       * below looks at them a little earlier than it used to.  A change of schedule only: which tests a ray performs is its own affair */
      const unsigned long long entered = tr.TAGS ? __builtin_amdgcn_ballot_w64(true) : 0ull;
      for (;;) {
        if (COUNT && PT_DIAG == 1 && !ORIGIN_ZERO) PT_DIAG_WAVE_SLOTS(c_floor);
        tr.node_step(sv, stack, c_nodes, c_prims);
        if (!tr.TAGS && __ballot(tr.leaf_n > 0) != 0) leaf_waiting = true;
        if (!tr.wants_node()) break;
        const unsigned long long now = __builtin_amdgcn_ballot_w64(true);
        if (tr.TAGS) leaf_waiting = now != entered;
        if (leaf_waiting && pt_popc_mask(now) < WALK_MIN) break;
      }
     }
    } else {
    for (;;) {
      /* keep walking while enough lanes still want a node step; once fewer than PT_WALK_MIN do and some lane
       * already holds a leaf, intersect the pending packets first (the stragglers resume afterwards) */
      const unsigned long long wm = tr.wants_node_mask();
      if (wm == 0) break;
      if (pt_popc_mask(wm) < WALK_MIN && tr.holds_leaf_mask() != 0) break;
      if (COUNT && PT_DIAG == 1 && !ORIGIN_ZERO) PT_DIAG_WAVE_SLOTS(c_floor);
      if (!tr.wants_node()) continue;
      tr.node_step(sv, stack, c_nodes, c_prims);
    }
    }
    if (tr.OTAG && PT_OCT_BATCH_FALLBACK) tr.resolve_pending(sv, c_prims); /* (a lane it lets into a leaf takes part in the leaf phase below) */
    if (COUNT && PT_DIAG == 6 && !ORIGIN_ZERO) { /* (diagnostic build: ticks the wave spends in the leaf phase, into c_prims; lane 0's copy is kept) */
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t0_ = __builtin_readcyclecounter();
      if (tr.holds_leaf()) tr.packet(sv, c_nodes, c_floor);
      __builtin_amdgcn_sched_barrier(0);
      c_prims += __builtin_readcyclecounter() - t0_;
    } else
    if (tr.holds_leaf()) tr.packet(sv, c_nodes, c_floor);
    /* nobody holds a leaf here: a safe place to stop.  The ballot sees the rays that are still walking (finished
     * lanes have left the loop), and every one of them sees the same count. */
    if (tc && pt_popc_mask(tr.walking_mask()) < tc->min_active) break;
  }
  if (tc) {
    tc->unfinished = valid && tr.alive(); /* (no lane holds a leaf here) */
    tc->node = tr.node;
  }
  if (COUNT && c_filter) {
    c_filter[0] += tr.n_undecided;
    c_filter[1] += tr.n_wave_fallbacks;
  }
#if PT_DIAG_VISIT_VALU || PT_DIAG_VISIT_LDS || PT_DIAG_VISIT_SALU
  if (tr.lkx_diag + tr.sdiag == 0xfffffff3u) tr.r.t = 0.0; /* (keeps the diagnostic's loads alive) */
#endif
  return tr.r;
}

/* (Round 5, measured and not kept -- profiles/r05_ab_packet_stack.txt: the shared stack below held in the LANES of three registers,
 * v_writelane on a push and v_readlane on a pop instead of the LDS round trip behind a fence and a wave barrier: bit-exact, and
 * +1.5 % on the headline and on cornell -- the camera rays' launches are bound by vector issue (0.85 busy) and the LDS pipe they
 * leave idle carried the stack for nothing; the node's words 6 / 7 handed to the lanes' box test instead of read again per lane: +-0.) */
/* Camera rays: the 64 rays of a wave are one 8x8 pixel tile of one pass, so they walk the tree TOGETHER -- one
 * shared stack of (node, lane mask) instead of 64 private ones.  A ray's own sequence of box tests and packet tests
 * is exactly pt_trace_ray's: its child order depends only on the signs of its direction (shape_tree.ml:201,209), the
 * wave is split into groups of equal signs (almost always one), a ray takes part in a node only if it hit the parent,
 * and a far child is tested when popped, against each ray's own closest hit.  What changes is the cost: node and
 * packet addresses are wave-uniform (LDS broadcasts, scalar control flow), no per-lane stack traffic, and the
 * packet loop runs in lockstep.  wstack: 3 words per level, shared by the wave. */
#ifndef PT_PRIMARY_PACKET
#define PT_PRIMARY_PACKET 1
#endif
template <int MODE, bool COUNT, bool ORIGIN_ZERO, bool SWZ>
__device__ __forceinline__ PtTraceResult pt_trace_packet(const PtSceneDev& sc, const PtSceneView& sv, uint32_t* wstack,
                                                         bool valid, V3 o, V3 d, unsigned long long& c_nodes,
                                                         unsigned long long& c_prims, unsigned long long& c_floor,
                                                         unsigned long long* c_filter = nullptr) {
  const int lane = pt_lane();
  PtTraverser<MODE, COUNT, ORIGIN_ZERO, uint32_t, SWZ> tr;
  unsigned long long no_count = 0; /* lanes without a sample run begin() on a dummy ray: keep them out of the counters */
  tr.begin(sc, sv, valid ? o : v3(0.0, 0.0, 0.0), valid ? d : v3(0.0, 0.0, -1.0), valid ? c_floor : no_count);
  unsigned long long remaining = __ballot(valid && tr.alive());
  while (remaining != 0) {
    /* the lanes that share the first remaining lane's direction signs */
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)tr.dirs, __ffsll((long long)remaining) - 1);
    unsigned long long m = __ballot(tr.dirs == d0) & remaining;
    remaining &= ~m;
    uint32_t node = SWZ ? sv.swz_root : 0u;
    int sp = 0;
    bool act = (m >> lane) & 1ull; /* this lane takes part in `node` */
    for (;;) {
      /* the node's links: one address for the whole wave */
      uint32_t ua, ub, n_real;
      if (SWZ) { /* node = byte offset in the binary32 image */
#if PT_SWZ_SIGNSEL
        const pt_u2 lk = *(const __attribute__((address_space(3))) pt_u2*)PT_LDS_AT(node + PT_SWZ_OFF_LINKS);
        const uint32_t w6 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lk.x);
        const uint32_t w7 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lk.y);
#else
        const unsigned char* nbase = sv.swz_nodes + node;
        const uint32_t w6 = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(nbase + PT_SWZ_OFF_LINKS));
        const uint32_t w7 = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(nbase + PT_SWZ_OFF_LINKS + 4));
#endif
        ua = w6 & 0xffffu;
        ub = (w6 >> 16) | ((w7 & 3u) << 30);
        n_real = w6 >> 16;
      } else {
        const PtNode* np = sv.nodes + node;
        ua = (uint32_t)__builtin_amdgcn_readfirstlane((int)np->a);
        ub = (uint32_t)__builtin_amdgcn_readfirstlane((int)np->b);
        n_real = (uint32_t)__builtin_amdgcn_readfirstlane((int)np->pad[0]);
      }
      bool hit = false;
      if (act) {
        if (COUNT) c_nodes++;
        uint32_t x0, x1, x2;
        hit = tr.test_box(sv, node, x0, x1, x2);
      }
      const unsigned long long hm = __ballot(hit);
      bool descended = false;
      if (hm != 0) {
        const uint32_t axis = ub >> 30;
        if (axis == PT_NODE_LEAF_AXIS) {
          if (COUNT && hit) c_prims += (unsigned long long)(SWZ ? (MODE == PT_MODE_SIMD ? ((n_real + 3u) & ~3u) : n_real) : (ub & 0x3fffffffu));
          if (MODE == PT_MODE_SIMD) {
            /* spheres_intersect_aux (lib.rs:102-178) in lockstep: slot k of the packet for every ray that hit the
             * leaf's box; the roots only where a ray's discriminant is >= +0 (same order per ray as packet()) */
            const double t_min = 0.0;
            for (uint32_t k = 0; k < n_real; ++k) {
              const double* sp4 = sv.sph + (size_t)(ua + k) * PT_SPH_STRIDE(SWZ);
              /* f = center - origin; for camera rays the origin is (+0, +0, +0) and x - (+0) == x bit for bit */
              const double fx = ORIGIN_ZERO ? sp4[0] : sp4[0] - tr.o.x, fy = ORIGIN_ZERO ? sp4[1] : sp4[1] - tr.o.y,
                           fz = ORIGIN_ZERO ? sp4[2] : sp4[2] - tr.o.z;
              const double r2 = sp4[3] * sp4[3];
              const double bp = pt_fma(fx, tr.d.x, pt_fma(fy, tr.d.y, fz * tr.d.z));
              const double bp_over_a = bp * tr.one_over_a;
              const double wx = pt_fma(tr.d.x, bp_over_a, -fx);
              const double wy = pt_fma(tr.d.y, bp_over_a, -fy);
              const double wz = pt_fma(tr.d.z, bp_over_a, -fz);
              const double disc = r2 - pt_fma(wx, wx, pt_fma(wy, wy, wz * wz));
              const bool found = hit && (disc == disc) && !pt_signbit(disc);
              if (__ballot(found) != 0) {
                if (found) {
                  const double c = pt_fma(fx, fx, pt_fma(fy, fy, fz * fz)) - r2;
                  const double q_rhs = pt_sqrt(tr.qa * disc);
                  const double qq = pt_signbit(bp) ? (bp - q_rhs) : (bp + q_rhs);
                  const double t = pt_signbit(c) ? (qq * tr.one_over_a) : (c / qq);
                  if (!(t < t_min) && t <= tr.r.t) {
                    tr.r.t = t;
                    tr.r.slot = (int)(ua + k);
                  }
                }
              }
            }
            if (SWZ) tr.update_t32();
          } else if (hit) {
            tr.leaf_first = (int)ua;
            tr.leaf_n = (int)n_real;
            tr.lkx = ua | (n_real << 16); /* (TAGGED: packet() decodes the leaf's word itself) */
            tr.packet(sv, c_nodes, c_floor);
          }
        } else {
          const uint32_t lhs = ua, rhs = ub & 0x3fffffffu;
          const bool lhs_first = (d0 >> axis) & 1u;
          if (lane == 0) {
            wstack[3 * sp] = lhs_first ? rhs : lhs;
            wstack[3 * sp + 1] = (uint32_t)hm;
            wstack[3 * sp + 2] = (uint32_t)(hm >> 32);
          }
          ++sp;
          node = lhs_first ? lhs : rhs;
          act = hit;
          descended = true;
        }
      }
      if (!descended) {
        if (sp == 0) break;
        --sp;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        node = (uint32_t)__builtin_amdgcn_readfirstlane((int)wstack[3 * sp]);
        const unsigned long long pm = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)wstack[3 * sp + 1]) |
                                      ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)wstack[3 * sp + 2]) << 32);
        act = (pm >> lane) & 1ull;
      }
    }
  }
  if (COUNT && c_filter) {
    c_filter[0] += tr.n_undecided;
    c_filter[1] += tr.n_wave_fallbacks;
  }
  return tr.r;
}

/* Where this workgroup traverses from.  LDS_SCENE: the whole tree and every leaf packet are first copied into LDS
 * behind the traversal stacks (nodes expanded to the swizzled image on the way); ends with a __syncthreads(). */
template <int MODE, bool LDS_SCENE, typename StackT>
__device__ __forceinline__ PtSceneView pt_scene_view(const PtSceneDev& sc, unsigned char* lds_raw, int stack_depth, bool want_top = false) {
  const uint32_t waves_per_block = blockDim.x >> 6;
  PtSceneView sv;
  sv.nodes = sc.nodes;
  sv.skip32 = sc.node_skip32;
  sv.nodes32 = (const unsigned char*)sc.nodes32;
  sv.nodes32o = (const unsigned char*)sc.nodes32o;
  sv.n_nodes = (uint32_t)sc.n_nodes;
  sv.swz_nodes = nullptr;
  sv.swz_root = 0u;
  sv.top = lds_raw;
  sv.has_top = false;
  sv.floor_lds = nullptr;
  sv.n_floor_lds = 0;
  if (!LDS_SCENE && want_top && sc.n_top > 0) { /* the tree's top into LDS (the caller's barrier follows) */
    const uint4* src = (const uint4*)sc.top_nodes;
    uint4* dst = (uint4*)lds_raw;
    for (int k = threadIdx.x; k < sc.n_top * (PT_TOP_NODE_BYTES / 16); k += blockDim.x) dst[k] = src[k];
    sv.has_top = true;
    sv.skip32 = sc.node_skip32_top;
  }
  sv.sph = sc.sph;
  sv.tri = sc.tri;
  sv.kind = sc.slot_kind;
  sv.cat = sc.slot_cat;
  sv.nodes64 = nullptr;
#if PT_LEAF_PREFETCH
  if (!LDS_SCENE && sc.all_triangles) sv.kind = nullptr; /* PtTraverser::packet: the pipelined triangle loop */
#endif
  if (LDS_SCENE) {
    size_t off = ((size_t)waves_per_block * PT_WAVE_STACK_BYTES(LDS_SCENE, StackT, stack_depth) + 63) & ~(size_t)63;
    unsigned char* l_nodes = lds_raw + off;
    off += ((size_t)sc.n_nodes * PT_SWZ_NODE_BYTES + 63) & ~(size_t)63; /* (the packets behind it are read 16 bytes at a time) */
    const int total_slots = sc.n_slots + sc.n_floor;
    double* l_sph = (double*)(lds_raw + off);
    off += (size_t)total_slots * PT_LDS_SPH_DOUBLES * sizeof(double);
    double* l_tri = (double*)(lds_raw + off);
    if (MODE == PT_MODE_ARRAY && sc.has_triangles) off += (size_t)total_slots * 10 * sizeof(double);
    uint8_t* l_kind = (uint8_t*)(lds_raw + off);
    if (MODE == PT_MODE_ARRAY) off += ((size_t)total_slots + 15) & ~(size_t)15;
    uint8_t* l_cat = (uint8_t*)(lds_raw + off);
    off += ((size_t)total_slots + 15) & ~(size_t)15;
    double* l_n64 = (double*)(lds_raw + off);
    /* nodes: the binary32 filter image (PT_SWZ_NODE_BYTES each); links become byte offsets into it -- with PT_SWZ_SIGNSEL,
     * absolute LDS addresses (layout 3 above) */
#if PT_SWZ_SIGNSEL
    const uint32_t nbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)l_nodes;
#else
    const uint32_t nbase = 0u;
#endif
    for (int k = threadIdx.x; k < sc.n_nodes; k += blockDim.x) {
      const PtNode* src = sc.nodes + k;
      uint32_t* w = (uint32_t*)(l_nodes + (size_t)k * PT_SWZ_NODE_BYTES);
      float mag = 0.0f;
      for (int ax = 0; ax < 3; ++ax) {
        const float lo = (float)src->mn[ax], hi = (float)src->mx[ax];
#if PT_SWZ_SIGNSEL
        w[3 * ax] = __float_as_uint(lo);
        w[3 * ax + 1] = __float_as_uint(hi);
        w[3 * ax + 2] = __float_as_uint(lo);
#else
        w[ax] = __float_as_uint(lo);
        w[3 + ax] = __float_as_uint(hi);
#endif
        mag = __builtin_fmaxf(mag, __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi)));
      }
      const uint32_t axis = src->b >> 30;
      const bool leaf = axis == PT_NODE_LEAF_AXIS;
      uint32_t* lk = w + PT_SWZ_OFF_LINKS / 4;
      lk[0] = leaf ? ((src->a & 0xffffu) | ((src->pad[0] & 0xffffu) << 16))
                   : ((nbase + src->a * PT_SWZ_NODE_BYTES) | ((nbase + (src->b & 0x3fffffffu) * PT_SWZ_NODE_BYTES) << 16));
      /* rounded up, then the two lowest mantissa bits carry the axis (a relative change below 2^-21, inside the slack) */
      lk[1] = ((__float_as_uint(mag * 1.000001f) + 4u) & ~3u) | axis;
      uint16_t* sk = (uint16_t*)((unsigned char*)w + PT_SWZ_OFF_SKIP);
      for (int o = 0; o < 8; ++o) {
        const uint32_t nx = sc.node_skip[(size_t)k * 8 + o];
        sk[o] = nx == 0xffffu ? (uint16_t)PT_SWZ_END : (uint16_t)(nbase + nx * PT_SWZ_NODE_BYTES);
#if PT_SWZ_NEAR
        uint16_t* nr = (uint16_t*)((unsigned char*)w + PT_SWZ_OFF_NEAR);
#if PT_SWZ_SIGNSEL && PT_SWZ_TAGGED
        /* tagged links (PT_SWZ_TAG_*): a leaf's entry is what follows it, with "holds a leaf" set */
        nr[o] = leaf ? (uint16_t)(sk[o] | PT_SWZ_TAG_LEAF) : (uint16_t)(nbase + (((o >> axis) & 1) ? src->a : (src->b & 0x3fffffffu)) * PT_SWZ_NODE_BYTES);
#else
        nr[o] = leaf ? (uint16_t)PT_SWZ_LEAF : (uint16_t)(nbase + (((o >> axis) & 1) ? src->a : (src->b & 0x3fffffffu)) * PT_SWZ_NODE_BYTES);
#endif
#endif
      }
    }
    {
      const uint4* src = (const uint4*)sc.sph;
      uint4* dst = (uint4*)l_sph;
      for (int k = threadIdx.x; k < total_slots * 2; k += blockDim.x) dst[(k >> 1) * (PT_LDS_SPH_DOUBLES / 2) + (k & 1)] = src[k];
    }
    if (MODE == PT_MODE_ARRAY) {
      if (sc.has_triangles) {
        const uint4* src = (const uint4*)sc.tri;
        uint4* dst = (uint4*)l_tri;
        for (int k = threadIdx.x; k < total_slots * 5; k += blockDim.x) dst[k] = src[k];
      }
      for (int k = threadIdx.x; k < total_slots; k += blockDim.x) l_kind[k] = sc.slot_kind[k];
    }
    for (int k = threadIdx.x; k < total_slots; k += blockDim.x) l_cat[k] = sc.slot_cat[k];
    if (sc.lds_nodes64) /* (workgroup-uniform) mn.xyz, mx.xyz: the first 48 bytes of a PtNode */
      for (int k = threadIdx.x; k < sc.n_nodes * 3; k += blockDim.x)
        ((double2*)l_n64)[k] = ((const double2*)(sc.nodes + k / 3))[k % 3];
    __syncthreads();
    sv.swz_nodes = l_nodes;
    sv.swz_root = nbase;
    sv.sph = l_sph;
    sv.tri = l_tri;
    sv.kind = l_kind;
    sv.cat = l_cat;
    if (sc.lds_nodes64) sv.nodes64 = l_n64;
  }
  return sv;
}

/* Dynamic hand-out of a launch's work units (64-ray chunks for trace, workgroup windows for shade).  A static
 * stride (unit = wave + k * n_waves) gave every wave ~28 units per launch whose costs differ several-fold (sky
 * against ground rows, 1 against 8 live bounces): the slowest wave ran ~1.7x the mean and a resident wave was alive
 * for only 62 % (trace) / 47 % (shade) of its kernel's duration (rocprofv3 SQ_WAVE_CYCLES against the kernel time,
 * profiles/r02a_sq.json).  Units are dealt from up to 8 counters (workgroups b and b + 8 share an XCD, so a counter's
 * line stays in one L2: MI355X_MICROARCH.md "dequeue": one word saturates at ~88 atomics / us, 8 sharded heads do
 * not), PT_CHUNK_FETCH units per atomic.  The counters must be zero at launch.  Wave-uniform. */
#ifndef PT_CHUNK_FETCH
#define PT_CHUNK_FETCH 2
#endif
#ifndef PT_DYNAMIC_CHUNKS
#define PT_DYNAMIC_CHUNKS 2
#endif
/* PT_DYNAMIC_CHUNKS: 0 = static stride per wave; 1 = global counters (above); 2 = the workgroup keeps its static
 * share (chunks blockIdx, blockIdx + gridDim, ...) and its waves take them from a counter in LDS: no global atomic,
 * and a workgroup's total is the sum of ~450 chunk costs instead of a wave's ~28, so the spread between workgroups is
 * a quarter of the spread between waves. */
struct PtChunkFeed {
  uint32_t* ctr;
  uint32_t nc, c, limit, next, end;
  __device__ __forceinline__ void init(uint32_t* work, uint32_t total_units, uint32_t* lds_ctr) {
#if PT_DYNAMIC_CHUNKS == 0
    /* static stride: unit = wave + k * n_waves */
    nc = gridDim.x * (blockDim.x >> 6);
    c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    limit = c < total_units ? (total_units - c + nc - 1u) / nc : 0u;
    next = 0u;
    end = limit;
    ctr = work;
#elif PT_DYNAMIC_CHUNKS == 2
    nc = gridDim.x;
    c = blockIdx.x;
    limit = c < total_units ? (total_units - c + nc - 1u) / nc : 0u;
    next = end = 0u;
    ctr = lds_ctr; /* zeroed by the caller before a workgroup barrier */
#else
    nc = gridDim.x < 8u ? gridDim.x : 8u;
    c = blockIdx.x % nc;
    ctr = work + c;
    limit = c < total_units ? (total_units - c + nc - 1u) / nc : 0u; /* units c, c + nc, c + 2 nc, ... */
    next = end = 0u;
#endif
  }
  __device__ __forceinline__ bool take(uint32_t& unit) {
#if PT_DYNAMIC_CHUNKS == 0
    if (next >= end) return false;
#else
    if (next >= end) {
      uint32_t k = 0u;
#if PT_DYNAMIC_CHUNKS == 2
      if (pt_lane() == 0) k = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t got = 1u;
#else
      if (pt_lane() == 0) k = atomicAdd(ctr, (uint32_t)PT_CHUNK_FETCH);
      const uint32_t got = PT_CHUNK_FETCH;
#endif
      k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
      if (k >= limit) {
        next = end = limit;
        return false;
      }
      next = k;
      end = (k + got < limit) ? k + got : limit;
    }
#endif
    unit = next * nc + c;
    ++next;
    return true;
  }
};

/* The traverse + intersect stage.  PRIMARY: bounce 0, rays come from the sampler + camera, not from a queue.
 * LDS_SCENE: the whole tree and every leaf packet are first copied into LDS (small scenes: Shirley is
 * 22 KB of nodes + 22 KB of packets), so node / packet reads are ds_read_b128 instead of L1 traffic.
 * LDS layout: [traversal stacks: waves x depth x 64 u32][nodes][sphere slots][triangle slots][slot kinds] */
#ifndef PT_TRACE_GLOBAL_WAVES
#define PT_TRACE_GLOBAL_WAVES 4
#endif
#ifndef PT_TRACE_PRIO
#define PT_TRACE_PRIO 0 /* s_setprio of the trace / pooled shade waves (0..3): matters only where both share a SIMD (two batches) */
#endif
#ifndef PT_SHADE_PRIO
#define PT_SHADE_PRIO 0
#endif
#ifndef PT_TAIL_CUT
#define PT_TAIL_CUT 16 /* 0 = off */
#endif
#ifndef PT_TAIL_CUT_GLOBAL
/* scenes walked from HBM / L2 (k_trace): a step of the walk is a round trip to the L2, so lanes that idle while a chunk's long
 * rays finish cost more there, and a chunk is cut earlier.  Ganesha-like, per-octant node image: trace 23.7 (cut at 16) ->
 * 22.8 ms (24, 32); round 3 measured 24 / 32 within noise on the three-load walk. */
#define PT_TAIL_CUT_GLOBAL 32
#endif
#ifndef PT_TRACE_DIV_LOOP
#define PT_TRACE_DIV_LOOP(LDS_SCENE) (PT_WALK_LOOP != 0 || !(LDS_SCENE))
#endif
#ifndef PT_TRACE_BLOCK_LDS
#define PT_TRACE_BLOCK_LDS 1024 /* workgroup size when the scene is copied to LDS (one copy per workgroup) */
#endif
#ifndef PT_TRACE_BLOCK_LDS_ARRAY
#define PT_TRACE_BLOCK_LDS_ARRAY 512 /* Array_leaf kernels need ~106 VGPRs (4 waves per SIMD either way): cornell +6.6 % over 1024 */
#endif
#define PT_TRACE_BLOCK_OF(MODE, LDS_SCENE) \
  ((LDS_SCENE) ? ((MODE) == PT_MODE_SIMD ? PT_TRACE_BLOCK_LDS : PT_TRACE_BLOCK_LDS_ARRAY) : PT_TRACE_BLOCK_GLOBAL)
#ifndef PT_TRACE_LDS_WAVES
#define PT_TRACE_LDS_WAVES 8 /* Simd_leaf only: waves per SIMD asked of the register allocator */
#endif
#ifndef PT_TRACE_BLOCK_GLOBAL
#define PT_TRACE_BLOCK_GLOBAL 256 /* workgroup size when the scene is traversed from HBM/L2 (512: -6 % on ganesha-like) */
#endif
template <int MODE, bool COUNT, bool PRIMARY, bool LDS_SCENE, bool PACKET>
/* Simd_leaf + LDS scene fits 64 VGPRs without spilling: ask for 2 x 1024-thread workgroups per CU.  The Array_leaf
 * variants (triangle / scalar-sphere code) need ~100 VGPRs: forcing 64 would spill to scratch (1.5 GB of HBM
 * writes per launch on cornell). */
__global__ __launch_bounds__(PT_TRACE_BLOCK_OF(MODE, LDS_SCENE), (LDS_SCENE && MODE == PT_MODE_SIMD) ? PT_TRACE_LDS_WAVES : PT_TRACE_GLOBAL_WAVES) void k_trace(PtSceneDev sc, PtQueue q, PtHits hits, int stack_depth,
                                               PtCounters* counters, PtGenParams g, const double* __restrict__ alpha,
                                               uint32_t n_primary, uint32_t* work, uint4* susp, int top_in_lds) {
  extern __shared__ __attribute__((aligned(64))) unsigned char lds_raw[];
#if PT_TRACE_PRIO
  __builtin_amdgcn_s_setprio(PT_TRACE_PRIO);
#endif
  const int lane = pt_lane();
  const int wave_in_block = (int)(threadIdx.x >> 6);
  /* LDS-resident scenes have < 65536 nodes: 16-bit stack entries halve the stack footprint */
  /* no per-lane stack anywhere: both walks are threaded.  PACKET on a scene walked from HBM / L2 (where wave packets do not
   * pay) selects the per-octant node image instead (PtThreadOctTag; the host launches it when PtSceneDev.nodes32o exists) */
  typedef typename std::conditional<LDS_SCENE, uint16_t, typename std::conditional<PACKET, PtThreadOctTag, PtThreadTag>::type>::type StackT;
  StackT* stack = (StackT*)(lds_raw + (size_t)wave_in_block * PT_WAVE_STACK_BYTES(LDS_SCENE, StackT, stack_depth));
  __shared__ uint32_t lds_chunk_ctr;
  if (threadIdx.x == 0) lds_chunk_ctr = 0u;
  PtSceneView sv = pt_scene_view<MODE, LDS_SCENE, StackT>(sc, lds_raw, stack_depth, top_in_lds != 0);
  if (!LDS_SCENE) {
    __shared__ double lds_floor[PT_FLOOR_LDS * 10];
    if (MODE == PT_MODE_ARRAY && sc.n_floor > 0) { /* the pre-tested floor triangles: see PtSceneView.floor_lds */
      sv.n_floor_lds = sc.n_floor < PT_FLOOR_LDS ? sc.n_floor : PT_FLOOR_LDS;
      sv.floor_lds = (const __attribute__((address_space(3))) double*)lds_floor;
      for (int k = threadIdx.x; k < sv.n_floor_lds * 10; k += blockDim.x) lds_floor[k] = sc.tri[(size_t)sc.n_slots * 10 + k];
    }
    __syncthreads(); /* pt_scene_view ends with a barrier only when it copies the whole scene */
  }
  const uint32_t n = PRIMARY ? n_primary : *q.count;
  PtChunkFeed feed;
  feed.init(work, (uint32_t)(((unsigned long long)n + PT_WAVE - 1) / PT_WAVE), &lds_chunk_ctr);
  uint32_t chunk;
  unsigned long long c_nodes = 0, c_prims = 0, c_floor = 0, c_seg = 0, c_filter[2] = {0, 0};

  if (PACKET && LDS_SCENE) { /* the 64 rays of the wave walk the tree together (pt_trace_packet) */
    /* the wave's private stack area (stack_depth x 64 entries) holds the shared (node, mask) stack: 12 B per level */
    uint32_t* wstack = (uint32_t*)(lds_raw + (size_t)wave_in_block * PT_WAVE_STACK_BYTES(LDS_SCENE, StackT, stack_depth));
    while (feed.take(chunk)) {
      const uint32_t i = chunk * PT_WAVE + lane;
      bool valid = i < n;
      V3 o = v3(0.0, 0.0, 0.0), d = v3(0.0, 0.0, -1.0);
      if (valid) {
        if (PRIMARY) {
          const PtPrimarySample ps = pt_primary_decode(g, i);
          valid = ps.valid;
          if (valid) d = pt_primary_dir(sc, g, ps, alpha);
        } else {
          pt_q_load_ray(q, i, o, d);
          if (pt_is_hole(d.x)) {
            valid = false;
            hits.slot[i] = PT_SLOT_HOLE;
            o = v3(0.0, 0.0, 0.0);
            d = v3(0.0, 0.0, -1.0);
          }
        }
      }
      if (COUNT && valid) c_seg++;
      const PtTraceResult r = pt_trace_packet<MODE, COUNT, PRIMARY, LDS_SCENE>(sc, sv, wstack, valid, o, d, c_nodes, c_prims, c_floor, c_filter);
      if (valid) {
        pt_hit_store(hits, i, r.t, r.slot, r.u, r.v, MODE == PT_MODE_ARRAY && sc.has_triangles);
      }
    }
  } else {
    /* One ray per lane.  TAIL: see PtTailCtl -- a chunk ends when fewer than PT_TAIL_CUT of its rays are still walking;
     * their states (16 bytes each) go to this wave's list in `susp` and once 64 - PT_TAIL_CUT have gathered the wave walks
     * them as a chunk of their own.  tools/sim_coherence.py: 0.527 -> 0.435 wave steps per ray at 16. */
    constexpr int CUT = LDS_SCENE ? PT_TAIL_CUT : PT_TAIL_CUT_GLOBAL;
    constexpr bool TAIL = CUT > 0 && PT_DIAG == 0; /* both walks are threaded (LDS image, HBM / L2); a parked camera ray is recomputed from its index */
    constexpr bool TAIL_UV = TAIL && MODE == PT_MODE_ARRAY; /* triangle hits carry barycentrics: a second 16 bytes per state */
    constexpr bool TAIL_W = TAIL && !LDS_SCENE;            /* 32-bit node index and slot: a third 16 bytes */
    uint4* my_susp = TAIL ? susp + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block) * (PT_WAVE * 3) : nullptr;
    uint32_t n_susp = 0; /* wave-uniform */
    bool more = true;
    for (;;) { /* no `continue` past a point where lanes have diverged: take() is wave-uniform */
      bool resume = false, valid = false;
      uint32_t i = 0;
      uint4 parked = make_uint4(0, 0, 0, 0), parked_uv = make_uint4(0, 0, 0, 0), parked_w = make_uint4(0, 0, 0, 0);
      if (TAIL && (n_susp > (uint32_t)(PT_WAVE - CUT) || (!more && n_susp > 0))) {
        resume = true;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); /* this wave's own parked states, written below */
        valid = (uint32_t)lane < n_susp;
        if (valid) parked = my_susp[lane];
        if (TAIL_UV && valid) parked_uv = my_susp[PT_WAVE + lane];
        if (TAIL_W && valid) parked_w = my_susp[2 * PT_WAVE + lane];
        i = parked.x;
        n_susp = 0;
      } else {
        if (!more) break;
        more = feed.take(chunk);
        if (more) {
          i = chunk * PT_WAVE + lane;
          valid = i < n;
        }
      }
      V3 o = v3(0.0, 0.0, 0.0), d = v3(0.0, 0.0, -1.0); /* P3.origin */
      if (valid) {
        if (PRIMARY) {
          const PtPrimarySample ps = pt_primary_decode(g, i);
          valid = ps.valid;
          if (valid) d = pt_primary_dir(sc, g, ps, alpha);
        } else {
          pt_q_load_ray(q, i, o, d);
          if (pt_is_hole(d.x)) { /* never parked, so never seen on resume */
            valid = false;
            hits.slot[i] = PT_SLOT_HOLE;
            o = v3(0.0, 0.0, 0.0);
            d = v3(0.0, 0.0, -1.0);
          }
        }
      }
      if (COUNT && valid && !resume) c_seg++;
      const unsigned long long diag_n0 = c_nodes;
      PtTailCtl tc;
      tc.min_active = (TAIL && more) ? CUT : 0; /* the last chunks of a wave run to completion */
      tc.resume = resume && valid;
      tc.node = TAIL_W ? parked_w.x : (parked.y & 0xffffu);
      tc.slot = TAIL_W ? (int)parked_w.y : ((int)(parked.y >> 16) == 0xffff ? -1 : (int)(parked.y >> 16));
      tc.t = __hiloint2double((int)parked.w, (int)parked.z);
      tc.u = __hiloint2double((int)parked_uv.y, (int)parked_uv.x);
      tc.v = __hiloint2double((int)parked_uv.w, (int)parked_uv.z);
      tc.unfinished = false;
      /* every lane goes in (wave-level ballots inside); lanes without a ray commit nothing */
      /* (the node loop as one divergent loop where the kernel runs at 4 waves per SIMD: the walk from HBM / L2 -- ganesha-like
       * frame 35.4 -> 33.6 ms; the LDS walk at 8 waves per SIMD loses 3 % with it) */
      const PtTraceResult r = pt_trace_ray<MODE, COUNT, PRIMARY, StackT, LDS_SCENE, PT_TRACE_DIV_LOOP(LDS_SCENE)>(sc, sv, stack, o, d, c_nodes, c_prims, c_floor, valid, TAIL ? &tc : nullptr, c_filter);
      if (COUNT && PT_DIAG == 2 && !PRIMARY && valid) {
        unsigned long long m = c_nodes - diag_n0;
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned long long other = __shfl_xor(m, off);
          m = other > m ? other : m;
        }
        PT_DIAG_WAVE_SLOTS(c_floor);
        if (lane == 0) c_floor += m * 64 - 64;
      }
      const bool park = TAIL && tc.unfinished;
      if (valid && !park) {
        pt_hit_store(hits, i, r.t, r.slot, r.u, r.v, MODE == PT_MODE_ARRAY && sc.has_triangles);
      }
      if (TAIL) {
        const unsigned long long pm = __ballot(park);
        if (pm != 0) {
          if (park) {
            const uint32_t k = n_susp + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
            my_susp[k] = make_uint4(i, tc.node | ((uint32_t)(r.slot < 0 ? 0xffff : r.slot) << 16),
                                    (uint32_t)__double2loint(r.t), (uint32_t)__double2hiint(r.t));
            if (TAIL_W) my_susp[2 * PT_WAVE + k] = make_uint4(tc.node, (uint32_t)r.slot, 0u, 0u);
            if (TAIL_UV)
              my_susp[PT_WAVE + k] = make_uint4((uint32_t)__double2loint(r.u), (uint32_t)__double2hiint(r.u), (uint32_t)__double2loint(r.v), (uint32_t)__double2hiint(r.v));
          }
          n_susp += (uint32_t)__popcll(pm);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
      }
    }
  }
  if (COUNT) {
    c_nodes = pt_wave_sum(c_nodes);
    c_prims = pt_wave_sum(c_prims);
    c_floor = pt_wave_sum(c_floor);
    c_seg = pt_wave_sum(c_seg);
    c_filter[0] = pt_wave_sum(c_filter[0]);
    c_filter[1] = pt_wave_sum(c_filter[1]);
    if (lane == 0) {
      atomicAdd(&counters->nodes, c_nodes);
      atomicAdd(&counters->prims, c_prims);
      atomicAdd(&counters->floor, c_floor);
      atomicAdd(&counters->segments, c_seg);
      atomicAdd(&counters->undecided, c_filter[0]);
      atomicAdd(&counters->fallback_steps, c_filter[1]);
    }
  }
}

/* ------------------------------------------------------------------ shade */
struct Quat {
  double r;
  V3 v;
};
/* Quaternion.mul (quaternion.ml:25-32) */
__device__ __forceinline__ Quat pt_quat_mul(Quat a, Quat b) {
  Quat o;
  o.r = (a.r * b.r) - v3_dot(a.v, b.v);
  o.v = v3_add(v3_add(v3_cross(a.v, b.v), v3_scale(b.v, a.r)), v3_scale(a.v, b.r));
  return o;
}
/* Quaternion.transform t v = ((t * (0, v)) * conj t).v (quaternion.ml:34-42) */
__device__ __forceinline__ V3 pt_quat_transform(Quat t, V3 v) {
  Quat p;
  p.r = 0.0;
  p.v = v;
  Quat c;
  c.r = t.r;
  c.v = v3_neg(t.v);
  return pt_quat_mul(pt_quat_mul(t, p), c).v;
}
/* Shader_space.create (shader_space.ml:11-23) + Quaternion.normalize (quaternion.ml:11-15) */
__host__ __device__ __forceinline__ Quat pt_shader_rotation(V3 normal) {
  const double epsilon = 1e-9;
  Quat q;
  if (normal.z > 1.0 - epsilon) {
    q.r = 1.0;
    q.v = v3(0.0, 0.0, 0.0);
  } else if (normal.z < epsilon - 1.0) {
    q.r = 0.0;
    q.v = v3(0.0, 1.0, 0.0);
  } else {
    const double r = 1.0 + normal.z;
    const V3 v = v3(normal.y, -normal.x, 0.0);
    const double s = pt_rnorm_frame(r, v.x, v.y); /* v.z = 0 */
    q.r = r * s;
    q.v = v3_scale(v, s);
  }
  return q;
}
/* Triangle.Hit.to_hit's geometric normal (triangle.ml:43-64) */
#define PT_TRI_FRAME_DOUBLES 12
__host__ __device__ __forceinline__ V3 pt_tri_normal(V3 a, V3 b, V3 c) { return v3_normalize(v3_cross(v3_sub(b, a), v3_sub(c, a))); }
__device__ __forceinline__ Quat pt_quat_conj(Quat q) {
  Quat c;
  c.r = q.r;
  c.v = v3_neg(q.v);
  return c;
}

/* What a segment needs of its slot's shading record (PtShadeRec), in registers: every address is known the moment the segment
 * starts (queue index and hit slot), so everything is requested in one round (pt_shade_entry) instead of field by field where
 * the code needs it.  Which fields are read is fixed by the category when the kernel is specialised for one. */
struct PtMatRegs {
  int kind, tex_kind, tex_w, tex_h;
  double index;
  V3 even, odd, emit;
};
template <int CAT, bool EMIT>
__device__ __forceinline__ PtMatRegs pt_mat_load(const PtShadeRec* m) {
  PtMatRegs r;
  r.kind = (CAT == PT_CAT_LAMBERT_SOLID || CAT == PT_CAT_LAMBERT_CHECKER) ? 0 : (CAT == PT_CAT_METAL ? 1 : (CAT == PT_CAT_DIELECTRIC ? 2 : m->kind));
  r.tex_kind = CAT == PT_CAT_LAMBERT_SOLID ? 0 : (CAT == PT_CAT_LAMBERT_CHECKER ? 1 : (CAT == PT_CAT_DIELECTRIC ? 0 : m->tex_kind));
  const bool tex = CAT != PT_CAT_DIELECTRIC;
  const bool checker = CAT != PT_CAT_DIELECTRIC && CAT != PT_CAT_LAMBERT_SOLID;
  r.tex_w = checker ? m->tex_w : 1;
  r.tex_h = checker ? m->tex_h : 1;
  r.index = (CAT == PT_CAT_DIELECTRIC || CAT == PT_CAT_NONE) ? m->index : 1.0;
  r.even = tex ? v3(m->even[0], m->even[1], m->even[2]) : v3(0.0, 0.0, 0.0);
  r.odd = checker ? v3(m->odd[0], m->odd[1], m->odd[2]) : v3(0.0, 0.0, 0.0);
  r.emit = EMIT ? v3(m->emit[0], m->emit[1], m->emit[2]) : v3(0.0, 0.0, 0.0);
  return r;
}

/* Texture.eval (texture.ml:16-31) */
__device__ __forceinline__ V3 pt_texture_eval(const PtMatRegs& t, double u, double v) {
  if (t.tex_kind == 0) return t.even;
  const double width = (double)(t.tex_w - 1), height = (double)(t.tex_h - 1);
  const double xp = u * width, yp = v * height;
  const long long px = ((long long)xp) & 1, py = ((long long)yp) & 1; /* Float.to_int a land 1 */
  if (px == py) return t.even;
  return t.odd;
}

/* schlick_reflectance (material.ml:16-20) */
__device__ __forceinline__ double pt_schlick(double cos_theta, double index) {
  const double qd = (1.0 - index) / (1.0 + index);
  const double r0 = qd * qd;
  return r0 + ((1.0 - r0) * pt_pow5(1.0 - cos_theta));
}

/* Scene.background (shirley_spheres/bin/main.ml:104-110) */
__device__ __forceinline__ V3 pt_background(const PtSceneDev& sc, V3 dir) {
  if (sc.bg_kind == 0) return v3(0.0, 0.0, 0.0);
  const V3 d = v3_normalize(dir);
  const double t = 0.5 * (v3_dot(d, v3(0.0, 1.0, 0.0)) + 1.0);
  return v3_lerp(t, v3(sc.bg_horizon[0], sc.bg_horizon[1], sc.bg_horizon[2]),
                 v3(sc.bg_zenith[0], sc.bg_zenith[1], sc.bg_zenith[2]));
}

/* What Sphere.hit (sphere.ml:56-69) / Triangle.Hit.to_hit (triangle.ml:43-64) build: the local geometry of a hit
 * and what Material.scatter's partial application captures. */
struct PtSurface {
  V3 point, normal; /* Shader_space.world_origin / world_normal */
  Quat rot;         /* Shader_space.rotation */
  V3 omega_i;       /* Shader_space.omega_i */
  double tu, tv;    /* Texture.Coord */
  bool hit_front;
};
/* geometry of the hit slot requested together with the rest of the segment's inputs (pt_shade_entry): the sphere's record on
 * scenes without triangles; with triangles the slot's kind, which decides what to fetch in a second round */
struct PtSlotGeom {
  double cx, cy, cz;
  int kind; /* PT_SLOT_* */
};
/* (the photon passes' form: one ray at a time, nothing to batch the loads with) */
__device__ __forceinline__ PtSlotGeom pt_slot_geom(const PtSceneDev& sc, int slot) {
  PtSlotGeom g;
  g.cx = g.cy = g.cz = 0.0;
  g.kind = sc.has_triangles ? (int)sc.slot_kind[slot] : PT_SLOT_SPHERE;
  if (!sc.has_triangles) {
    const double* s = sc.sph + (size_t)slot * 4;
    g.cx = s[0];
    g.cy = s[1];
    g.cz = s[2];
  }
  return g;
}
/* CAT: the shading category when the caller knows it at compile time (per-category shade kernels), PT_CAT_NONE when it
 * has to be read from the slot's record */
template <int CAT = PT_CAT_NONE>
__device__ __forceinline__ PtSurface pt_surface_hit(const PtSceneDev& sc, V3 o, V3 d, int slot, double t_hit, double bu,
                                                    double bv, const PtMatRegs& m, const PtSlotGeom& geom) {
  const double pi = 3.14159265358979323846;
  PtSurface sf;
  sf.tu = 0.0;
  sf.tv = 0.0;
  /* tex_coord feeds Texture.eval only, and only a checker reads it */
  const bool need_uv = CAT == PT_CAT_LAMBERT_SOLID || CAT == PT_CAT_DIELECTRIC ? false
                       : (CAT == PT_CAT_LAMBERT_CHECKER ? true : ((m.kind != 2) && (m.tex_kind != 0)));
  if (geom.kind == PT_SLOT_SPHERE) {
    /* Sphere.hit (sphere.ml:56-69) */
    V3 center = v3(geom.cx, geom.cy, geom.cz);
    if (sc.has_triangles) { /* (mixed scenes learn the slot's kind in the first round and fetch its geometry in a second) */
      const double* s = sc.sph + (size_t)slot * 4;
      center = v3(s[0], s[1], s[2]);
    }
    sf.point = v3_add(o, v3_scale(d, t_hit)); /* Ray.point_at, ray.ml:15 */
    V3 normal = v3_normalize(v3_sub(sf.point, center));
    sf.hit_front = v3_dot(d, normal) < 0.0;
    if (!sf.hit_front) normal = v3_neg(normal);
    sf.normal = normal;
    if (need_uv) { /* tex_coord (sphere.ml:25-33) feeds Texture.eval only */
      const double one_over_pi = 1.0 / pi, one_over_two_pi = 1.0 / (2.0 * pi);
      const double theta = pt_acos(-normal.y);
      const double phi = pi + pt_atan2(-normal.z, normal.x);
      sf.tu = phi * one_over_two_pi;
      sf.tv = theta * one_over_pi;
    }
  } else {
    /* Triangle.Hit.to_hit (triangle.ml:43-64) */
    const double* tvx = sc.tri + (size_t)slot * 10;
    const V3 a = pt_load_v3(tvx), b = pt_load_v3(tvx + 3), c = pt_load_v3(tvx + 6);
    const bool framed = sc.tri_frame != nullptr; /* (wave-uniform) */
    V3 g_normal;
    if (framed) g_normal = pt_load_v3(sc.tri_frame + (size_t)slot * PT_TRI_FRAME_DOUBLES);
    else g_normal = pt_tri_normal(a, b, c);
    const double u = bu, v = bv;
    const double w = 1.0 - u - v;
    sf.point = v3_add(v3_add(v3_scale(a, w), v3_scale(b, u)), v3_scale(c, v));
    const double* uv = sc.tri_uv + (size_t)slot * 6;
    sf.tu = (uv[0] * w) + (uv[2] * u) + (uv[4] * v);
    sf.tv = (uv[1] * w) + (uv[3] * u) + (uv[5] * v);
    sf.hit_front = v3_dot(d, g_normal) < 0.0;
    sf.normal = sf.hit_front ? g_normal : v3_neg(g_normal);
    if (framed) { /* Shader_space.create of that normal: one of the slot's two precomputed rotations (PtSceneDev.tri_frame); both are
                     requested with the normal -- a load that depends on hit_front would be one more round trip in the step's chain */
      const double2* r = (const double2*)(sc.tri_frame + (size_t)slot * PT_TRI_FRAME_DOUBLES + 4);
      const double2 f0 = r[0], f1 = r[1], b0 = r[2], b1 = r[3];
      sf.rot.r = sf.hit_front ? f0.x : b0.x;
      sf.rot.v = v3(sf.hit_front ? f0.y : b0.y, sf.hit_front ? f1.x : b1.x, sf.hit_front ? f1.y : b1.y);
      sf.omega_i = pt_quat_transform(sf.rot, v3_neg(d));
      return sf;
    }
  }
  sf.rot = pt_shader_rotation(sf.normal);
  sf.omega_i = pt_quat_transform(sf.rot, v3_neg(d)); /* Shader_space.omega_i */
  return sf;
}

/* Material.scatter applied to u (material.ml:22-57): kind 0 Absorb | 1 Specular (wo = shader-space direction of the
 * scattered ray, attenuation) | 2 Diffuse (attenuation = the texture colour) */
struct PtScatter {
  int kind;
  V3 attenuation, wo;
};
template <int CAT = PT_CAT_NONE>
__device__ __forceinline__ PtScatter pt_material_scatter(const PtSceneDev& sc, const PtSurface& sf, const PtMatRegs& m, double su) {
  const V3 omega_i = sf.omega_i;
  PtScatter r;
  r.attenuation = v3(1.0, 1.0, 1.0);
  r.wo = v3(0.0, 0.0, 0.0);
  const int mkind = (CAT == PT_CAT_LAMBERT_SOLID || CAT == PT_CAT_LAMBERT_CHECKER) ? 0
                    : (CAT == PT_CAT_METAL ? 1 : (CAT == PT_CAT_DIELECTRIC ? 2 : m.kind));
  if (mkind == 0) {
    r.kind = 2;
    r.attenuation = pt_texture_eval(m, sf.tu, sf.tv);
  } else if (mkind == 1) {
    const V3 omega_r = v3(-omega_i.x, -omega_i.y, omega_i.z); /* Shader_space.reflect */
    if (omega_r.z <= 0.0) {
      r.kind = 0;
    } else {
      r.kind = 1;
      const V3 a = pt_texture_eval(m, sf.tu, sf.tv);
      const double sp5 = pt_pow5(1.0 - omega_i.z);
      const V3 c = v3_scale(v3_sub(v3(1.0, 1.0, 1.0), a), sp5);
      r.attenuation = v3_add(a, c);
      r.wo = omega_r;
    }
  } else {
    r.kind = 1;
    const double index = m.index, index_inv = 1.0 / m.index;
    const double wi_z = omega_i.z;
    const double c = wi_z < 0.0 ? 0.0 : (1.0 < wi_z ? 1.0 : wi_z); /* Float.clamp_exn */
    const double sn = pt_sqrt_nonneg(1.0 - c * c);
    const double refract_ratio = sf.hit_front ? index_inv : index;
    /* Both candidate directions are cheap: evaluate them unconditionally and select.  (A divergent
     * `a || f(x) > u` branch here was miscompiled by hipcc -O3 -- caught by the bit-exact sample test.) */
    const bool reflect = (refract_ratio * sn > 1.0) || (pt_schlick(c, refract_ratio) > su);
    /* Shader_space.refract (shader_space.ml:41-49) */
    const double cc = pt_base_min(omega_i.z, 1.0);
    const V3 perp = v3_scale(v3_sub(v3(0.0, 0.0, cc), omega_i), refract_ratio);
    const V3 para = v3(0.0, 0.0, -pt_sqrt_nonneg(pt_fabs(1.0 - v3_quadrance(perp))));
    const V3 refr = v3_add(perp, para);
    r.wo.x = reflect ? -omega_i.x : refr.x; /* Shader_space.reflect (shader_space.ml:34-39) */
    r.wo.y = reflect ? -omega_i.y : refr.y;
    r.wo.z = reflect ? omega_i.z : refr.z;
  }
  return r;
}

/* per-path final colours, one 32-byte record {r, g, b, -} per contribution id: a path ends in an arbitrary lane of an
 * arbitrary wave, so its three doubles go out as ONE 32-byte sector instead of three 8-byte writes to three arrays
 * (each its own sector: 96 B of HBM traffic for 24 B of data) */
struct PtContrib {
  double4* rgbx;
};

/* The bin a survivor is appended under.  Scenes whose primitives lie in a slab (Shirley: spheres on a ground plane;
 * PtSceneDev.sort_by_elevation, decided at scene creation): the ELEVATION of the new direction above that plane in 8 steps --
 * a predictor of the LENGTH of the next walk: rays that climb leave the slab after a few node tests, grazing rays cross the
 * whole scene, and what a wave of one-ray-per-lane walks loses is the spread of its rays' lengths.  tools/sim_coherence.py on
 * bounce-1 rays, 512-entry windows: wave steps per ray 0.656 (octant) -> 0.584 (elevation); sorting by the true length would give
 * 0.485.  Measured: frame 36.96 -> 35.87 ms.  Other scenes (cornell's closed box, a mesh): the direction OCTANT (rays of a
 * wave share the child order), which measured 1.5-2 % better there. */
template <int BINS = 8>
__device__ __forceinline__ int pt_bin_key(const PtSceneDev& sc, V3 o, V3 d) {
  if (sc.sort_by_elevation) {
    const float e = (float)d.x * (float)sc.sort_axis[0] + (float)d.y * (float)sc.sort_axis[1] + (float)d.z * (float)sc.sort_axis[2];
    const int b = (int)((e + 1.0f) * (0.5f * (float)BINS));
    return b < 0 ? 0 : (b > BINS - 1 ? BINS - 1 : b);
  }
  if (sc.sort_by_root) {
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    const float ix = 1.0f / (float)d.x, iy = 1.0f / (float)d.y, iz = 1.0f / (float)d.z;
    const float ax = (sc.root_mn[0] - ox) * ix, bx = (sc.root_mx[0] - ox) * ix;
    const float ay = (sc.root_mn[1] - oy) * iy, by = (sc.root_mx[1] - oy) * iy;
    const float az = (sc.root_mn[2] - oz) * iz, bz = (sc.root_mx[2] - oz) * iz;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const bool reaches = !(t_out < t_in); /* a NaN (0 * inf) counts as reaching */
    return (reaches ? 4 : 0) | (d.x >= 0.0 ? 1 : 0) | (d.z >= 0.0 ? 2 : 0);
  }
  return (d.x >= 0.0 ? 1 : 0) | (d.y >= 0.0 ? 2 : 0) | (d.z >= 0.0 ? 4 : 0);
}
/* What one segment leaves behind: either the path's final colour (written to `contrib` inside) or the next ray. */
struct PtShadeOut {
  bool keep;
  V3 n_o, n_d, n_attn, n_emit;
  uint32_t id;
  int offset;
};
/* The body of `loop` in Integrator.path_tracer (integrator.ml:30-66) for entry i of the queue (PRIMARY: virtual
 * entry i = a camera sample).  CAT = the entry's shading category if the kernel is specialised for one. */
template <bool EMIT, bool PRIMARY, int CAT>
__device__ __forceinline__ void pt_shade_entry(const PtSceneDev& sc, const PtQueue& q, const PtHits& hits, const PtContrib& contrib,
                                               const double* __restrict__ alpha, int bounce, int last_bounce,
                                               const PtGenParams& g, uint32_t i, bool live, PtShadeOut& so,
                                               int known_slot = PT_SLOT_HOLE /* the entry's hit slot if the caller holds it */) {
    const double pi = 3.14159265358979323846;
    bool keep = false;
    V3 n_o = v3(0, 0, 0), n_d = v3(0, 0, 0), n_attn = v3(0, 0, 0), n_emit = v3(0, 0, 0);
    uint32_t id = 0;
    int offset = 0;
    PtPrimarySample ps;
    if (PRIMARY && live) {
      ps = pt_primary_decode(g, i);
      live = ps.valid;
    }
    if (live) {
      V3 o, d, attn0, emit0 = v3(0.0, 0.0, 0.0);
      if (PRIMARY) {
        o = v3(0.0, 0.0, 0.0);
        d = pt_primary_dir(sc, g, ps, alpha); /* (handed over by the walk as a 32-byte {t, direction} record instead: headline +-0, cornell +2 %, mesh +3 %: profiles/r05_ab_primary_dir.txt) */
        attn0 = v3(1.0, 1.0, 1.0); /* Color.white, integrator.ml:68 */
        id = ps.id;
        offset = ps.offset;
      } else {
        pt_q_load_ray(q, i, o, d);
        pt_q_load_path<EMIT>(q, i, attn0, id, offset, emit0);
      }
      const int slot = CAT == PT_CAT_MISS ? -1 : (known_slot != PT_SLOT_HOLE ? known_slot : hits.slot[i]);
      V3 result = v3(0, 0, 0);
      bool done = true;
      if (CAT == PT_CAT_MISS || (CAT == PT_CAT_NONE && slot < 0)) {
        /* None -> add_mul emit0 attn0 (background ray), integrator.ml:36 */
        result = v3_fma(attn0, pt_background(sc, d), emit0);
      } else {
        /* the rest of the segment's inputs, requested in the same round as the queue records above (PtMatRegs): the hit
         * distance, the slot's shading record, and its geometry (or, with triangles in the scene, its kind) */
        double t_hit, bu = 0.0, bv = 0.0;
        PtSlotGeom geom;
        geom.cx = geom.cy = geom.cz = 0.0;
        geom.kind = PT_SLOT_SPHERE;
        if (sc.has_triangles) {
          geom.kind = (int)sc.slot_kind[slot];
#if PT_RECOMPUTE_HIT
          /* (t, u, v) of the hit again, from the ray and the primitive the walk settled on: Array_leaf's own element tests
           * (PtTraverser::packet, begin's floor test) on the same operands.  Their acceptance range only ever decided WHICH
           * primitive won; the values do not depend on it. */
          t_hit = 0.0;
          if (geom.kind == PT_SLOT_SPHERE) {
            const double* sp = sc.sph + (size_t)slot * 4;
            (void)pt_sphere_intersect_scalar(v3(sp[0], sp[1], sp[2]), sp[3], o, d, 0.0, PT_MAX_FINITE, &t_hit);
          } else {
            const double* tvx = sc.tri + (size_t)slot * 10;
            (void)pt_triangle_intersect(pt_load_v3(tvx), pt_load_v3(tvx + 3), pt_load_v3(tvx + 6), o, d, 0.0, PT_MAX_FINITE, &t_hit, &bu, &bv);
          }
#else
          const double2* hp = (const double2*)(hits.tuv + i); /* one 32-byte record (PtHits) */
          const double2 h0 = hp[0], h1 = hp[1];
          t_hit = h0.x;
          bu = h0.y;
          bv = h1.x;
#endif
        } else {
          t_hit = hits.t[i];
          const double2* sp = (const double2*)(sc.sph + (size_t)slot * 4);
          const double2 s0 = sp[0], s1 = sp[1];
          geom.cx = s0.x;
          geom.cy = s0.y;
          geom.cz = s1.x;
        }
        const PtMatRegs m = pt_mat_load<CAT, EMIT>(sc.slot_shade + slot);
        const bool is_tri = geom.kind != PT_SLOT_SPHERE;
        const PtSurface sf = pt_surface_hit<CAT>(sc, o, d, slot, t_hit, is_tri ? bu : 0.0, is_tri ? bv : 0.0, m, geom);
        const V3 point = sf.point;
        const Quat rot_inv = pt_quat_conj(sf.rot);
        const V3 emit = m.emit;
        /* take_2d (), integrator.ml:20-28,39: dims 2+2k, 3+2k for the k-th hit */
        const double su = pt_lds_get(alpha, offset, 2 + 2 * bounce);
        const double sv = pt_lds_get(alpha, offset, 3 + 2 * bounce);
        const PtScatter scat = pt_material_scatter<CAT>(sc, sf, m, su);
        const int sc_kind = scat.kind;
        V3 attenuation = scat.attenuation;
        V3 wo = scat.wo;

        if (sc_kind == 0) {
          result = v3_fma(attn0, emit, emit0); /* Absorb, integrator.ml:41 */
        } else {
          bool scatter_ok = true;
          if (sc_kind == 2) {
            /* Pdf.sample / Pdf.eval (pdf.ml:5-15, shader_space.ml:56-64) */
            const double r = pt_sqrt_nonneg(su);
            const double theta = sv * 2.0 * pi;
            double sn, cs;
            pt_sincos(theta, &sn, &cs);
            wo = v3(r * cs, r * sn, pt_sqrt_nonneg(1.0 - su));
            const double diffuse_pd = (wo.z < 0.0) ? 0.0 : wo.z / pi;
            if (diffuse_pd == 0.0) {
              scatter_ok = false;
            } else {
              const double pd = diffuse_pd / diffuse_pd; /* divisor = Pdf.eval diffuse_plus_light = the same */
              if (!pt_isfinite(pd)) scatter_ok = false;
              else attenuation = v3_scale(attenuation, pd);
            }
          }
          if (!scatter_ok) {
            result = v3_fma(attn0, emit, emit0); /* integrator.ml:53,58 */
          } else {
            /* Shader_space.world_ray (shader_space.ml:51-54) */
            const V3 dir = pt_quat_transform(rot_inv, wo);
            n_o = v3_add(point, v3_scale(dir, 1e-3));
            n_d = dir;
            n_emit = v3_fma(attenuation, emit0, emit); /* add_mul emit attenuation emit0 */
            n_attn = v3_mul(attenuation, attn0);
            if (last_bounce) {
              /* the recursive call sees max_bounces <= 0: add_mul emit0 attn0 Color.black (integrator.ml:31-32) */
              result = v3_fma(n_attn, v3(0.0, 0.0, 0.0), n_emit);
            } else {
              done = false;
            }
          }
        }
      }
      if (done) {
        contrib.rgbx[id] = make_double4(result.x, result.y, result.z, 0.0);
      } else {
        keep = true;
      }
    }
    so.keep = keep;
    so.n_o = n_o; so.n_d = n_d; so.n_attn = n_attn; so.n_emit = n_emit;
    so.id = id;
    so.offset = offset;
}

#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4
#endif
#ifndef PT_SHADE_TIMING
#define PT_SHADE_TIMING 0
#endif

/* ------------------------------------------------------------------ the shade stage without workgroup barriers
 * Rounds 1-2 shaded category-sorted 512-entry windows (workgroup barriers around a sort and an append): between two barriers the
 * window's waves ran materials of very different length (a miss is ~150 instructions, a checker Lambertian ~1200), and the
 * kernel's own clock showed 33 % of a wave's life at the append's first barrier waiting for the window's slowest wave, 10 %
 * behind the append's atomic, 12 % in the sort, 39 % in the segment's arithmetic (removed in round 5).
 *
 * Here a wave never waits for another one (except while a full output block is being replaced, below):
 *  - POOLS.  Every wave owns one list of queue indices per category in LDS (128 entries each).  It classifies raw chunks of 64
 *    entries (hit slot -> category) into its lists and, whenever a list holds 64, shades those 64 together: one category per
 *    wave step by construction, no sort.  When the input is exhausted the lists are drained (the only partly filled steps).
 *  - BLOCKED OUTPUT.  The workgroup owns one output block of PT_POOL_BLOCK entries per bin (pt_bin_key: elevation / octant);
 *    a wave reserves room for its survivors of a bin with ONE LDS atomic on a word that holds (block, cursor) together, so the
 *    block it writes to is the one its reservation belongs to.  The wave whose reservation crosses the end of the block fills
 *    it up, gets the next block from the queue's global counter (one device atomic per PT_POOL_BLOCK survivors, as few as the
 *    windowed append) and publishes it; waves that arrive in between spin on the LDS word.  A block is single-bin, so every
 *    wave of the next trace launch walks rays of one bin.
 *  - HOLES.  When the kernel ends the blocks in use are partly filled.  The last wave of the workgroup marks the unused
 *    entries by a direction whose x is a NaN with a payload no arithmetic produces (PT_HOLE_HI); k_trace skips such an entry and
 *    records slot PT_SLOT_HOLE for it, which the next launch of this kernel drops when it classifies.  The queue's count
 *    is then a number of ENTRIES (a multiple of the block), not of rays; the segment counters count rays.
 * Every ray still gets exactly pt_shade_entry's arithmetic; the order of the queue changes, which no result depends on
 * (contributions are stored per path id and summed in pass order). */
#define PT_N_SHADE_CAT 5 /* PT_CAT_MISS .. PT_CAT_DIELECTRIC */
#ifndef PT_POOL_BLOCK
#define PT_POOL_BLOCK 256 /* < 4096 - 16 * 64: the cursor field must hold a full block plus one stray reservation per wave */
#endif
#ifndef PT_POOL_THREADS
#define PT_POOL_THREADS 512 /* largest workgroup of k_shade_pool (the pools live in dynamic LDS, sized by blockDim).  Nothing in it is
                               workgroup-wide but the output blocks: when two batches share every CU it runs 256-thread workgroups,
                               which interleave with the other batch's trace workgroups at a finer grain (headline frame 27.6 ms
                               against 29.6 ms with 512, 29.0 ms with 128); alone on the chip 512 is 2-6 % faster (fewer
                               part-filled blocks, larger shares) */
#endif
#ifndef PT_POOL_BINS
#define PT_POOL_BINS 8 /* output bins per workgroup (<= 64: one lane each in pt_pool_push); octant-keyed scenes use the first 8 */
#endif
#ifndef PT_POOL_RUN
#define PT_POOL_RUN 32 /* a workgroup's share of the input: runs of this many consecutive chunks, dealt round-robin; its waves take
                          the chunks of a run one by one (LDS counter), so they shade neighbouring chunks at the same time and a block of
                          survivors holds rays that were neighbours -- the next trace launch walks them together (single chunks dealt
                          round-robin cost that launch 8 %).  Runs handed to single waves from device counters: shade +18 %. */
#endif
#ifndef PT_POOL_MIN_CHUNKS
#define PT_POOL_MIN_CHUNKS 16 /* raw chunks per wave below which fewer workgroups take part (the drain costs up to 5 part-filled steps) */
#endif
#define PT_POOL_NO_BLOCK 0xfffffu

/* blk_list / blk_n (LDS, or null): every block this workgroup takes from the output queue is also noted there (PtSolo: the workgroup
 * reads its own survivors back in the next bounce of the same launch) */
template <bool EMIT>
__device__ __forceinline__ void pt_pool_push(const PtSceneDev& sc, const PtQueue& out, const PtShadeOut& so, uint32_t* lds_out,
                                             uint32_t* blk_list = nullptr, uint32_t* blk_n = nullptr, bool one_bin = false) {
  const int lane = pt_lane();
  /* one_bin (wave-uniform): a workgroup whose rays would not fill a block does not spread them over eight (PtSolo) */
  const int bin = one_bin ? 0 : pt_bin_key<PT_POOL_BINS>(sc, so.n_o, so.n_d);
  if (__ballot(so.keep) == 0) return;
  /* all bins at once: lane b < 8 holds bin b's survivor count and makes its reservation -- one LDS atomic instruction */
  uint32_t rank = 0, kk = 0;
#pragma unroll
  for (int b = 0; b < PT_POOL_BINS; ++b) {
    const unsigned long long m = __ballot(so.keep && bin == b);
    if (bin == b) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == b) kk = (uint32_t)__popcll(m);
  }
  uint32_t st = 0;
  if (kk != 0) st = __hip_atomic_fetch_add(lds_out + lane, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  /* a reservation that does not fit its block (pos + k > PT_POOL_BLOCK): rare, one bin at a time, wave-uniform */
  unsigned long long slow = __ballot(kk != 0 && (st & 0xfffu) + kk > (uint32_t)PT_POOL_BLOCK);
  uint32_t blk2 = 0, head = 0xffffffffu; /* lane b: the second block of a reservation that crossed into it, and how much went to the first */
  while (slow != 0) {
    const int b = __ffsll((long long)slow) - 1;
    slow &= slow - 1;
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)kk, b);
    uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)st, b);
    for (;;) {
      const uint32_t pos = sb & 0xfffu, blk = sb >> 12;
      if (pos + k <= (uint32_t)PT_POOL_BLOCK) break; /* a retry that fits */
      if (pos <= (uint32_t)PT_POOL_BLOCK) {
        /* this reservation crosses the end of the block (or finds it exactly full, or finds no block yet): fill it up and
         * bring the next one.  Everybody else sees pos > PT_POOL_BLOCK until the new word is stored. */
        uint32_t nb = 0;
        if (lane == 0) {
          nb = atomicAdd(out.count, (uint32_t)PT_POOL_BLOCK) / (uint32_t)PT_POOL_BLOCK;
          if (blk_list) blk_list[__hip_atomic_fetch_add(blk_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = nb;
        }
        nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        const uint32_t h = (uint32_t)PT_POOL_BLOCK - pos;
        if (lane == 0) __hip_atomic_store(lds_out + b, (nb << 12) | (k - h), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == b) { blk2 = nb; head = h; }
        break;
      }
      /* the block is being replaced by another wave: wait for the new one, reserve again */
      uint32_t cur;
      do {
        __builtin_amdgcn_s_sleep(2);
        cur = __hip_atomic_load(lds_out + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } while ((cur >> 12) == blk);
      uint32_t again = 0;
      if (lane == 0) again = __hip_atomic_fetch_add(lds_out + b, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)again);
    }
    if (lane == b) st = sb;
  }
  /* every lane picks up its bin's reservation */
  const uint32_t my_st = (uint32_t)__shfl((int)st, bin, 64);
  const uint32_t my_head = (uint32_t)__shfl((int)head, bin, 64);
  const uint32_t my_blk2 = (uint32_t)__shfl((int)blk2, bin, 64);
  if (so.keep) {
    const uint32_t dst = rank < my_head ? (my_st >> 12) * (uint32_t)PT_POOL_BLOCK + (my_st & 0xfffu) + rank
                                        : my_blk2 * (uint32_t)PT_POOL_BLOCK + (rank - my_head);
    pt_q_store<EMIT>(out, dst, so.n_o, so.n_d, so.n_attn, so.n_emit, so.id, so.offset);
  }
}

template <bool EMIT, bool PRIMARY>
__global__ __launch_bounds__(PT_POOL_THREADS, PT_SHADE_WAVES) void k_shade_pool(PtSceneDev sc, PtQueue q, PtHits hits, PtQueue out, PtContrib contrib,
                                                const double* __restrict__ alpha, int bounce, int last_bounce,
                                                PtGenParams g, uint32_t n_primary, uint32_t* work) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_pool_raw[]; /* [waves][PT_N_SHADE_CAT][128] x (queue index, hit slot) */
  __shared__ uint32_t lds_out[PT_POOL_BINS];
  __shared__ uint32_t lds_chunk_ctr, lds_done;
#if PT_SHADE_PRIO
  __builtin_amdgcn_s_setprio(PT_SHADE_PRIO);
#endif
  const int lane = pt_lane(), wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
  const uint32_t n = PRIMARY ? n_primary : *q.count;
  const uint32_t total_chunks = (uint32_t)(((unsigned long long)n + PT_WAVE - 1) / PT_WAVE);
  uint32_t n_wg = total_chunks / (uint32_t)(PT_POOL_MIN_CHUNKS * nw);
  n_wg = n_wg < 1u ? 1u : (n_wg > gridDim.x ? gridDim.x : n_wg);
  if (blockIdx.x >= n_wg) return; /* workgroup-uniform */
  if (threadIdx.x == 0) { lds_chunk_ctr = 0u; lds_done = 0u; }
  if (threadIdx.x < PT_POOL_BINS) lds_out[threadIdx.x] = (PT_POOL_NO_BLOCK << 12) | (uint32_t)PT_POOL_BLOCK; /* "full": the first push brings a block */
  __syncthreads(); /* the only workgroup barrier */
  uint2 (*pool)[128] = (uint2 (*)[128])lds_pool_raw + (size_t)wave * PT_N_SHADE_CAT;
  uint32_t cnt[PT_N_SHADE_CAT];
#pragma unroll
  for (int k = 0; k < PT_N_SHADE_CAT; ++k) cnt[k] = 0u;
  /* raw chunks are classified through a two-stage pipeline, so that neither of the two dependent loads (hit slot, then the
   * slot's category) is waited for: stage B holds a chunk whose slots are on their way, stage C one whose categories are */
  uint32_t iB = 0u, iC = 0u;
  int slotB = PT_SLOT_HOLE, slotC = PT_SLOT_HOLE, catC = PT_CAT_NONE;
  bool haveB = false, haveC = false; /* wave-uniform */
#define PT_POOL_TAKE_B() do { \
    uint32_t unit_ = 0u; \
    if (lane == 0) unit_ = __hip_atomic_fetch_add(&lds_chunk_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    unit_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit_); \
    unit_ = (unit_ / PT_POOL_RUN) * (n_wg * PT_POOL_RUN) + blockIdx.x * PT_POOL_RUN + (unit_ % PT_POOL_RUN); \
    haveB = unit_ < total_chunks; \
    iB = unit_ * PT_WAVE + (uint32_t)lane; \
    slotB = PT_SLOT_HOLE; \
    if (haveB) { \
      bool valid_ = iB < n; \
      if (PRIMARY && valid_) valid_ = pt_primary_decode(g, iB).valid; \
      if (valid_) slotB = hits.slot[iB]; \
    } \
  } while (0)
#define PT_POOL_ADVANCE() do { \
    iC = iB; haveC = haveB; slotC = slotB; \
    catC = slotB == PT_SLOT_HOLE ? PT_CAT_NONE : (slotB < 0 ? PT_CAT_MISS : (int)sc.slot_cat[slotB]); \
    PT_POOL_TAKE_B(); \
  } while (0)
  PT_POOL_TAKE_B();
  PT_POOL_ADVANCE();
  bool more = haveC;
#if PT_SHADE_TIMING
  unsigned long long tm_refill = 0, tm_entry = 0, tm_push = 0, tm_steps = 0, tm_live = 0;
  const unsigned long long tm_begin = __builtin_readcyclecounter();
  unsigned long long tm_t = tm_begin;
#define PT_TM(var, since) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); var += now_ - since; since = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PT_TM(var, since) do { } while (0)
#endif
  for (;;) {
    /* the fullest list that holds a whole step; once the input is exhausted, the fullest list */
    int c = -1;
    uint32_t best = more ? (uint32_t)(PT_WAVE - 1) : 0u;
#pragma unroll
    for (int k = 0; k < PT_N_SHADE_CAT; ++k)
      if (cnt[k] > best) { best = cnt[k]; c = k; }
    if (c >= 0) {
      const uint32_t take = best < (uint32_t)PT_WAVE ? best : (uint32_t)PT_WAVE;
      const uint32_t start = best - take;
      const bool live = (uint32_t)lane < take;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); /* this wave's own pushes below */
      uint32_t i = 0u;
      int sl = -1;
      PtShadeOut so;
      so.keep = false;
      /* one category per step (wave-uniform `c`): each case reads its own list and runs its own specialisation */
#define PT_POOL_STEP(K, ...)                                                                                               \
  case K: {                                                                                                                \
    cnt[K] = start;                                                                                                        \
    if (live) {                                                                                                            \
      const uint2 e = pool[K][start + lane];                                                                               \
      i = e.x;                                                                                                             \
      sl = (int)e.y;                                                                                                       \
    }                                                                                                                      \
    pt_shade_entry<EMIT, PRIMARY, K>(sc, q, hits, contrib, alpha, bounce, last_bounce, g, i, live, so, ##__VA_ARGS__);    \
  } break;
      switch (c) {
        PT_POOL_STEP(PT_CAT_MISS)
        PT_POOL_STEP(PT_CAT_LAMBERT_SOLID, sl)
        PT_POOL_STEP(PT_CAT_LAMBERT_CHECKER, sl)
        PT_POOL_STEP(PT_CAT_METAL, sl)
        PT_POOL_STEP(PT_CAT_DIELECTRIC, sl)
        default: break;
      }
#undef PT_POOL_STEP
      PT_TM(tm_entry, tm_t);
      if (c != PT_CAT_MISS && !last_bounce) pt_pool_push<EMIT>(sc, out, so, lds_out);
      PT_TM(tm_push, tm_t);
#if PT_SHADE_TIMING
      tm_steps++;
      tm_live += take;
#endif
      continue;
    }
    if (!more) break;
#pragma unroll
    for (int k = 0; k < PT_N_SHADE_CAT; ++k) {
      const unsigned long long m = __ballot(catC == k);
      if (catC == k) pool[k][cnt[k] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(iC, (uint32_t)slotC);
      cnt[k] += (uint32_t)__popcll(m);
    }
    PT_POOL_ADVANCE();
    more = haveC;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    PT_TM(tm_refill, tm_t);
  }
#if PT_SHADE_TIMING
  if (lane == 0) {
    uint32_t* tw = work + 16; /* diagnostic build only */
    const unsigned long long life = __builtin_readcyclecounter() - tm_begin;
    atomicAdd(tw + 0, (uint32_t)(tm_refill >> 8));
    atomicAdd(tw + 1, (uint32_t)(tm_entry >> 8));
    atomicAdd(tw + 2, (uint32_t)(tm_push >> 8));
    atomicAdd(tw + 3, (uint32_t)tm_steps);
    atomicAdd(tw + 4, (uint32_t)(life >> 8));
    atomicAdd(tw + 5, 1u);
    atomicAdd(tw + 6, (uint32_t)tm_live);
    atomicMax(tw + 7, (uint32_t)(life >> 8));
  }
#endif
#undef PT_TM
#undef PT_POOL_TAKE_B
#undef PT_POOL_ADVANCE
  if (last_bounce) return;
  /* the workgroup's last wave marks what is left of its blocks as holes */
  uint32_t done = 0u;
  if (lane == 0) done = __hip_atomic_fetch_add(&lds_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  done = (uint32_t)__builtin_amdgcn_readfirstlane((int)done);
  if (done != (uint32_t)(nw - 1)) return;
  for (int b = 0; b < PT_POOL_BINS; ++b) {
    const uint32_t st = __hip_atomic_load(lds_out + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t blk = st >> 12, pos = st & 0xfffu;
    if (blk == PT_POOL_NO_BLOCK) continue;
    for (uint32_t e = pos + (uint32_t)lane; e < (uint32_t)PT_POOL_BLOCK; e += PT_WAVE)
      out.ray[(size_t)blk * PT_POOL_BLOCK + e].dx = __hiloint2double((int)PT_HOLE_HI, 0);
  }
}

/* ------------------------------------------------------------------ one kernel per bounce
 * k_trace and k_shade_pool of two batches share every CU so that one batch's vector-bound walk fills the other's waits for
 * memory -- but a 128-VGPR shade wave that waits two thirds of its life holds its quarter of the register file while it
 * does, and the split 4 trace + 2 shade waves per SIMD halves the shade stage's own latency hiding (the frame gained 6 % over
 * running the two kernels one after the other).  k_bounce is both stages in ONE wave: it walks a chunk of
 * 64 queued rays (pt_trace_ray with k_trace's tail cut; the walks it cuts short are pooled per workgroup in LDS), files the finished rays in its
 * per-category pools, and whenever a pool holds 64 entries shades them (pt_shade_entry, pt_pool_push exactly as in
 * k_shade_pool).  Every wave of the CU then spends most of its life walking, the waits of a wave that shades are covered by
 * the three others on its SIMD, the hit slot never goes through memory, and a bounce is one launch.  One 1024-thread
 * workgroup per CU: [stacks][scene image][16 waves x 5 pools][parked walks] in LDS.  Same tests in the same order per ray, same
 * arithmetic per segment: the results are those of the two-kernel path bit for bit. */
#ifndef PT_BOUNCE_THREADS
#define PT_BOUNCE_THREADS 1024
#endif
#ifndef PT_BOUNCE_THREADS_GLOBAL
#define PT_BOUNCE_THREADS_GLOBAL 1024 /* k_bounce workgroup on scenes walked from HBM / L2: nothing is shared but the output blocks and the chunk
                                         hand-out, and the larger the group that shares them the better -- ganesha-like frame 28.4 (256) / 28.0 (512) /
                                         26.1 ms (1024) against 28.95 ms for k_trace + k_shade_pool */
#endif
#ifndef PT_BOUNCE_WAVES
#define PT_BOUNCE_WAVES 4 /* waves per SIMD asked of the register allocator (PT_BOUNCE_THREADS / 256); 768 threads / 3 waves: +16 % */
#endif
#ifndef PT_BOUNCE_DIV_LOOP
#define PT_BOUNCE_DIV_LOOP(MODE) ((MODE) == PT_MODE_SIMD || (PT_SWZ_SIGNSEL && PT_SWZ_TAGGED)) /* the walk phase's node loop as one divergent
                                                              loop (pt_trace_ray DIV_LOOP): headline -0.7 ... -1.1 %; cornell (Array_leaf) +1 % with
                                                              the untagged links of round 4, -1.5 % with tagged ones (no ballot of "a lane holds a
                                                              leaf" per turn any more: profiles/r05_ab_divloop_array_and_global_thresholds.txt) */
#endif
#ifndef PT_BOUNCE_MIN_CHUNKS
#define PT_BOUNCE_MIN_CHUNKS 2 /* chunks per wave below which fewer workgroups take part (k_shade_pool: 16 -- there a chunk is a few microseconds) */
#endif
#ifndef PT_SHADE_LDS_GEOM
#define PT_SHADE_LDS_GEOM 1 /* k_bounce's shade steps read slot kinds / spheres / triangles from the LDS image instead of global memory */
#endif
#ifndef PT_LDS_CAT
#define PT_LDS_CAT 1 /* k_bounce reads a finished ray's shading category from the LDS copy (PtSceneView.cat); 0: from global memory */
#endif
#ifndef PT_DIAG_FLOOR
#define PT_DIAG_FLOOR 0 /* diagnostic builds only: 1 = k_bounce returns once the scene image is in LDS, 2 = at once (the launch floor) */
#endif
#ifndef PT_BOUNCE_FENCE_WG
#define PT_BOUNCE_FENCE_WG 0 /* 1: always workgroup-scope fences (s_waitcnt vmcnt(0)) around the wave's own hit / parked records instead of wavefront
                                scope; at run time: PTX_BOUNCE_FENCE_WG=1 (k_bounce's fence_wg argument) */
#endif
/* LDS_SCENE = false (round 5): the same kernel for scenes walked from HBM / L2 over the per-octant node image (PtThreadOctTag; the
 * host launches it when PtSceneDev.nodes32o exists): the walk is k_trace's (threaded, binary32 filter, chunk cut at
 * PT_TAIL_CUT_GLOBAL with the third parked word for 32-bit node indices and slots, camera rays walked one per lane and parked like
 * any other), the pools and shade steps are the ones above, nothing of the scene is copied to LDS but the pre-tested floor
 * triangles, and the hit slot no longer travels through memory between two launches. */
/* SOLO (round 5): the small launches at the end of a batch.  Once a bounce's input has shrunk to PtSolo.max_entries queue entries, a
 * launch is a fixed cost -- the gap to the previous launch, the scene image built in LDS again, a handful of waves with one chunk
 * each and their drain -- and a frame of few pixels (one rank's share of an 8-rank job, the reference's 600 x 300 command) consists
 * of little else: 14 of its 16 launches per batch sat on that floor.  Paths are independent, so nothing requires the bounces of
 * DIFFERENT workgroups to stay in step: the first launch that finds its input that small runs ALL remaining bounces -- every
 * workgroup keeps its static share, notes the output blocks it takes (pt_pool_push), and after a workgroup barrier reads exactly
 * those blocks back as the next bounce's input, the scene image still in LDS.  It leaves its bounce number + 1 in PtSolo.flag; the
 * batch's later launches find it and return at once.  Workgroups drift apart in bounce number, so nothing written during the
 * launch may be overwritten during it: each of the two queue buffers hands out blocks from ONE cursor for the whole launch (the
 * buffer that holds the launch's input starts behind it), and the launch runs solo only if the buffers are large enough for
 * every remaining bounce's survivors and holes in the worst case.  Same walks, same shade steps, same sampler dimensions per bounce: the
 * results do not depend on the order of a queue, so they are the step-by-step launches' bit for bit. */
/* a workgroup-wide lock in LDS for the few-instruction critical sections of the parked-walk pool (every lane of the wave calls both) */
__device__ __forceinline__ void pt_lds_lock(uint32_t* l) {
  if (pt_lane() == 0)
    while (__hip_atomic_exchange(l, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void pt_lds_unlock(uint32_t* l) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (pt_lane() == 0) __hip_atomic_store(l, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* entries the parked-walk pool of a k_bounce workgroup of nw waves must hold: a wave takes a fresh chunk only while fewer than 64
 * walks are parked and parks fewer than the cut when it ends; a resumed chunk takes 64 out before it can put any back */
#define PT_PARK_CAP(nw, cut) (PT_WAVE + (nw) * (cut))
struct PtSolo {
  uint32_t* flag;       /* per batch, zero at its start; null = never run solo */
  uint32_t max_entries; /* run solo when the input queue holds at most this many entries (0 = never) */
  int32_t max_bounces;
  uint32_t cap_entries; /* capacity of each of the two queues, in entries */
};
#define PT_BOUNCE_POOL_ENTRY_BYTES(LDS_SCENE_) ((LDS_SCENE_) ? 6u : 8u) /* k_bounce's pool entries: 32-bit queue index + 16- / 32-bit slot */
#ifndef PT_SOLO_ONE_BIN_CHUNKS
#define PT_SOLO_ONE_BIN_CHUNKS 2 /* a workgroup whose input of a solo turn is at most this many chunks per wave puts all survivors into one bin */
#endif
#ifndef PT_SOLO_MAX_BLOCKS
#define PT_SOLO_MAX_BLOCKS 256 /* output blocks a workgroup can note per bounce; a launch whose shares could need more does not run solo */
#endif
template <int MODE, bool COUNT, bool EMIT, bool PRIMARY, bool LDS_SCENE = true, bool SOLO_T = false>
__global__ __launch_bounds__(PT_BOUNCE_THREADS, PT_BOUNCE_WAVES) void k_bounce(PtSceneDev sc, PtQueue q, PtHits hits, PtQueue out, PtContrib contrib,
                                                                 const double* __restrict__ alpha, int bounce, int last_bounce, PtGenParams g,
                                                                 uint32_t n_primary, int stack_depth, uint32_t pool_off,
                                                                 PtCounters* counters, int fence_wg, PtSolo solo) {
  extern __shared__ __attribute__((aligned(64))) unsigned char lds_raw[];
  __shared__ uint32_t lds_out[PT_POOL_BINS];
  __shared__ uint32_t lds_chunk_ctr, lds_done;
  __shared__ uint32_t lds_park_n, lds_park_lock; /* the workgroup's parked walks (below) */
  /* (a template switch, launched only when PTX_SOLO_ENTRIES asks for it: the loop around the bounces costs the kernel 6 - 11 VGPRs it does
   * not have -- the instantiations without it keep 0 spilled registers) */
  constexpr bool SOLO = SOLO_T && !PRIMARY && PT_DIAG == 0 && PT_DIAG_FLOOR == 0;
  __shared__ uint32_t lds_blk[SOLO ? 2 : 1][SOLO ? PT_SOLO_MAX_BLOCKS : 1]; /* the blocks this workgroup wrote in the bounce before / is writing now */
  __shared__ uint32_t lds_nblk[2];
  const int lane = pt_lane(), wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
  typedef typename std::conditional<LDS_SCENE, uint16_t, PtThreadOctTag>::type StackT;
  StackT* stack = (StackT*)(lds_raw + (size_t)wave * PT_WAVE_STACK_BYTES(LDS_SCENE, StackT, stack_depth));
  const uint32_t n = PRIMARY ? n_primary : *q.count;
  const uint32_t total_chunks = (uint32_t)(((unsigned long long)n + PT_WAVE - 1) / PT_WAVE);
  uint32_t n_wg = total_chunks / (uint32_t)(PT_BOUNCE_MIN_CHUNKS * nw);
  n_wg = n_wg < 1u ? 1u : (n_wg > gridDim.x ? gridDim.x : n_wg);
  bool solo_on = false; /* launch-uniform */
  if (SOLO && solo.flag != nullptr) {
    const uint32_t f = __hip_atomic_load(solo.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f != 0u && f != (uint32_t)bounce + 1u) return; /* an earlier launch of this batch ran (or is the one running) this bounce */
    /* the largest share of this launch in output blocks: runs of PT_POOL_RUN chunks dealt round-robin, every entry a survivor, one
     * part-filled block per bin */
    const uint32_t runs = (total_chunks + PT_POOL_RUN - 1u) / PT_POOL_RUN;
    const uint32_t share_blocks = ((runs + n_wg - 1u) / n_wg) * PT_POOL_RUN * PT_WAVE / PT_POOL_BLOCK + PT_POOL_BINS;
    /* each buffer takes the output of every second remaining bounce: at most the launch's input + one part-filled block per
     * (workgroup, bin) each time, the input's buffer behind the input */
    const unsigned long long per_bounce = (unsigned long long)n + (unsigned long long)n_wg * PT_POOL_BINS * PT_POOL_BLOCK;
    const unsigned long long worst = (unsigned long long)((solo.max_bounces - bounce) / 2 + 1) * per_bounce + (unsigned long long)n + PT_POOL_BLOCK;
    solo_on = !last_bounce && n <= solo.max_entries && share_blocks <= (uint32_t)PT_SOLO_MAX_BLOCKS && worst <= (unsigned long long)solo.cap_entries;
    if (solo_on && blockIdx.x == 0 && threadIdx.x == 0) {
      __hip_atomic_store(solo.flag, (uint32_t)bounce + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (COUNT) atomicAdd(&counters->solo, 1ull);
    }
  }
  if (blockIdx.x >= n_wg) return; /* workgroup-uniform */
#if PT_DIAG_FLOOR == 2
  if (gridDim.x > 0) return; /* diagnostic build (tools/README.md): what a launch costs before it does anything */
#endif
  if (threadIdx.x == 0) { lds_chunk_ctr = 0u; lds_done = 0u; lds_nblk[0] = lds_nblk[1] = 0u; lds_park_n = 0u; lds_park_lock = 0u; }
  if (threadIdx.x < PT_POOL_BINS) lds_out[threadIdx.x] = (PT_POOL_NO_BLOCK << 12) | (uint32_t)PT_POOL_BLOCK; /* "full": the first push brings a block */
  PtSceneView sv = pt_scene_view<MODE, LDS_SCENE, StackT>(sc, lds_raw, stack_depth); /* LDS_SCENE: ends with the workgroup's only barrier (SOLO: per bounce, two more) */
  if (!LDS_SCENE) {
    __shared__ double lds_floor[PT_FLOOR_LDS * 10];
    if (MODE == PT_MODE_ARRAY && sc.n_floor > 0) { /* the pre-tested floor triangles: see PtSceneView.floor_lds */
      sv.n_floor_lds = sc.n_floor < PT_FLOOR_LDS ? sc.n_floor : PT_FLOOR_LDS;
      sv.floor_lds = (const __attribute__((address_space(3))) double*)lds_floor;
      for (int k = threadIdx.x; k < sv.n_floor_lds * 10; k += blockDim.x) lds_floor[k] = sc.tri[(size_t)sc.n_slots * 10 + k];
    }
    __syncthreads(); /* the workgroup's only barrier */
  }
#if PT_DIAG_FLOOR == 1
  if (gridDim.x > 0) return; /* diagnostic build: launch + the scene image in LDS, nothing else */
#endif
  /* the shade steps read the slots' kinds and geometry where the walk reads them: the LDS image (generic pointers: flat loads) */
  static_assert(!PT_SHADE_LDS_GEOM || PT_LDS_SPH_DOUBLES == 4, "the shade step strides sphere records by 4 doubles (6: measured, no gain -- the scan's bank conflicts are not what binds)");
  PtSceneDev scl = sc;
  if (PT_SHADE_LDS_GEOM && LDS_SCENE) {
    scl.slot_kind = sv.kind;
    scl.sph = sv.sph;
    scl.tri = sv.tri;
  }
  /* the wave's five pools: queue indices (32 bits) and hit slots apart -- an LDS-resident scene has fewer than 65536 slots, so
   * its slots take 16 bits: 6 bytes per entry instead of 8, 20 KB of a 1024-thread workgroup's LDS for the scene image */
  typedef typename std::conditional<LDS_SCENE, uint16_t, uint32_t>::type PoolSlotT;
  uint32_t (*pool_i)[128] = (uint32_t (*)[128])(lds_raw + pool_off) + (size_t)wave * PT_N_SHADE_CAT;
  PoolSlotT (*pool_s)[128] = (PoolSlotT (*)[128])(lds_raw + pool_off + (size_t)nw * PT_N_SHADE_CAT * 128 * sizeof(uint32_t)) + (size_t)wave * PT_N_SHADE_CAT;
  uint32_t cnt[PT_N_SHADE_CAT];
#pragma unroll
  for (int k = 0; k < PT_N_SHADE_CAT; ++k) cnt[k] = 0u;
  constexpr int CUT = LDS_SCENE ? PT_TAIL_CUT : PT_TAIL_CUT_GLOBAL;
  constexpr bool TAIL = CUT > 0 && !(PRIMARY && LDS_SCENE); /* LDS scenes: camera rays walk as a packet (pt_trace_packet) and finish together */
  constexpr bool TAIL_UV = TAIL && MODE == PT_MODE_ARRAY;
  constexpr bool TAIL_W = TAIL && !LDS_SCENE; /* 32-bit node index and slot: a third 16 bytes (as in k_trace) */
  /* Parked walks (PtTailCtl) are pooled per WORKGROUP in LDS, behind the shade pools: whichever wave next looks for work and finds
   * 64 of them walks them as a dense chunk.  (Round 4 kept them per wave, in global memory: a wave had to collect 48 of its own
   * stragglers -- three or four cut chunks -- before it could resume any, which is what held the cut at 16 rays.)  16-byte entries
   * {queue index, node | slot << 16, t}; + {u, v} for triangle hits; + {node, slot} as 32-bit words on the walk from HBM / L2. */
  const uint32_t park_cap = (uint32_t)PT_PARK_CAP(nw, CUT);
  uint4* const park0 = (uint4*)(lds_raw + pool_off + (size_t)nw * PT_N_SHADE_CAT * 128 * PT_BOUNCE_POOL_ENTRY_BYTES(LDS_SCENE));
  uint4* const park_uv = park0 + park_cap;
  uint4* const park_w = park_uv + (TAIL_UV ? park_cap : 0u);
  bool more = true;    /* wave-uniform: the workgroup's share of the queue is not exhausted */
  unsigned long long c_nodes = 0, c_prims = 0, c_floor = 0, c_seg = 0, c_filter[2] = {0, 0}; /* COUNT: as in k_trace */
  /* SOLO: the bounce this workgroup is at, its queues, and whether its input is the block list it noted in the bounce before */
  uint32_t own_chunks = 0u, cur_list = 0u; /* own_blocks: chunks of lds_blk[cur_list]; pushes note into lds_blk[cur_list ^ 1] */
  bool own_blocks = false;
  uint32_t* const solo_cursor[2] = {out.count, out.count + 1}; /* the block cursors of the buffer written first / of the input's buffer */
  uint32_t solo_turn = 0u;
  if (SOLO && solo_on) hits.t += (size_t)(bounce & 1) * hits.t_parity_stride;
  /* PT_DIAG == 5 (tools/diag_phases.py; counting renders of diagnostic builds, queued rays' launches): where a wave's life goes,
   * in ticks of the 100 MHz clock summed over waves -- nodes = walks, prims = shade steps (with their pushes), floor = everything
   * else in the loop (chunk hand-out, ray loads, filing, parking), segments = the whole loop; undecided / fallback_steps = the
   * number of shade steps / walks */
  constexpr bool DIAG_T = COUNT && (((PT_DIAG == 5 || PT_DIAG == 6 || PT_DIAG == 7) && !PRIMARY) || (PT_DIAG == 8 && PRIMARY)); /* 8: as 5, the camera rays' launches */ /* 6: prims = the walks' leaf phases instead of the shade steps; 7: prims = the pushes alone */
  unsigned long long tm_last = DIAG_T ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long tm_begin = tm_last;
#define PT_TM5(var) do { if (DIAG_T) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); (var) += now_ - tm_last; tm_last = now_; __builtin_amdgcn_sched_barrier(0); } } while (0)
 for (;;) { /* (one turn per bounce; a launch that does not run solo leaves after the first) */
  for (;;) {
    PT_TM5(c_floor);
    const uint32_t park_hint = TAIL ? __hip_atomic_load(&lds_park_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u; /* (wave-uniform: one LDS word) */
    const bool input_left = more || park_hint > 0;
    /* the fullest pool that holds a whole step; once nothing is left to walk, the fullest pool */
    int c = -1;
    uint32_t best = input_left ? (uint32_t)(PT_WAVE - 1) : 0u;
#pragma unroll
    for (int k = 0; k < PT_N_SHADE_CAT; ++k)
      if (cnt[k] > best) { best = cnt[k]; c = k; }
    if (c >= 0) {
      const uint32_t take = best < (uint32_t)PT_WAVE ? best : (uint32_t)PT_WAVE;
      const uint32_t start = best - take;
      const bool live = (uint32_t)lane < take;
      uint32_t i = 0u;
      int sl = -1;
      PtShadeOut so;
      so.keep = false;
#define PT_POOL_STEP(K, ...)                                                                                               \
  case K: {                                                                                                                \
    cnt[K] = start;                                                                                                        \
    if (live) {                                                                                                            \
      i = pool_i[K][start + lane];                                                                                         \
      sl = (int)pool_s[K][start + lane];                                                                                   \
      if (LDS_SCENE && sl == 0xffff) sl = -1; /* (misses are filed with slot -1) */                                        \
    }                                                                                                                      \
    pt_shade_entry<EMIT, PRIMARY, K>((PT_SHADE_LDS_GEOM && LDS_SCENE) ? scl : sc, q, hits, contrib, alpha, bounce, last_bounce, g, i, live, so, ##__VA_ARGS__); \
  } break;
      switch (c) {
        PT_POOL_STEP(PT_CAT_MISS)
        PT_POOL_STEP(PT_CAT_LAMBERT_SOLID, sl)
        PT_POOL_STEP(PT_CAT_LAMBERT_CHECKER, sl)
        PT_POOL_STEP(PT_CAT_METAL, sl)
        PT_POOL_STEP(PT_CAT_DIELECTRIC, sl)
        default: break;
      }
#undef PT_POOL_STEP
      if (PT_DIAG == 7) PT_TM5(c_floor);
      if (c != PT_CAT_MISS && !last_bounce) {
        if (SOLO && solo_on) pt_pool_push<EMIT>(sc, out, so, lds_out, lds_blk[SOLO ? (cur_list ^ 1u) : 0u], &lds_nblk[cur_list ^ 1u],
                                                own_blocks && own_chunks <= (uint32_t)(PT_SOLO_ONE_BIN_CHUNKS * nw));
        else pt_pool_push<EMIT>(sc, out, so, lds_out);
      }
      if (PT_DIAG == 6) PT_TM5(c_floor);
      else PT_TM5(c_prims);
      if (DIAG_T) c_filter[0] += (lane == 0);
      continue;
    }
    if (!input_left) break;
    /* walk: 64 parked rays once enough have gathered (or nothing else is left), else the next chunk of the share */
    bool resume = false, valid = false;
    uint32_t i = 0;
    uint4 parked = make_uint4(0, 0, 0, 0), parked_uv = make_uint4(0, 0, 0, 0), parked_w = make_uint4(0, 0, 0, 0);
    if (TAIL && (park_hint >= (uint32_t)PT_WAVE || (!more && park_hint > 0))) {
      /* 64 parked walks (the most recently parked: a stack), or what is left once this wave's share of the input is exhausted */
      pt_lds_lock(&lds_park_lock);
      const uint32_t np = __hip_atomic_load(&lds_park_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t take = (np >= (uint32_t)PT_WAVE || !more) ? (np < (uint32_t)PT_WAVE ? np : (uint32_t)PT_WAVE) : 0u;
      const uint32_t base = np - take;
      if ((uint32_t)lane < take) {
        parked = park0[base + lane];
        if (TAIL_UV) parked_uv = park_uv[base + lane];
        if (TAIL_W) parked_w = park_w[base + lane];
      }
      if (lane == 0 && take) __hip_atomic_store(&lds_park_n, base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      pt_lds_unlock(&lds_park_lock);
      if (take == 0u) continue; /* (another wave was faster) */
      resume = true;
      valid = (uint32_t)lane < take;
      i = parked.x;
    } else {
      if (!more) continue; /* (only here when the hint and the pool disagreed a moment ago) */
      /* the next chunk of the workgroup's share (runs of PT_POOL_RUN consecutive chunks, dealt round-robin, as in k_shade_pool;
       * taking a chunk one turn ahead and touching its ray records cost 2 % in round 3 and 3.5 % (cornell 5 %) in round 4: a line
       * touched ~10 us early is evicted from the L2 again before the wave comes back for it, and is then fetched twice) */
      uint32_t unit = 0u;
      if (lane == 0) unit = __hip_atomic_fetch_add(&lds_chunk_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
      if (SOLO && own_blocks) { /* the workgroup's own blocks of the bounce before (block | entries written << 20), PT_POOL_BLOCK / 64 chunks each */
        more = unit < own_chunks;
        if (more) {
          const uint32_t e = lds_blk[SOLO ? cur_list : 0u][unit / (PT_POOL_BLOCK / PT_WAVE)];
          const uint32_t fill = e >> 20, off = (unit % (PT_POOL_BLOCK / PT_WAVE)) * PT_WAVE;
          if (off >= fill) continue; /* (wave-uniform) the unwritten part of a bin's last block: nothing to read, nothing was marked */
          i = (e & 0xfffffu) * (uint32_t)PT_POOL_BLOCK + off + (uint32_t)lane;
          valid = off + (uint32_t)lane < fill;
        }
      } else {
      unit = (unit / PT_POOL_RUN) * (n_wg * PT_POOL_RUN) + blockIdx.x * PT_POOL_RUN + (unit % PT_POOL_RUN);
      more = unit < total_chunks;
      if (more) {
        i = unit * PT_WAVE + (uint32_t)lane;
        valid = i < n;
      }
      }
    }
    V3 o = v3(0.0, 0.0, 0.0), d = v3(0.0, 0.0, -1.0); /* P3.origin */
    if (valid) {
      if (PRIMARY) {
        const PtPrimarySample ps = pt_primary_decode(g, i);
        valid = ps.valid;
        if (valid) d = pt_primary_dir(sc, g, ps, alpha);
      } else {
        pt_q_load_ray(q, i, o, d);
        if (pt_is_hole(d.x)) { /* never parked, so never seen on resume */
          valid = false;
          o = v3(0.0, 0.0, 0.0);
          d = v3(0.0, 0.0, -1.0);
        }
      }
    }
    if (COUNT && !DIAG_T && valid && !resume) c_seg++;
    PtTailCtl tc;
    tc.min_active = (TAIL && more) ? CUT : 0; /* the last chunks of a wave run to completion */
    tc.resume = resume && valid;
    tc.node = TAIL_W ? parked_w.x : (parked.y & 0xffffu);
    tc.slot = TAIL_W ? (int)parked_w.y : ((int)(parked.y >> 16) == 0xffff ? -1 : (int)(parked.y >> 16));
    tc.t = __hiloint2double((int)parked.w, (int)parked.z);
    tc.u = __hiloint2double((int)parked_uv.y, (int)parked_uv.x);
    tc.v = __hiloint2double((int)parked_uv.w, (int)parked_uv.z);
    tc.unfinished = false;
    PtTraceResult r;
    PT_TM5(c_floor);
    unsigned long long dg_n = 0, dg_p = 0, dg_f = 0; /* (diagnostic builds: the packet walk's own counters go nowhere) */
    if constexpr (PRIMARY && LDS_SCENE) r = pt_trace_packet<MODE, COUNT, true, true>(sc, sv, (uint32_t*)stack, valid, o, d, DIAG_T ? dg_n : c_nodes, DIAG_T ? dg_p : c_prims, DIAG_T ? dg_f : c_floor, DIAG_T ? nullptr : c_filter);
    else r = pt_trace_ray<MODE, COUNT, PRIMARY, StackT, LDS_SCENE, LDS_SCENE ? PT_BOUNCE_DIV_LOOP(MODE) : PT_TRACE_DIV_LOOP(false), LDS_SCENE && !COUNT && MODE == PT_MODE_SIMD /* (Array_leaf kernels have no registers to pin: cornell +0.5 %; the walk from HBM / L2 in assembly: +1.9 %, profiles/r05_ab_oct_asm.txt) */>(sc, sv, stack, o, d, c_nodes, c_prims, c_floor, valid, TAIL ? &tc : nullptr, DIAG_T ? nullptr : c_filter);
    PT_TM5(c_nodes);
    if (DIAG_T) c_filter[1] += (lane == 0);
    const bool park = TAIL && tc.unfinished;
    const bool done = valid && !park;
    int cat = PT_CAT_NONE;
    if (done) {
      /* the hit distance (and a triangle's barycentrics) reach the shade step through memory -- this wave's own L1 / L2
       * lines; the slot travels in the pool entry */
      if (MODE == PT_MODE_ARRAY && sc.has_triangles) {
        if (!PT_RECOMPUTE_HIT) hits.tuv[i] = make_double4(r.t, r.u, r.v, 0.0); /* (else the shade step recomputes it: PtHits) */
      } else hits.t[i] = r.t;
      cat = r.slot < 0 ? PT_CAT_MISS : (int)((PT_LDS_CAT || !LDS_SCENE) ? sv.cat : sc.slot_cat)[r.slot]; /* (walks from HBM / L2: sv.cat is sc.slot_cat) */
    }
#pragma unroll
    for (int k = 0; k < PT_N_SHADE_CAT; ++k) {
      const unsigned long long m = __ballot(cat == k);
      if (cat == k) {
        const uint32_t at = cnt[k] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        pool_i[k][at] = i;
        pool_s[k][at] = (PoolSlotT)r.slot;
      }
      cnt[k] += (uint32_t)__popcll(m);
    }
    if (TAIL) {
      const unsigned long long pm = __ballot(park);
      if (pm != 0) {
        pt_lds_lock(&lds_park_lock);
        const uint32_t base = __hip_atomic_load(&lds_park_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (park) {
          const uint32_t k = base + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
          park0[k] = make_uint4(i, tc.node | ((uint32_t)(r.slot < 0 ? 0xffff : r.slot) << 16),
                                (uint32_t)__double2loint(r.t), (uint32_t)__double2hiint(r.t));
          if (TAIL_W) park_w[k] = make_uint4(tc.node, (uint32_t)r.slot, 0u, 0u);
          if (TAIL_UV)
            park_uv[k] = make_uint4((uint32_t)__double2loint(r.u), (uint32_t)__double2hiint(r.u), (uint32_t)__double2loint(r.v), (uint32_t)__double2hiint(r.v));
        }
        if (lane == 0) __hip_atomic_store(&lds_park_n, base + (uint32_t)__popcll(pm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pt_lds_unlock(&lds_park_lock);
      }
    }
    /* this wave's own stores (hit records, parked states, pool entries) before its own later loads of them: program order
     * is enough for that (one wave's vector memory instructions reach the L1 / L2 in order), the compiler must not move them */
    /* fence_wg (PTX_BOUNCE_FENCE_WG=1, wave-uniform): workgroup-scope fences instead -- every store of the turn is waited for
     * (s_waitcnt vmcnt(0)) before the wave goes on.  The safety net should the in-order assumption ever fail on some part; the
     * parity tests run both (tests/test_gpu_parity.py). */
    if (PT_BOUNCE_FENCE_WG || fence_wg) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (!(SOLO && solo_on) || last_bounce) break;
  /* SOLO: this workgroup's next bounce.  Every wave has filed, shaded and pushed its last ray of this one when the barrier
   * opens; the unused tails of the part-filled blocks become holes, the blocks noted by the pushes become the input. */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* the survivors' records, for the other waves of this workgroup */
  __syncthreads();
  {
    /* how far each noted block was written: a bin's last block up to its cursor, every other one completely.  The next turn reads
     * exactly that much -- no holes are marked, none are looked for */
    const uint32_t nb = lds_nblk[cur_list ^ 1u];
    for (uint32_t t = threadIdx.x; t < nb; t += blockDim.x) {
      const uint32_t blk = lds_blk[SOLO ? (cur_list ^ 1u) : 0u][t];
      uint32_t fill = (uint32_t)PT_POOL_BLOCK;
      for (int b = 0; b < PT_POOL_BINS; ++b) {
        const uint32_t st = lds_out[b];
        if ((st >> 12) == blk) fill = st & 0xfffu;
      }
      lds_blk[SOLO ? (cur_list ^ 1u) : 0u][t] = blk | (fill << 20);
    }
    own_chunks = nb * (uint32_t)(PT_POOL_BLOCK / PT_WAVE);
  }
  __syncthreads(); /* everybody has read the bins and the list's length */
  if (threadIdx.x == 0) {
    lds_chunk_ctr = 0u;
    lds_nblk[cur_list] = 0u;
    /* the input's buffer is written from the second turn on: its cursor starts behind the launch's input (idempotent) */
    if (solo_turn == 0u) atomicMax(solo_cursor[1], ((n + (uint32_t)PT_POOL_BLOCK - 1u) / (uint32_t)PT_POOL_BLOCK) * (uint32_t)PT_POOL_BLOCK);
  }
  if (threadIdx.x < PT_POOL_BINS) lds_out[threadIdx.x] = (PT_POOL_NO_BLOCK << 12) | (uint32_t)PT_POOL_BLOCK;
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (own_chunks == 0u) break; /* (workgroup-uniform) none of this workgroup's paths is left */
  cur_list ^= 1u;
  own_blocks = true;
  more = true;
  ++bounce;
  last_bounce = bounce == solo.max_bounces - 1;
  { const PtQueue t_ = q; q = out; out = t_; }
  ++solo_turn;
  out.count = solo_cursor[solo_turn & 1u];
  hits.t += (bounce & 1) ? (ptrdiff_t)hits.t_parity_stride : -(ptrdiff_t)hits.t_parity_stride;
 }
  if (DIAG_T) {
    PT_TM5(c_floor);
    c_seg = (lane == 0) ? tm_last - tm_begin : 0ull;
    if (lane != 0) c_nodes = c_prims = c_floor = 0ull; /* wave-uniform quantities: one lane's copy */
  }
#undef PT_TM5
  if (COUNT && !(PT_DIAG != 0 && (PRIMARY != (PT_DIAG == 8)))) { /* (diagnostic builds measure the queued rays' launches only; 8: the camera rays') */
    c_nodes = pt_wave_sum(c_nodes);
    c_prims = pt_wave_sum(c_prims);
    c_floor = pt_wave_sum(c_floor);
    c_seg = pt_wave_sum(c_seg);
    c_filter[0] = pt_wave_sum(c_filter[0]);
    c_filter[1] = pt_wave_sum(c_filter[1]);
    if (lane == 0) {
      atomicAdd(&counters->nodes, c_nodes);
      atomicAdd(&counters->prims, c_prims);
      atomicAdd(&counters->floor, c_floor);
      atomicAdd(&counters->segments, c_seg);
      atomicAdd(&counters->undecided, c_filter[0]);
      atomicAdd(&counters->fallback_steps, c_filter[1]);
    }
  }
  if (last_bounce || (SOLO && solo_on)) return; /* (a solo launch leaves no queue behind: its paths have all ended) */
  /* the workgroup's last wave marks what is left of its blocks as holes */
  uint32_t fin = 0u;
  if (lane == 0) fin = __hip_atomic_fetch_add(&lds_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  fin = (uint32_t)__builtin_amdgcn_readfirstlane((int)fin);
  if (fin != (uint32_t)(nw - 1)) return;
  for (int b = 0; b < PT_POOL_BINS; ++b) {
    const uint32_t st = __hip_atomic_load(lds_out + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t blk = st >> 12, pos = st & 0xfffu;
    if (blk == PT_POOL_NO_BLOCK) continue;
    for (uint32_t e = pos + (uint32_t)lane; e < (uint32_t)PT_POOL_BLOCK; e += PT_WAVE)
      out.ray[(size_t)blk * PT_POOL_BLOCK + e].dx = __hiloint2double((int)PT_HOLE_HI, 0);
  }
}

/* ------------------------------------------------------------------ accumulate + film */
/* raw[pix] += contributions of this batch's passes, in pass order (the order render_tile's pass loop
 * feeds the film, integrator.ml:96) */
/* [p0, p1): the pixels of this launch -- the whole image, or one row slab of the frame's last batch (ptx_render: the film and the
 * copy of slab k to the host run while slab k + 1 is still being summed) */
__global__ __launch_bounds__(256) void k_accum(PtContrib contrib, long long npix, int n_pass, double* __restrict__ raw, long long p0, long long p1) {
  const long long p = p0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= p1) return;
  double r = raw[3 * p], g = raw[3 * p + 1], b = raw[3 * p + 2];
  for (int k = 0; k < n_pass; ++k) {
    const long long j = (long long)k * npix + p;
    const double4 c = contrib.rgbx[j];
    r = r + c.x;
    g = g + c.y;
    b = b + c.z;
  }
  raw[3 * p] = r;
  raw[3 * p + 1] = g;
  raw[3 * p + 2] = b;
}

struct PtFilm3 {
  double w[9];
};
/* Where image row y lives in the raw-sum buffer the film reads.  world <= 1: row y of a (height x width) image.
 * world > 1: the buffer is the GATHERED multi-rank layout [rank][pad_rows][width][3] -- rank r's compact rows exactly
 * as ptx_render_raw_device left them (interleaved bands r, r + world, ... of band_rows rows) -- so the film reads the
 * bands in place and rank 0 never un-permutes them. */
struct PtBandMap {
  int world, band_rows, pad_rows;
};
__device__ __forceinline__ long long pt_band_row(const PtBandMap& m, int y) {
  if (m.world <= 1) return y;
  const int band = y / m.band_rows;
  const int rank = band % m.world;
  const int local = (band / m.world) * m.band_rows + (y - band * m.band_rows);
  return (long long)rank * m.pad_rows + local;
}
/* Film_tile.write_pixel splats sample s at its own pixel q to q + (dx, dy) with weight k[dy][dx]
 * (film_tile.ml:23-45); stitch_tile drops what falls outside the image (integrator.ml:114-128).  As a
 * gather: P = sum_taps k[dy][dx] * S(P - (dx, dy)) over in-image neighbours, then sqrt(v * (1/spp)). */
__global__ __launch_bounds__(256) void k_film(const double* __restrict__ raw, int width, int height, double spp_inv,
                                              PtFilm3 kern, PtBandMap map, double* __restrict__ out, int row0, int row1) {
  const long long p = (long long)row0 * width + (long long)blockIdx.x * blockDim.x + threadIdx.x; /* rows [row0, row1) of the image */
  if (p >= (long long)width * row1) return;
  const int x = (int)(p % width), y = (int)(p / width);
  double r = 0.0, g = 0.0, b = 0.0;
  int k = 0;
  for (int dy = -1; dy <= 1; ++dy) {
    const int sy = y - dy;
    const long long srow = (sy >= 0 && sy < height) ? pt_band_row(map, sy) : 0;
    for (int dx = -1; dx <= 1; ++dx, ++k) {
      const int sx = x - dx;
      if (sx < 0 || sx >= width || sy < 0 || sy >= height) continue;
      const double* s = raw + (srow * width + sx) * 3;
      const double wgt = kern.w[k];
      r = pt_fma(wgt, s[0], r);
      g = pt_fma(wgt, s[1], g);
      b = pt_fma(wgt, s[2], b);
    }
  }
  out[3 * p] = pt_sqrt(r * spp_inv);
  out[3 * p + 1] = pt_sqrt(g * spp_inv);
  out[3 * p + 2] = pt_sqrt(b * spp_inv);
}

/* ------------------------------------------------------------------ unit entry points */
__global__ void k_lds_sample(const double* __restrict__ alpha, long long n, const int32_t* __restrict__ offsets,
                             const int32_t* __restrict__ dims, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pt_lds_get(alpha, offsets[i], dims[i]);
}

__global__ void k_math_eval(int fn, long long n, const double* __restrict__ a, const double* __restrict__ b,
                            double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0;
  double r;
  switch (fn) {
    case 0: r = pt_hypot(x, y); break;
    case 1: r = pt_sin(x); break;
    case 2: r = pt_cos(x); break;
    case 3: r = pt_acos(x); break;
    case 4: r = pt_atan2(x, y); break;
    case 5: r = pt_pow5(x); break;
    case 6: r = pt_sqrt(x); break;
    case 7: r = x / y; break;
    case 8: r = pt_fma(x, y, y); break;
    case 9: r = pt_rnorm3(x, y, x - y); break;             /* the fused forms against the oracle's nested literal ones */
    case 10: r = pt_rnorm_frame(1.0 + x, y, x - y); break;
    case 11: r = pt_sqrt_nonneg(x); break;
    case 12: r = pt_rcp_mid(x); break;                     /* mid-range operands only (pt_math.h) */
    case 13: r = pt_div_mid(x, y); break;
    case 14: r = pt_sqrt_mid(x); break;
    default: r = pt_nan(); break;
  }
  out[i] = r;
}

/* copies explicit rays into a queue (ptx_intersect_rays) */
__global__ void k_load_rays(long long n, const double* __restrict__ o, const double* __restrict__ d, PtQueue q) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pt_q_store_ray(q, (uint32_t)i, v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
}

#if PT_FILTER_DEBUG
/* Diagnostic builds only: the PRODUCTION box test of the walk from HBM / L2 (PtTraverser::begin + test_box on the tagged per-octant
 * record) on a list of (ray, node, closest hit so far) triples, and beside it the reference's binary64 quantities of the same test
 * (Bbox.is_hit, bbox.ml:40-56, the arithmetic of pt_slab_hit_fast: finite 1 / d only).  out[6 i ..] = u32, m2, lo64, hi64,
 * the filter's outcome (1 hit / 0 miss decided, -1 undecided), the outcome test_box returns (its fallback included). */
__global__ __launch_bounds__(256) void k_filter_error(PtSceneDev sc, long long n, const double* __restrict__ o3, const double* __restrict__ d3,
                                                      const int32_t* __restrict__ nodes, const double* __restrict__ tmax, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PtSceneView sv;
  sv.nodes = sc.nodes; sv.skip32 = sc.node_skip32; sv.nodes32 = (const unsigned char*)sc.nodes32; sv.nodes32o = (const unsigned char*)sc.nodes32o;
  sv.n_nodes = (uint32_t)sc.n_nodes; sv.swz_nodes = nullptr; sv.swz_root = 0u; sv.top = nullptr; sv.has_top = false; sv.sph = sc.sph; sv.tri = sc.tri;
  sv.kind = sc.slot_kind; sv.cat = sc.slot_cat; sv.nodes64 = nullptr; sv.floor_lds = nullptr; sv.n_floor_lds = 0;
  PtSceneDev scn = sc;
  scn.n_floor = 0; /* (no floor pre-test: the closest hit so far is the caller's) */
  const V3 o = v3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), d = v3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
  PtTraverser<PT_MODE_ARRAY, false, false, PtThreadOctTag, false> tr;
  unsigned long long cf = 0;
  tr.begin(scn, sv, o, d, cf);
  tr.r.t = tmax[i];
  tr.update_t32();
  const uint32_t k = (uint32_t)nodes[i];
  uint32_t na, nb, nr;
  const bool final_hit = tr.test_box(sv, tr.skip_off + k, na, nb, nr);
  const float u = tr.dbg_u, m2 = tr.dbg_m2;
  const PtNode* np = sc.nodes + k;
  const V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
  const double t0x = (np->mn[0] - o.x) * inv.x, t0y = (np->mn[1] - o.y) * inv.y, t0z = (np->mn[2] - o.z) * inv.z;
  const double t1x = (np->mx[0] - o.x) * inv.x, t1y = (np->mx[1] - o.y) * inv.y, t1z = (np->mx[2] - o.z) * inv.z;
  const double a = __builtin_fmax(__builtin_fmin(t0x, t1x), __builtin_fmax(__builtin_fmin(t0y, t1y), __builtin_fmin(t0z, t1z)));
  const double b = __builtin_fmin(__builtin_fmax(t0x, t1x), __builtin_fmin(__builtin_fmax(t0y, t1y), __builtin_fmax(t0z, t1z)));
  double* r = out + 6 * i;
  r[0] = (double)u;
  r[1] = (double)m2;
  r[2] = __builtin_fmax(0.0, a);
  r[3] = __builtin_fmin(tmax[i], b);
  r[4] = (__builtin_fabsf(u) >= m2) ? (u >= m2 ? 1.0 : 0.0) : -1.0;
  r[5] = final_hit ? 1.0 : 0.0;
}
#endif

#ifndef PT_KERNELS_ONLY /* (tools/quick_kernel.sh instantiates single kernels of this file for register / ISA studies) */
#include "bvh_build_gpu.inc"
#include "ppm.inc"
#include "ptx_api.inc"
#endif
