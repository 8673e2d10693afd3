/* pt_scene.h -- the flattened scene as it lives in HBM, shared by the host builder
 * (bvh_build.cpp) and the HIP kernels (kernels.hip).
 *
 * Layout decisions:
 *  - BVH node = one 64-byte record (bbox 6 x f64 + two u32 links), 64-byte aligned: a lane
 *    that visits a node touches exactly one cache line / four 16-byte LDS reads.
 *  - leaf primitive slots are stored in LEAF ORDER, contiguous per leaf ("packets"):
 *      sphere slot   = 4 x f64 {x, y, z, r}        (32 B; NaN padding slots for Simd_leaf)
 *      triangle slot = 10 x f64 {a, b, c, pad}     (80 B, 16-byte aligned)
 *    so a leaf is one contiguous burst, never an index gather.
 *  - a slot's shading data (material id, kind, primitive id, triangle UVs) sits in
 *    separate arrays touched only by the shade stage for the ONE slot that was hit.
 */
#ifndef PT_SCENE_H
#define PT_SCENE_H

#include <stdint.h>

#define PT_NODE_LEAF_AXIS 3u /* axis code stored in the two top bits of `b` for leaves */
#define PT_TOP_FLAG 0x20000000u /* a node reference into the LDS-resident top image (PtSceneDev.top_nodes): flag | byte offset */
#define PT_TOP_NODE_BYTES 64
#define PT_MAX_FINITE 1.7976931348623157e308

/* One BVH node, 64 bytes.  Tree.t = Bbox.t * (Leaf | Branch {axis; lhs; rhs}), shape_tree.ml:153-161 */
struct __attribute__((aligned(64))) PtNode {
  double mn[3];
  double mx[3];
  uint32_t a; /* branch: index of lhs        | leaf: first slot            */
  uint32_t b; /* branch: index of rhs + axis<<30 | leaf: slot count + 3<<30 */
  uint32_t pad[2]; /* leaf: pad[0] = number of real (non-padding) slots */
};

#define PT_SLOT_SPHERE 0
#define PT_SLOT_TRIANGLE 1
#define PT_SLOT_PAD 2 /* NaN padding slot of a Simd_leaf packet */

/* shading categories: what a segment will execute in the shade stage */
#define PT_CAT_MISS 0
#define PT_CAT_LAMBERT_SOLID 1
#define PT_CAT_LAMBERT_CHECKER 2
#define PT_CAT_METAL 3
#define PT_CAT_DIELECTRIC 4
#define PT_CAT_NONE 5 /* beyond the end of the queue */
#define PT_N_CAT 6

/* scene "mode" = which Leaf implementation the tree was built with */
#define PT_MODE_SIMD 0  /* Simd_leaf packets: every slot a sphere, Rust x86 arithmetic */
#define PT_MODE_ARRAY 1 /* Array_leaf: spheres (scalar Sphere.intersect) and/or triangles */

struct PtMaterial { /* 48 B */
  int32_t kind;
  int32_t texture;
  double index;
  double emit[3];
  double pad;
};
struct PtTexture { /* 64 B */
  int32_t kind;
  int32_t width, height;
  int32_t pad;
  double even[3];
  double odd[3];
  double pad2;
};

/* What shading one hit on a slot needs besides the slot's geometry, in ONE 96-byte record per slot: material kind /
 * index / emission and the texture it points at.  The shade stage used to reach this through three dependent gathers
 * (slot -> slot_material -> materials -> textures: each ~1-2 us under load, with the kernel 70 % of its time parked on
 * s_waitcnt); now the hop after the hit slot is the last one. */
struct __attribute__((aligned(16))) PtShadeRec {
  int32_t kind;     /* PTX_MAT_* */
  int32_t tex_kind; /* PTX_TEX_* (Lambertian / Metal) */
  int32_t tex_w, tex_h;
  double index;
  double even[3];
  double odd[3];
  double emit[3];
};

/* everything a kernel needs, passed by value (pointers are device pointers) */
struct PtSceneDev {
  const PtNode* nodes;
  int32_t n_nodes;
  int32_t depth;       /* tree depth = max traversal stack entries */
  int32_t mode;        /* PT_MODE_* */
  int32_t n_slots;
  const double* sph;   /* n_slots x 4 (valid where slot_kind != TRIANGLE) */
  const double* tri;   /* n_slots x 10 (valid where slot_kind == TRIANGLE); NULL if no triangles */
  const double* tri_uv;/* n_slots x 6 */
  /* Small scenes with triangles: what Triangle.Hit.to_hit + Shader_space.create derive from the triangle ALONE, per slot, 12 doubles:
   * {g_normal xyz, -, rotation of +g_normal (r, x, y, z), rotation of -g_normal}.  Computed once on the host with the functions
   * the shade step itself uses (pt_surface_hit), so a load replaces ~130 vector instructions per triangle hit with the same
   * bits.  NULL: the shade step computes them (large meshes: the table would be one more gathered line per segment). */
  const double* tri_frame;
  const uint8_t* slot_kind;
  const uint8_t* slot_cat;  /* shading category of the slot's material, PT_CAT_* (wave-coherent shading) */
  const int32_t* slot_material;
  const int32_t* slot_prim; /* build-list primitive index, -1 for padding */
  /* floor triangles tested before the tree (ganesha Floor): appended after the tree slots
   * in tri / tri_uv / slot_* at indices n_slots .. n_slots + n_floor - 1 */
  int32_t n_floor;
  int32_t has_triangles;
  int32_t has_emit;
  int32_t has_checker;
  /* LDS-resident scenes: 1 = the LDS image also carries every node's six binary64 bounds (48 bytes per node) for the tests the
   * binary32 filter leaves undecided -- 2 % (Shirley) to 5.5 % (cornell) of a walk's wave steps, each of which otherwise waits for a
   * global load.  Set by the host when the scene still fits LDS with them (scene_upload). */
  int32_t lds_nodes64;
  const PtMaterial* materials;
  const PtTexture* textures;
  const PtShadeRec* slot_shade; /* per slot (padding slots zero) */
  /* n_nodes x 8 (u16 node indices, 0xffff = none): per direction octant, the node visited after a node's subtree
   * (the threading of the LDS node image, kernels.hip); NULL when the tree has 65535 nodes or more */
  const uint16_t* node_skip;
  /* the same threading with 32-bit node indices (0xffffffff = none): scenes walked from HBM / L2 */
  const uint32_t* node_skip32;
  /* n_nodes x 32 bytes: mn.xyz, mx.xyz rounded to binary32, a, b (leaf b: padded count | real count << 15 | tag): the filter
   * image of the walk from HBM / L2 -- half the bytes per visit of the 64-byte binary64 node, which only undecided tests read */
  const void* nodes32;
  /* 8 x n_nodes x 32 bytes, or NULL: the same image once per direction octant with that octant's skip link inside the record
   * (mn.xyz, mx.xyz, link, skip) -- a visit is two loads instead of three (kernels.hip, PtThreadOctTag).  Built when the tree
   * is in pre-order with lhs = node + 1 (always) and every leaf fits the packed link (first slot < 2^22, <= 255 real slots). */
  const void* nodes32o;
  /* Scenes walked from HBM / L2: the TOP of the tree (the first n_top nodes in breadth-first order) as 64-byte records that
   * every trace workgroup copies into LDS -- 6 binary32 bounds, links a / b as in nodes32, eight 16-bit skip links (byte
   * offsets into this image; the node that follows a top node's subtree is an ancestor's sibling, hence a top node too), the
   * node's own index.  A link to a top node is PT_TOP_FLAG | byte offset, here and in node_skip32_top (= node_skip32 with
   * top targets encoded that way).  Half of a large mesh's node visits are within its first ~1000 nodes: those steps stop
   * waiting for L2.  NULL / 0 when the scene has no such image. */
  const uint32_t* top_nodes;
  const uint32_t* node_skip32_top;
  int32_t n_top;
  int32_t all_triangles; /* 1: every tree slot is a triangle (a mesh): the leaf loop needs no per-slot kind */
  /* unit vector (camera space) along which the primitives' centres vary least = the normal of the scene's ground plane when it
   * has one.  A HEURISTIC sort key only (shade bins survivors by the elevation of the new direction above that plane, a
   * predictor of how long the next walk is); it never enters a pixel value. */
  double sort_axis[3];
  int32_t sort_by_elevation; /* 1: the centres do lie in a slab (smallest variance < 2 % of the largest): bin by elevation; 0: by direction octant */
  /* 1: bin by whether the new ray reaches the tree's bounding box at all (+ two direction signs).  Scenes whose rays mostly start
   * OUTSIDE the tree (a mesh standing on a floor that is tested before the tree): a ray that misses the box costs one node test, one
   * that enters it walks a hundred nodes, and a wave that holds both waits for the longest.  Heuristic, binary32, never enters a pixel. */
  int32_t sort_by_root;
  float root_mn[3], root_mx[3];
  /* max |coordinate| of the tree's root box, rounded UP to binary32 (+inf when it exceeds the format): every node lies inside
   * the root box, so this bounds every node's magnitude.  The binary32 node filter applies to a ray only if
   * (root_mag + max|o|) * max|1/d| < 2^100 -- then no binary32 intermediate of the filter can overflow (kernels.hip, begin()). */
  float root_mag;
  float pad_f;
  double cam_llx, cam_lly, cam_vx, cam_vy;
  int32_t bg_kind;
  int32_t pad0;
  double bg_horizon[3];
  double bg_zenith[3];
};

#endif /* PT_SCENE_H */
