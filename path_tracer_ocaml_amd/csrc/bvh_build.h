/* bvh_build.h -- host-side binned-SAH BVH builder producing the flattened HBM layout.
 * Behaviourally identical to Shape_tree.Make(Leaf).create (path_tracer/src/shape_tree.ml:72-196,252-263):
 * same splits, same leaf decisions, same element order inside every leaf (tie-breaking depends on it). */
#ifndef BVH_BUILD_H
#define BVH_BUILD_H

#include <vector>

#include "pt_scene.h"
#include "pt_vec.h"

struct BvhResult {
  std::vector<PtNode> nodes;      /* pre-order: node, lhs subtree, rhs subtree */
  std::vector<int32_t> slot_prim; /* per leaf slot: index into the input boxes, -1 = padding */
  int depth = 0;                  /* Shape_tree.depth (leaf = 0) */
  int leaves = 0;
};

/* boxes: Leaf.elt_bbox of every element in build-list order.
 * num_bins: ?num_bins (32).  length_cutoff: Leaf.length_cutoff.
 * pad4: Simd_leaf.of_elts pads every leaf to a multiple of 4 slots (main.ml:177-186). */
BvhResult bvh_build(const std::vector<Box>& boxes, int num_bins, int length_cutoff, bool pad4);

#endif
