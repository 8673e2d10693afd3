/* bvh_build.cpp -- see bvh_build.h.  Works on an index permutation instead of the reference's
 * Slice of boxed records; every arithmetic expression and every tie rule follows shape_tree.ml. */
#include "bvh_build.h"

#include <algorithm>
#include <cstring>

namespace {

struct Split {
  bool valid = false;
  double cost = 0.0;
  int axis = 0;
  int index = 0;    /* split_index p: element goes left iff bin <= p */
  double scale = 0.0;
  double origin = 0.0; /* cb_min on the axis */
  Box lhs, rhs;
};

/* OCaml's Float.compare: nan = nan, nan < everything else (Proposal.compare, shape_tree.ml:121) */
inline int ocaml_compare(double a, double b) {
  if (a != a) return (b != b) ? 0 : -1;
  if (b != b) return 1;
  return a < b ? -1 : (a > b ? 1 : 0);
}

class Builder {
 public:
  Builder(const std::vector<Box>& boxes, int num_bins, int cutoff, bool pad4)
      : boxes_(boxes), bins_(num_bins), cutoff_(cutoff), pad4_(pad4) {
    const int n = (int)boxes.size();
    order_.resize(n);
    centroid_.resize(n);
    for (int i = 0; i < n; ++i) {
      order_[i] = i;
      centroid_[i] = box_center(boxes[i]); /* Bshape.create, shape_tree.ml:15-19 */
    }
    bin_box_.resize(bins_);
    bin_has_.resize(bins_);
    bin_count_.resize(bins_);
    left_box_.resize(bins_);
    left_has_.resize(bins_);
    right_box_.resize(bins_);
    right_has_.resize(bins_);
  }

  BvhResult run() {
    const int n = (int)order_.size();
    Box root = boxes_[0];
    for (int i = 1; i < n; ++i) root = box_union(root, boxes_[i]); /* shape_tree.ml:257-260 */
    int depth = 0;
    build(root, 0, n, &depth);
    out_.depth = depth;
    return std::move(out_);
  }

 private:
  int bin_of(const Split& s, int elt) const {
    /* to_bin b = Float.to_int (scale * (centroid - cb_min)), shape_tree.ml:133 */
    return (int)(s.scale * (v3_axis(centroid_[elt], s.axis) - s.origin));
  }

  /* Proposal.propose_split_one_axis, shape_tree.ml:123-139 */
  Split propose_axis(int lo, int hi, int axis, const Box& cbox) {
    Split best;
    const double epsilon = 1e-6;
    const double cb_min = v3_axis(cbox.mn, axis), cb_max = v3_axis(cbox.mx, axis);
    const double scale = (double)bins_ * (1.0 - epsilon) / (cb_max - cb_min);
    if (!pt_isfinite(scale)) return best;
    std::fill(bin_has_.begin(), bin_has_.end(), (char)0);
    std::fill(bin_count_.begin(), bin_count_.end(), 0);
    Split probe;
    probe.scale = scale;
    probe.origin = cb_min;
    probe.axis = axis;
    for (int k = lo; k < hi; ++k) { /* Bin.insert, :41-51, in slice order */
      const int e = order_[k];
      const int b = bin_of(probe, e);
      if (bin_has_[b]) bin_box_[b] = box_union(bin_box_[b], boxes_[e]);
      else {
        bin_box_[b] = boxes_[e];
        bin_has_[b] = 1;
      }
      bin_count_[b]++;
    }
    /* Bin.populate_bbox_r (:53-60) and populate_bbox_l (:62-69): union_opt (bbox bin_j) neighbour */
    for (int j = bins_ - 1; j >= 0; --j) {
      const bool nb = (j + 1 < bins_) && right_has_[j + 1];
      if (bin_has_[j] && nb) right_box_[j] = box_union(bin_box_[j], right_box_[j + 1]);
      else if (bin_has_[j]) right_box_[j] = bin_box_[j];
      else if (nb) right_box_[j] = right_box_[j + 1];
      right_has_[j] = (bin_has_[j] || nb) ? 1 : 0;
    }
    for (int j = 0; j < bins_; ++j) {
      const bool nb = (j > 0) && left_has_[j - 1];
      if (bin_has_[j] && nb) left_box_[j] = box_union(bin_box_[j], left_box_[j - 1]);
      else if (bin_has_[j]) left_box_[j] = bin_box_[j];
      else if (nb) left_box_[j] = left_box_[j - 1];
      left_has_[j] = (bin_has_[j] || nb) ? 1 : 0;
    }
    /* candidates (:91-119): total bbox = bbox_l of the last bin */
    const double total_area = box_surface_area(left_box_[bins_ - 1]);
    const int total = hi - lo;
    /* The reference conses candidates for p = 0.. and takes List.min_elt (first minimum of the
     * consed list = HIGHEST p among equal costs).  Walk p upward and replace on <=. */
    int n_left = 0;
    for (int p = 0; p < bins_ - 1; ++p) {
      n_left += bin_count_[p];
      if (!left_has_[p] || !right_has_[p + 1]) continue;
      const int n_right = total - n_left;
      const double lhs_area = (double)n_left * box_surface_area(left_box_[p]);
      const double rhs_area = (double)n_right * box_surface_area(right_box_[p + 1]);
      const double cost = 0.25 + ((lhs_area + rhs_area) * 1.0 / total_area); /* costT + (..)*costI/total */
      if (!best.valid || ocaml_compare(cost, best.cost) <= 0) {
        best.valid = true;
        best.cost = cost;
        best.axis = axis;
        best.index = p;
        best.scale = scale;
        best.origin = cb_min;
        best.lhs = left_box_[p];
        best.rhs = right_box_[p + 1];
      }
    }
    return best;
  }

  /* Proposal.create (:141-146): X, Y, Z; the first minimum wins */
  Split propose(int lo, int hi) {
    Box cbox;
    cbox.mn = cbox.mx = centroid_[order_[lo]]; /* Bshape.centroid_bbox, :21-24 */
    for (int k = lo + 1; k < hi; ++k) {
      Box c;
      c.mn = c.mx = centroid_[order_[k]];
      cbox = box_union(cbox, c);
    }
    Split best;
    for (int axis = 0; axis < 3; ++axis) {
      Split s = propose_axis(lo, hi, axis, cbox);
      if (!s.valid) continue;
      if (!best.valid || ocaml_compare(s.cost, best.cost) < 0) best = s;
    }
    return best;
  }

  /* Slice.partition_in_place (slice.ml:67-80) on order_[lo, hi) */
  int partition(int lo, int hi, const Split& s) {
    int i = 0, j = (hi - lo) - 1;
    auto left = [&](int k) { return bin_of(s, order_[lo + k]) <= s.index; };
    while (i < j) {
      while (left(i) && i < j) ++i;
      while (j >= 0 && !left(j)) --j;
      if (i < j) std::swap(order_[lo + i], order_[lo + j]);
    }
    return i;
  }

  int make_leaf(const Box& bbox, int lo, int hi) {
    const int n = hi - lo;
    PtNode node;
    std::memset(&node, 0, sizeof node);
    store_box(node, bbox);
    int len = n;
    if (pad4_) len = (n + 3) & ~3;
    node.a = (uint32_t)out_.slot_prim.size();
    node.b = (uint32_t)len | (PT_NODE_LEAF_AXIS << 30);
    node.pad[0] = (uint32_t)n; /* real elements: the NaN padding slots can never be selected */
    for (int k = 0; k < len; ++k) out_.slot_prim.push_back(k < n ? order_[lo + k] : -1);
    out_.nodes.push_back(node);
    out_.leaves++;
    return (int)out_.nodes.size() - 1;
  }

  static void store_box(PtNode& node, const Box& b) {
    node.mn[0] = b.mn.x; node.mn[1] = b.mn.y; node.mn[2] = b.mn.z;
    node.mx[0] = b.mx.x; node.mx[1] = b.mx.y; node.mx[2] = b.mx.z;
  }

  /* Tree.create loop (:177-196) */
  int build(const Box& bbox, int lo, int hi, int* depth_out) {
    const int n = hi - lo;
    Split s = propose(lo, hi);
    const double leaf_cost = 1.0 * (double)n; /* Proposal.leaf_cost :84 */
    if (!s.valid || (s.cost >= leaf_cost && n <= cutoff_) || n <= 4) {
      *depth_out = 0;
      return make_leaf(bbox, lo, hi);
    }
    const int i = partition(lo, hi, s);
    const int me = (int)out_.nodes.size();
    out_.nodes.emplace_back();
    int dl = 0, dr = 0;
    const int lhs = build(s.lhs, lo, lo + i, &dl);
    const int rhs = build(s.rhs, lo + i, hi, &dr);
    PtNode& node = out_.nodes[me];
    std::memset(&node, 0, sizeof node);
    store_box(node, bbox);
    node.a = (uint32_t)lhs;
    node.b = (uint32_t)rhs | ((uint32_t)s.axis << 30);
    *depth_out = 1 + std::max(dl, dr);
    return me;
  }

  const std::vector<Box>& boxes_;
  int bins_, cutoff_;
  bool pad4_;
  std::vector<int> order_;
  std::vector<V3> centroid_;
  std::vector<Box> bin_box_, left_box_, right_box_;
  std::vector<char> bin_has_, left_has_, right_has_;
  std::vector<int> bin_count_;
  BvhResult out_;
};

}  // namespace

BvhResult bvh_build(const std::vector<Box>& boxes, int num_bins, int length_cutoff, bool pad4) {
  if (boxes.empty()) return BvhResult();
  Builder b(boxes, num_bins, length_cutoff, pad4);
  return b.run();
}
