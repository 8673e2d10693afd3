// Sanitizer driver for the HOST code of the product (CPU build only: -fsanitize=address,undefined): the PLY reader
// (host/ply.cpp mirrors a parser of untrusted binary input, ply_format/src/ply.ml:208-235,288-352), the scene builders
// (host/scenes.cpp), the PNG writer and the host BVH builder (csrc/bvh_build.cpp).  Built by `make asan` in
// path_tracer_ocaml_amd/host, run by tests/test_sanitizers.py.  Every mode exits 0 unless a sanitizer fires (they abort).
//   ply <file>            load, read every column / row the ganesha scene needs, build the ganesha scene from it
//   scene <name>          build a scene, boxes of its primitives, Shape_tree.create on the host, print the tree size
//   png <out.png>         write a small image
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../path_tracer_ocaml_amd/csrc/bvh_build.h"
#include "../../path_tracer_ocaml_amd/host/host.h"

static Box box_of_point(V3 p) {
  Box b;
  b.mn = b.mx = p;
  return b;
}

static int build_tree(const ptx_scene_desc* d) {
  std::vector<Box> boxes;
  for (int i = 0; i < d->n_triangles; ++i) {
    Box b = box_of_point(v3(d->vertex_x[d->tri_indices[3 * i]], d->vertex_y[d->tri_indices[3 * i]], d->vertex_z[d->tri_indices[3 * i]]));
    for (int k = 1; k < 3; ++k) {
      const int v = d->tri_indices[3 * i + k];
      b = box_union(b, box_of_point(v3(d->vertex_x[v], d->vertex_y[v], d->vertex_z[v])));
    }
    boxes.push_back(b);
  }
  for (int i = 0; i < d->n_spheres; ++i) {
    Box b;
    const double r = d->sphere_r[i];
    b.mn = v3(d->sphere_x[i] - r, d->sphere_y[i] - r, d->sphere_z[i] - r);
    b.mx = v3(d->sphere_x[i] + r, d->sphere_y[i] + r, d->sphere_z[i] + r);
    boxes.push_back(b);
  }
  const bool simd = d->leaf_kind == PTX_LEAF_SIMD;
  const BvhResult t = bvh_build(boxes, d->num_bins > 0 ? d->num_bins : 32, d->length_cutoff, simd);
  std::printf("prims %zu nodes %zu slots %zu depth %d leaves %d\n", boxes.size(), t.nodes.size(), t.slot_prim.size(), t.depth, t.leaves);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  if (mode == "ply") {
    pth_ply* p = pth_ply_load(argv[2]);
    if (!p) {
      std::printf("rejected: %s\n", pth_last_error());
    } else {
      double acc = 0.0;
      const int64_t nv = pth_ply_count(p, "vertex");
      for (const char* name : {"x", "y", "z", "nx", "confidence"}) {
        const double* f = pth_ply_floats(p, "vertex", name);
        const int64_t* q = pth_ply_ints(p, "vertex", name);
        for (int64_t i = 0; i < nv && (f || q); ++i) acc += f ? f[i] : (double)q[i];
      }
      const int32_t* lens = nullptr;
      const int64_t* rows = pth_ply_rows(p, "vertex_indices", &lens);
      const int64_t nr = pth_ply_count(p, "vertex_indices");
      int64_t off = 0;
      for (int64_t r = 0; rows && r < nr; ++r)
        for (int32_t k = 0; k < lens[r]; ++k) acc += (double)rows[off++];
      std::printf("loaded: vertex %lld rows %lld checksum %.17g\n", (long long)nv, (long long)nr, acc);
      pth_ply_free(p);
    }
    pth_scene* s = pth_scene_ganesha_ply(argv[2], 64, 36); /* Mesh.create + floor + camera */
    if (!s) {
      std::printf("no scene: %s\n", pth_last_error());
    } else {
      build_tree(pth_scene_desc(s));
      pth_scene_free(s);
    }
    return 0;
  }
  if (mode == "scene") {
    const std::string name = argv[2];
    pth_scene* s = name == "shirley"      ? pth_scene_shirley(96, 48, 0, 42)
                   : name == "shirley_array" ? pth_scene_shirley(96, 48, 1, 42)
                   : name == "cornell"   ? pth_scene_cornell(64, 64, 12.0)
                                          : pth_scene_ganesha_like(64, 36, 3000, 7);
    if (!s) return 1;
    build_tree(pth_scene_desc(s));
    ptx_light lights[2];
    if (name == "cornell") pth_lights_cornell(64, 64, lights);
    if (name == "ganesha") pth_lights_ganesha(s, lights);
    pth_scene_free(s);
    return 0;
  }
  if (mode == "png") {
    const int w = 37, h = 11;
    std::vector<double> img((size_t)w * h * 3);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (double)(i % 97) / 64.0 - 0.1; /* values below 0 and above 1 too */
    img[5] = NAN;
    std::vector<double> g(img.size());
    pth_ppm_gamma(img.data(), (int64_t)img.size(), 3, g.data());
    return pth_write_png(argv[2], w, h, img.data()) == 0 ? 0 : 1;
  }
  return 2;
}
