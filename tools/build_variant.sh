#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"   -> build_variants/libptx_<name>.so (not tracked; travels with gpurun)
set -e
cd "$(dirname "$0")/../path_tracer_ocaml_amd/csrc"
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-math-errno -Wall -Wno-unused-function \
  -Xarch_host -march=x86-64-v3 -shared -o ../../build_variants/libptx_$1.so kernels.hip -x hip bvh_build.cpp $2
