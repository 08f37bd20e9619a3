#!/bin/bash
# Build a variant of the working tree's library with extra compiler flags into ddp_amd/lib_<name>/ (git-ignored; shipped
# by gpurun) for a same-box A/B with scripts/ab_bench.py:   scripts/variant_build.sh g16 -DDDP_GL_TH=16 -DDDP_GL_TW=16
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=lib_$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $*"
mkdir -p "$ROOT/ddp_amd/$name"
cd "$ROOT/ddp_amd/csrc"
for f in ddp_api ddp_gemm ddp_gemm_bf16 ddp_kernels ddp_layer_tail; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $f.hip -o "$ROOT/ddp_amd/$name/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ddp_amd/$name/libddp_mi355x.so" "$ROOT/ddp_amd/$name"/*.o
rm -f "$ROOT/ddp_amd/$name"/*.o
echo "built ddp_amd/$name/libddp_mi355x.so with $*"
