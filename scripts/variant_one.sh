#!/bin/bash
# Variant of ONE translation unit (the others are taken as the objects of the working tree's last build, ddp_amd/lib/*.o) into
# ddp_amd/lib_<name>/ for a same-box A/B with scripts/ab_bench.py:   scripts/variant_one.sh h5 ddp_kernels -DDDP_GL_HALO=5
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=lib_$1; tu=$2; shift; shift
mkdir -p "$ROOT/ddp_amd/$name"
cd "$ROOT/ddp_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -x hip -c $tu.hip -o "$ROOT/ddp_amd/$name/$tu.o"
objs="$ROOT/ddp_amd/$name/$tu.o"
for f in ddp_api ddp_gemm ddp_gemm_bf16 ddp_kernels ddp_layer_tail; do
  [ "$f" = "$tu" ] || objs="$objs $ROOT/ddp_amd/lib/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$ROOT/ddp_amd/csrc/exports.map" -o "$ROOT/ddp_amd/$name/libddp_mi355x.so" $objs
rm -f "$ROOT/ddp_amd/$name/$tu.o"
echo "built ddp_amd/$name/libddp_mi355x.so ($tu with $*)"
