#!/bin/bash
# Same-box A/B helper: build libddp_mi355x.so of one or more git refs into ddp_amd/lib_<name>/ (git-ignored, shipped by
# gpurun), next to the working tree's ddp_amd/lib/.  Then compare inside ONE gpurun call (box-to-box spread is +-3 %):
#   scripts/ab_build.sh HEAD~1 exp/gather-t8
#   gpurun --timeout 300 -- 'for v in lib lib_HEAD_1 lib_exp_gather-t8 lib lib_HEAD_1 lib_exp_gather-t8; do \
#       DDP_LIB_PATH=$PWD/ddp_amd/$v/libddp_mi355x.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline | \
#       python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"roofline\"][\"avg_launch_ms\"])"; done'
# (remove the lib_* directories afterwards; DDP_LIB_PATH is honoured by ddp_amd/_lib.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden"
for ref in "$@"; do
  name=lib_$(echo "$ref" | tr '/~^' '___')
  tmp=$(mktemp -d)
  git -C "$ROOT" archive "$ref" ddp_amd/csrc include | tar -x -C "$tmp"
  mkdir -p "$ROOT/ddp_amd/$name"
  ( cd "$tmp/ddp_amd/csrc"
    for f in ddp_api ddp_gemm ddp_gemm_bf16 ddp_kernels ddp_layer_tail; do
      /opt/rocm/bin/hipcc $FLAGS -x hip -c $f.hip -o "$tmp/$f.o" &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ddp_amd/$name/libddp_mi355x.so" "$tmp"/*.o )
  rm -rf "$tmp"
  echo "built ddp_amd/$name/libddp_mi355x.so from $ref"
done
