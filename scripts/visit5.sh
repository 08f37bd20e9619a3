#!/bin/bash
# GPU visit 5 (round 2): the u-chain (fused tail + next step head) against the previous state, parity of the new path,
# and the kernel split of the other BASELINE configurations
set -u
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python scripts/ab_bench.py 'prev=ddp_amd/lib_HEAD_1' 'uchain=ddp_amd/lib' --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sample or nan or aligned" -s 2>&1 | grep -v "amdgpu.ids\|^$" | tail -12 | tee $OUT/pytest_fast.txt
timeout 300 python -m pytest tests/test_full_size_parity.py -m gpu -q -rf -s -k "c2 or c1" 2>&1 | grep -v "amdgpu.ids\|^$" | grep "C[0-9]\|passed\|failed\|Error\|assert" | tee $OUT/pytest_full.txt
for wl in city_swin_l_k10_4x1024x2048 kitti_depth_k20_16x352x1216 bev_fusion_k3_8x200x200; do
  timeout 200 python scripts/ab_bench.py "main=ddp_amd/lib" --workload $wl --rounds 1 --reps 2 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/$wl /" | tee -a $OUT/other_workloads.txt
done
