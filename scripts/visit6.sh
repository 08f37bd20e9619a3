#!/bin/bash
# GPU visit 6 (round 2): MODE 3 (depth step head, layer-0 projections for bev / head_forward): parity, then timing of C4 / C5
set -u
OUT=gpurun_out/r02f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_plugin_gpu.py -m gpu -q -rf 2>&1 | grep -v "amdgpu.ids\|^$" | tail -15 | tee $OUT/pytest_fast.txt
for wl in kitti_depth_k20_16x352x1216 bev_fusion_k3_8x200x200; do
  timeout 200 python scripts/ab_bench.py "main=ddp_amd/lib" "prev=ddp_amd/lib_HEAD_1" --workload $wl --rounds 2 --reps 2 2>&1 | grep -v amdgpu.ids | sed "s/^/$wl /" | tee -a $OUT/other_workloads.txt
done
timeout 300 python -m pytest tests/test_full_size_parity.py -m gpu -q -rf -s -k "c4 or c5" 2>&1 | grep -v "amdgpu.ids\|^$" | grep "C[0-9]\|passed\|failed\|Error\|assert" | tee $OUT/pytest_full.txt
