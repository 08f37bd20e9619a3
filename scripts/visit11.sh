#!/bin/bash
# GPU visit 11 (round 2): gather tile shapes at two blocks per CU: 8x16 (shipped) vs 16x16 and 8x32 (256-token tiles)
set -u
OUT=gpurun_out/r02m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python scripts/ab_bench.py 't8x16=ddp_amd/lib' 't16x16=ddp_amd/lib_g16' 't8x32=ddp_amd/lib_g832' --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
for wl in kitti_depth_k20_16x352x1216 bev_fusion_k3_8x200x200; do
timeout 200 python scripts/ab_bench.py 't8x16=ddp_amd/lib' 't16x16=ddp_amd/lib_g16' 't8x32=ddp_amd/lib_g832' --rounds 2 --reps 2 --workload $wl 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/ab_other.txt
done
DDP_LIB_PATH=$PWD/ddp_amd/lib_g16/libddp_mi355x.so timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "msda or sample_golden" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -5 | tee $OUT/pytest_g16.txt
