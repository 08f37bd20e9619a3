#!/bin/bash
# GPU visit 12 (round 2): gather - L2 prefetch of the next head's window; 8x8 tiles with 256 threads (more blocks per CU)
set -u
OUT=gpurun_out/r02n
mkdir -p $OUT
export TMPDIR=/tmp
V="base=ddp_amd/lib pf=ddp_amd/lib_pf s8=ddp_amd/lib_s8 s8pf=ddp_amd/lib_s8pf"
timeout 200 python scripts/ab_bench.py $V --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
for wl in kitti_depth_k20_16x352x1216 bev_fusion_k3_8x200x200; do
timeout 200 python scripts/ab_bench.py $V --rounds 2 --reps 2 --workload $wl 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/ab_other.txt
done
for v in pf s8pf; do
DDP_LIB_PATH=$PWD/ddp_amd/lib_$v/libddp_mi355x.so timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "msda or sample_golden" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -3 | tee $OUT/pytest_$v.txt
done
