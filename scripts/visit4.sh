#!/bin/bash
# GPU visit 4 (round 2): gather variants (swizzle / split-first, double-buffered), then the tests that failed for test bugs
set -u
OUT=gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 240 python scripts/ab_bench.py 'l6_old=ddp_amd/lib:DDP_GATHER=l6' 'l1_var1=ddp_amd/lib:DDP_GATHER=l1' 'l0_var1=ddp_amd/lib:DDP_GATHER=l0' \
    'l4_var1=ddp_amd/lib:DDP_GATHER=l4' 'l7_db=ddp_amd/lib:DDP_GATHER=l7' --rounds 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
for g in l1 l7; do
DDP_GATHER=$g timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sample or msda or self_aligned or nan" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -4 | tee $OUT/pytest_$g.txt
done
timeout 400 python -m pytest tests/test_full_size_parity.py -m gpu -q -rf -s -k "c2 or c3" 2>&1 | grep -v "amdgpu.ids\|^$" | grep "^C[0-9]\|passed\|failed\|Error\|assert" | tee $OUT/pytest_gpu_full.txt
