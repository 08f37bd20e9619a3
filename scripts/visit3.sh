#!/bin/bash
# GPU visit 3 (round 2): LDS-gather tile shapes in one process, then the GPU suite (cheap files first, full size last)
set -u
OUT=gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 240 python scripts/ab_bench.py 't8=ddp_amd/lib:DDP_GATHER=t' 'l0=ddp_amd/lib:DDP_GATHER=l0' 'l1=ddp_amd/lib:DDP_GATHER=l1' \
    'l2=ddp_amd/lib:DDP_GATHER=l2' 'l3=ddp_amd/lib:DDP_GATHER=l3' 'l4=ddp_amd/lib:DDP_GATHER=l4' --rounds 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
timeout 400 python -m pytest tests/test_b3_arithmetic.py tests/test_hip_parity.py tests/test_plugin_gpu.py -m gpu -q -rf -s 2>&1 | grep -v "amdgpu.ids\|^$" | tail -70 > $OUT/pytest_gpu_fast.txt
tail -25 $OUT/pytest_gpu_fast.txt
timeout 500 python -m pytest tests/test_full_size_parity.py -m gpu -q -rf -s 2>&1 | grep -v "amdgpu.ids\|^$" | tail -60 > $OUT/pytest_gpu_full.txt
cat $OUT/pytest_gpu_full.txt
