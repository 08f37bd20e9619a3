#!/bin/bash
# GPU visit 10 (round 2): q as fp32 fragments between the kernels - timing vs the previous state, parity, cycle stamps
set -u
OUT=gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python scripts/ab_bench.py 'prev=ddp_amd/lib_HEAD' 'qf32=ddp_amd/lib' --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_plugin_gpu.py -m gpu -q -rf 2>&1 | grep -v "amdgpu.ids\|^$" | tail -12 | tee $OUT/pytest_fast.txt
timeout 100 python scripts/stamp_layer.py lib_stamp 2>&1 | grep -v amdgpu.ids > $OUT/stamps.json; grep -A3 "cycles_per_tile\|P0\|P1 \|LN1" $OUT/stamps.json | grep "cycles\|P0\|P1\|LN1" | paste - - | cut -c1-160
