#!/bin/bash
# GPU visit 2 (round 2): one-process A/B of the gather variants + the hidden-load layer kernel, then the whole GPU suite
set -u
OUT=gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/ab_bench.py 't8=ddp_amd/lib:DDP_GATHER=t' 'lds=ddp_amd/lib:DDP_GATHER=l' 'w8=ddp_amd/lib:DDP_GATHER=w' \
    'asm_t8=ddp_amd/lib_exp_asm-loads:DDP_GATHER=t' --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
timeout 900 python -m pytest tests -m gpu -q -rf -s 2>&1 | grep -v "amdgpu.ids\|^$" | tail -150 > $OUT/pytest_gpu_full.txt
tail -60 $OUT/pytest_gpu_full.txt
