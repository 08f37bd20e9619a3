#!/bin/bash
# one-off visit: gather latency-chain variants (A/B) + power calibration
set -u
OUT=gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
python -c "from ddp_amd import build; print(build.source_hash())" > $OUT/source_sha.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "msda or sample_golden or c2_ade or c2_size_trained" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -5 > $OUT/pytest_subset.txt
cat $OUT/pytest_subset.txt
timeout 400 python scripts/ab_bench.py main=ddp_amd/lib glv0=ddp_amd/lib_glv0 kp=ddp_amd/lib_kp --rounds 3 2>&1 | grep -v amdgpu.ids > $OUT/ab_gather_chain.txt
tail -12 $OUT/ab_gather_chain.txt
timeout 400 python scripts/ab_bench.py main=ddp_amd/lib glv0=ddp_amd/lib_glv0 kp=ddp_amd/lib_kp --rounds 2 --workload city_swin_l_k10_4x1024x2048 --reps 2 2>&1 | grep -v amdgpu.ids > $OUT/ab_gather_chain_city.txt
tail -8 $OUT/ab_gather_chain_city.txt
timeout 400 python scripts/power_calibration.py --sweep > $OUT/power_calibration.json 2> $OUT/power_calibration.err
cat $OUT/power_calibration.json; tail -5 $OUT/power_calibration.err
exit 0
