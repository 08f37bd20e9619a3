#!/bin/bash
# GPU visit 14 (round 2): gather with one head per block (head = XCD), head-major sample table from the layer kernel
set -u
OUT=gpurun_out/r02p
mkdir -p $OUT
export TMPDIR=/tmp
V="base=ddp_amd/lib_HEAD hm=ddp_amd/lib"
timeout 200 python scripts/ab_bench.py $V --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
for wl in city_swin_l_k10_4x1024x2048 kitti_depth_k20_16x352x1216 bev_fusion_k3_8x200x200; do
timeout 200 python scripts/ab_bench.py $V --rounds 2 --reps 2 --workload $wl 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $OUT/ab_other.txt
done
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_plugin_gpu.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids\|^$" | tail -5 | tee $OUT/pytest.txt
REPO=$PWD
BENCH="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
cd /tmp
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $REPO/$OUT/pmc_cache -o ddp -- $BENCH > $REPO/$OUT/pmc_cache.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_rd -o ddp -- $BENCH > $REPO/$OUT/pmc_rd.log 2>&1
cd $REPO
for d in pmc_cache pmc_rd; do echo "== $d"; python scripts/pmc_summary.py $(dirname $(find $OUT/$d -name '*counter_collection.csv' | head -1)) 2>&1 | grep "msda" | cut -c1-400; done | tee $OUT/pmc_gather.txt
