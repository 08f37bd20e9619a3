#!/bin/bash
# GPU visit 13 (round 2): gather - asynchronous L2 prefetch of the next head's window (LDS-DMA dword into a sink);
# cache / LDS / VALU counters of the shipped gather (what bounds it)
set -u
OUT=gpurun_out/r02o
mkdir -p $OUT
export TMPDIR=/tmp
V="base=ddp_amd/lib pf2=ddp_amd/lib_pf2"
timeout 200 python scripts/ab_bench.py $V --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
DDP_LIB_PATH=$PWD/ddp_amd/lib_pf2/libddp_mi355x.so timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "msda or sample_golden" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -3 | tee $OUT/pytest_pf2.txt
REPO=$PWD
BENCH="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
cd /tmp
rocprofv3 -L > $REPO/$OUT/counters_avail.txt 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $REPO/$OUT/pmc_cache -o ddp -- $BENCH > $REPO/$OUT/pmc_cache.log 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $REPO/$OUT/pmc_sq -o ddp -- $BENCH > $REPO/$OUT/pmc_sq.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $REPO/$OUT/pmc_sq2 -o ddp -- $BENCH > $REPO/$OUT/pmc_sq2.log 2>&1
cd $REPO
for d in pmc_cache pmc_sq pmc_sq2; do echo "== $d"; python scripts/pmc_summary.py $(dirname $(find $OUT/$d -name '*counter_collection.csv' | head -1)) 2>&1 | grep "msda\|layer<7" | cut -c1-600; tail -2 $OUT/$d.log | cut -c1-200; done | tee $OUT/pmc_gather.txt
