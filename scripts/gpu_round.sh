#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, PMC passes (HBM traffic, MFMA busy).
# Usage (from the repo root):   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r01'
# Copy what should be judged from gpurun_out/<tag>/ into profiles/ afterwards (scripts/collect_profiles.py).
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "from ddp_amd import build; print(build.source_hash())" > $OUT/source_sha.txt
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 2 --next-rows > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
REPO=$PWD
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o ddp -- $BENCH > $REPO/$OUT/prof_run.log 2>&1
# the rows either side of the loop (FPN, MultiStageMerging, post-loop epilogue) in their own kernel-stats pass
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_next -o ddp -- $BENCH --next-rows > $REPO/$OUT/prof_next_run.log 2>&1
# counters in their own passes (kernel-trace only, no other trace domains); FETCH_SIZE and WRITE_SIZE do not fit
# one pass ("exceeds the capabilities of the hardware" - rocprofv3 then hangs, hence the timeouts)
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_hbm_rd -o ddp -- $BENCH > $REPO/$OUT/pmc_hbm_rd.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_hbm_wr -o ddp -- $BENCH > $REPO/$OUT/pmc_hbm_wr.log 2>&1
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $REPO/$OUT/pmc_mfma -o ddp -- $BENCH > $REPO/$OUT/pmc_mfma.log 2>&1
cd $REPO
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -30 "$f"
python scripts/collect_profiles.py $TAG --print-only
