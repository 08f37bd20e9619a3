#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Usage (from the repo root):
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r01'
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o ddp -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/$OUT/prof_run.log 2>&1
cd $REPO
ls -R $OUT/prof | head -20
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f"
