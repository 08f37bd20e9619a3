#!/bin/bash
# round 3, first GPU visit: the new C-ABI entries (product gather behind msda interface, aug epilogue, geometry re-prepare),
# the B = 1 strong-scaling shard, the size stream, power sampling, reference-vs-reference drift at C2 / C3.
set -u
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "from ddp_amd import build; print(build.source_hash())" > $OUT/source_sha.txt
timeout 600 python -m pytest tests -m gpu -q -x -s -k "msda or aug or size_stream or post or plugin or segmentor or smoke or sample_golden" --durations=15 2>&1 | grep -v "amdgpu.ids\|^$" > $OUT/pytest_new.txt
tail -25 $OUT/pytest_new.txt
python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
python bench.py --workload ade_swin_t_k3_1x512x1024 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_b1.json 2> $OUT/bench_b1.err
cut -c1-1200 $OUT/bench_b1.json; tail -3 $OUT/bench_b1.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-power --size-stream 50 > $OUT/bench_stream.json 2> $OUT/bench_stream.err
python -c "import json;print(json.load(open('$OUT/bench_stream.json'))['size_stream'])"; tail -3 $OUT/bench_stream.err
REPO=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_b1 -o ddp -- python $REPO/bench.py --workload ade_swin_t_k3_1x512x1024 --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --no-power > $REPO/$OUT/prof_b1.log 2>&1
cd $REPO
f=$(find $OUT/prof_b1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -14 "$f"
timeout 1500 python -m pytest tests/test_full_size_parity.py -m gpu -q -x -s -k "c2_ade or c3_" --durations=10 2>&1 | grep -v "amdgpu.ids\|^$" > $OUT/pytest_full.txt
tail -30 $OUT/pytest_full.txt
