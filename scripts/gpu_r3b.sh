#!/bin/bash
# round 3, second GPU visit: the layer kernel with invisible (asm) loads and fp32 S fragments - parity, then a same-box A/B
# against the SB build and the build with compiler-visible loads; power sampling through amdsmi.
set -u
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "from ddp_amd import build; print(build.source_hash())" > $OUT/source_sha.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "msda or aug or size_stream or plugin or segmentor or sample_golden or head_forward or c1_ade or resources" --durations=5 2>&1 | grep -v "amdgpu.ids\|^$" > $OUT/pytest_quick.txt
tail -12 $OUT/pytest_quick.txt
python scripts/ab_bench.py main=ddp_amd/lib sb=ddp_amd/lib_sb old=ddp_amd/lib_HEAD --rounds 3 > $OUT/ab.txt 2>&1
grep -v amdgpu.ids $OUT/ab.txt | tail -12
python scripts/ab_bench.py main=ddp_amd/lib sb=ddp_amd/lib_sb old=ddp_amd/lib_HEAD --rounds 2 --workload ade_swin_t_k3_1x512x1024 --reps 20 > $OUT/ab_b1.txt 2>&1
grep -v amdgpu.ids $OUT/ab_b1.txt | tail -8
python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['power'], d['joules_per_image'])"
tail -3 $OUT/bench.err
python scripts/stamp_layer.py lib_stamp > $OUT/layer_cycle_stamps.json 2> $OUT/stamps.err
python -c "
import json; d=json.load(open('$OUT/layer_cycle_stamps.json')); print(d['cycles_per_tile']); [print(k[:44].ljust(46), v['cycles'], v['mfma_busy_in_phase']) for k,v in d['phases'].items()]"
