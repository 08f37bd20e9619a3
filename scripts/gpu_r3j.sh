#!/bin/bash
# round 3: early residual fetch + next-tile prefetch in the layer kernel - parity, same-box A/B, cycle stamps
set -u
TAG=${1:-r03j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "sample_golden or head_forward or c1_ade or plugin or segmentor or edge or ragged" --durations=3 2>&1 | grep -v "amdgpu.ids\|^$" > $OUT/pytest_quick.txt
tail -6 $OUT/pytest_quick.txt
python scripts/ab_bench.py main=ddp_amd/lib noqe=ddp_amd/lib_noqe --rounds 3 > $OUT/ab.txt 2>&1
grep -v amdgpu.ids $OUT/ab.txt | tail -8
python scripts/ab_bench.py main=ddp_amd/lib noqe=ddp_amd/lib_noqe --rounds 2 --workload ade_swin_t_k3_1x512x1024 --reps 20 > $OUT/ab_b1.txt 2>&1
grep -v amdgpu.ids $OUT/ab_b1.txt | tail -3
python scripts/ab_bench.py main=ddp_amd/lib noqe=ddp_amd/lib_noqe --rounds 2 --workload kitti_depth_k20_16x352x1216 --reps 2 > $OUT/ab_kitti.txt 2>&1
grep -v amdgpu.ids $OUT/ab_kitti.txt | tail -3
python scripts/stamp_layer.py lib_stamp > $OUT/layer_cycle_stamps.json 2> $OUT/stamps.err
python -c "
import json; d=json.load(open('$OUT/layer_cycle_stamps.json')); print(d['cycles_per_tile']); [print(k[:44].ljust(46), v['cycles'], v['mfma_busy_in_phase']) for k,v in d['phases'].items()]"
