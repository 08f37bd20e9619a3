#!/bin/bash
# round 3: the necks on the stream-GEMM kernel (k_layer MODE 5) - parity, then timing against the round-2 neck (lib_HEAD)
set -u
TAG=${1:-r03d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "neck or fpn or msm or fcn" --durations=5 2>&1 | grep -v "amdgpu.ids\|^$" > $OUT/pytest_neck.txt
tail -8 $OUT/pytest_neck.txt
for v in lib lib_HEAD lib lib_HEAD; do
  DDP_LIB_PATH=$PWD/ddp_amd/$v/libddp_mi355x.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-power --next-rows 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); n=d['next_rows']; print('$v', d['value'], n['neck_fpn_ms'], n['neck_multi_stage_merging_ms'], n['post_epilogue_ms'], n['end_to_end_images_per_s'])"
done | tee $OUT/neck_ab.txt
REPO=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_next -o ddp -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-power --next-rows > $REPO/$OUT/prof_next.log 2>&1
cd $REPO
f=$(find $OUT/prof_next -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | grep -v "Cijk\|at::native" | head -24 | cut -c1-200
