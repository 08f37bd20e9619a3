#!/bin/bash
# round 3: FCNHeadWithTime on the stream GEMM - parity + timing of the head at C2 size (2 convs, 150 classes)
set -u
TAG=${1:-r03k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -s -k "fcn or neck or fpn or msm or msda or resources" --durations=3 2>&1 | grep -v "amdgpu.ids\|^$" > $OUT/pytest_fcn.txt
tail -12 $OUT/pytest_fcn.txt
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee $OUT/fcn_time.txt
import time, torch, ddp_amd
from ddp_amd.utils import synthetic
torch.manual_seed(0)
head = ddp_amd.FCNHeadWithTime(num_convs=2, kernel_size=3, concat_input=False, dilation=1, in_channels=256, channels=256, num_classes=150,
                               in_index=0, norm_cfg=dict(type='BN')).cuda().eval()
x = torch.randn(8, 256, 128, 256, device='cuda'); t = torch.randn(1, 1024, device='cuda').expand(8, 1024).contiguous()
for _ in range(2): head([x], t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): head([x], t)
torch.cuda.synchronize(); print('FCNHeadWithTime 2 convs 8x256x128x256 -> 150 classes: %.3f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
P
