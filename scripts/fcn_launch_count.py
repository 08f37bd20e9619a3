#!/usr/bin/env python
"""Six calls of the K-step sampler with ``FCNHeadWithTime`` as decode head through ONE engine, for a rocprofv3 kernel trace
(scripts/gpu_round.sh): with ``ddp_prepare_fcn`` the loop-invariant kernels (k_fcn_fold, k_pack_conv3x3_scaled, k_split_weights,
k_build_stages, k_matvec, k_build_lut, k_pack_cols, ...) must appear K x num_convs (+ a few) times in TOTAL - once per engine -
not once per call.  C2-sized batch (8 x 128 x 256 tokens, 150 classes, 2 convs); prints the wall time of every call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddp_amd  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402

K, ncls, nconv = 3, 150, 2
model = ddp_amd.build_segmentor(dict(
    type='DDP', timesteps=K, bit_scale=0.01, accumulation=True,
    decode_head=dict(type='FCNHeadWithTime', num_convs=nconv, concat_input=False, in_channels=256, channels=256, num_classes=ncls,
                     in_index=0, norm_cfg=dict(type='BN'))))
model.load_state_dict(synthetic.make_fcn_segmentor_state_dict(nconv, ncls, True, False, 140), strict=True)
model = model.cuda().eval()
x, n = synthetic.make_inputs(8, 128, 256, 1, 256, 256, seed=1)
x, n = x.cuda(), n.cuda()
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.ddim_sample(x, noise=n)
    torch.cuda.synchronize()
    print('call', i, round((time.perf_counter() - t0) * 1e3, 2), 'ms', flush=True)
