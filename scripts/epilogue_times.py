#!/usr/bin/env python
"""Times of the four fused post-loop epilogues at the sizes of the BASELINE configurations / the reference's test
protocols, HIP events around 50 back-to-back launches each, with the bytes each one HAS to move (low-resolution inputs read once,
the result written once) and the rate that makes of them.  Prints one JSON object.
  gpurun -- 'python scripts/epilogue_times.py > gpurun_out/<tag>/epilogue_times.json'"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddp_amd.engine import depth_postprocess, seg_aug_postprocess, seg_postprocess, seg_slide_postprocess, slide_windows  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def entry(name, ms, rd, wr, note):
    return {'case': name, 'ms': round(ms, 4), 'bytes_read': int(rd), 'bytes_written': int(wr),
            'tb_per_s_on_those_bytes': round((rd + wr) / ms / 1e9, 3), 'note': note}


def main():
    res = []
    # C2: 8 x 150 x 128 x 256 scores -> 8 x 512 x 1024 class map
    s = synthetic.make_scores(8, 150, 128, 256, 1).cuda()
    ms = timed(lambda: seg_postprocess(s, (512, 1024)))
    res.append(entry('ddp_seg_postprocess C2 (8x150x128x256 -> 8x512x1024 uint8)', ms, s.numel() * 4, 8 * 512 * 1024,
                     'k_seg_postprocess_x4; the reference materialises 2 x 2.5 GB of fp32 scores here'))
    # Cityscapes: 4 x 19 x 256 x 512 -> 4 x 1024 x 2048
    s = synthetic.make_scores(4, 19, 256, 512, 2).cuda()
    ms = timed(lambda: seg_postprocess(s, (1024, 2048)))
    res.append(entry('ddp_seg_postprocess C3 shard (4x19x256x512 -> 4x1024x2048 uint8)', ms, s.numel() * 4, 4 * 1024 * 2048, ''))
    # ADE multi-scale + flip: 6 augmentations of a 512 x 683 image, 150 classes
    ori = (512, 683)
    augs = []
    for i, sc in enumerate((0.5, 1.0, 1.5)):
        H, W = int(ori[0] * sc + 0.5), int(ori[1] * sc + 0.5)
        Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
        for flip in (None, 'horizontal'):
            augs.append((synthetic.make_scores(1, 150, Hp // 4, Wp // 4, 70 + i).cuda(), dict(img_size=(Hp, Wp), crop_size=(H, W), flip=flip)))
    sc_l, metas = [a[0] for a in augs], [a[1] for a in augs]
    ms = timed(lambda: seg_aug_postprocess(sc_l, metas, ori))
    res.append(entry('ddp_seg_aug_postprocess ADE (6 augmentations, 150 classes -> 512x683 uint8)', ms, sum(t.numel() for t in sc_l) * 4,
                     ori[0] * ori[1], 'per pixel: 6 x 150 two-stage interpolations + softmax (compute on L2-resident inputs)'))
    # Cityscapes sliding window: 1024 x 2048, crop 512 x 1024, stride 341 x 683 -> 3 x 3 windows, 19 classes, b = 1
    ys, xs, crop = slide_windows((1024, 2048), (512, 1024), (341, 683))
    sw = torch.stack([synthetic.make_scores(1, 19, 128, 256, 900 + i) for i in range(9)]).cuda()
    ms = timed(lambda: seg_slide_postprocess(sw, ys, xs, crop, (1024, 2048)))
    res.append(entry('ddp_seg_slide_postprocess Cityscapes (3x3 windows of 512x1024, 19 classes -> 1024x2048 uint8)', ms, sw.numel() * 4,
                     1024 * 2048, 'per pixel: 19 classes x <= 4 covering windows'))
    # KITTI: 16 x 88 x 304 x 2 augmentations -> 16 x 352 x 1216 fp32
    a, b = synthetic.make_depth_map(16, 88, 304, 50).cuda(), synthetic.make_depth_map(16, 88, 304, 51).cuda()
    out = torch.empty((16, 1, 352, 1216), device='cuda')
    ms = timed(lambda: depth_postprocess([a, b], [None, 'horizontal'], (352, 1216), 1e-3, 80.0, out=out))
    res.append(entry('ddp_depth_postprocess C4 (16x88x304 x 2 augmentations -> 16x352x1216 fp32)', ms, (a.numel() + b.numel()) * 4, out.numel() * 4,
                     'k_depth_aug_postprocess: one 16-B store per thread'))
    ms = timed(lambda: depth_postprocess([a[:1], b[:1]], [None, 'horizontal'], (352, 1216), 1e-3, 80.0, out=out[:1]))
    res.append(entry('ddp_depth_postprocess KITTI harness call (1 image x 2 augmentations -> 352x1216 fp32)', ms, 2 * 88 * 304 * 4, 352 * 1216 * 4,
                     'launch-latency bound: 1.7 MB'))
    print(json.dumps({'epilogues': res, 'hbm_peak_tb_per_s': 8.0}, indent=1))


if __name__ == '__main__':
    main()
