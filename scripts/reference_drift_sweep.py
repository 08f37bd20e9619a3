#!/usr/bin/env python
"""CPU only, no GPU, no ddp_amd: how far does the REFERENCE (the pinned oracle) drift from its own restatements when the sampler
runs freely, image by image?  For each image of a C3-size problem (4 x 256 x 512 tokens, 19 classes, 10 DDIM steps) the oracle
runs as pinned (fp32, grid_sample core), with the explicit-tap core (the arithmetic of mmcv's compiled kernel) and in fp64;
prints max-rel, pixels above 1e-4 and differing decisions of each pair (oracle.reference_drift_seg).  The yardstick of
tests/test_full_size_parity.py is one such draw; this is its distribution.

  python scripts/reference_drift_sweep.py --weights-seed 3 --inputs-seed 30 > profiles/<tag>_reference_drift_c3.txt"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddp_amd.utils import synthetic  # noqa: E402  (pure torch-CPU data generators)
from oracle import ddp_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--weights-seed', type=int, default=3)
    ap.add_argument('--inputs-seed', type=int, default=30)
    ap.add_argument('--images', default='0,1,2,3')
    ap.add_argument('--variants', default='taps,fp64')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--hw', default='256,512')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--classes', type=int, default=19)
    ap.add_argument('--accumulation', action='store_true')
    args = ap.parse_args()
    h, w = [int(v) for v in args.hw.split(',')]
    sd = synthetic.make_state_dict('seg', args.classes, 6, 256, seed=args.weights_seed)
    x, noise = synthetic.make_inputs(args.batch, h, w, 1, 256, 256, seed=args.inputs_seed)
    variants = tuple(v for v in args.variants.split(',') if v)
    print(f'# weights seed {args.weights_seed}, inputs seed {args.inputs_seed}, {args.batch} x {h} x {w} tokens, {args.classes} classes, '
          f'{args.steps} steps, accumulation {args.accumulation}, torch threads {torch.get_num_threads()}', flush=True)
    for b in [int(i) for i in args.images.split(',') if i]:
        t0 = time.perf_counter()
        dr = O.reference_drift_seg(x[b:b + 1], noise[b], sd, timesteps=args.steps, accumulation=args.accumulation, bit_scale=0.01,
                                   variants=variants)
        print(f'image {b}: reference-vs-reference {dr["ref_vs_ref"]:.3e}  ' +
              ', '.join(f'[{v}: {d["max_rel"]:.3e}, {d["pixels_above_1e-4"]} px above 1e-4, {d["decisions_differ"]} decisions differ]'
                        for v, d in dr['variants'].items()) + f'  ({time.perf_counter() - t0:.0f} s)', flush=True)


if __name__ == '__main__':
    main()
