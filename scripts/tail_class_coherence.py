#!/usr/bin/env python
"""How much of the fused tail's time is the gather of T = LUT . W_m^T rows by the argmax class?  Every token reads the 1-KiB row
of ITS class; with the synthetic weights of the benchmark the classes are spatially incoherent (32 different rows per load
instruction), a trained segmentor's are piecewise constant.  Teacher forcing (DDP_FLAG_FORCE_X0) lets the decisions be chosen:
the engine's own (incoherent), uniform random, and one class everywhere (every lane reads the same row) - same arithmetic, same
bytes everywhere else.  Prints the average time of the layer + tail kernel (call site 10) for each.

  gpurun -- 'python scripts/tail_class_coherence.py > gpurun_out/<tag>/tail_class_coherence.json'"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddp_amd import _lib  # noqa: E402
from ddp_amd.engine import DDPEngine, PackedWeights  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS['ade_swin_t_k3_8x512x1024']
    sd = synthetic.make_state_dict('seg', 150, 6, 256, seed=2)
    weights = PackedWeights(sd, 'seg', 6, dev)
    x, noise = synthetic.make_inputs(wl['batch'], wl['h'], wl['w'], 1, 256, 256, seed=0)
    dx, dn = x.to(dev), noise.to(dev)
    kw = dict(h=wl['h'], w=wl['w'], batch=wl['batch'], randsteps=1, timesteps=3, num_classes=150, bit_scale=0.01, accumulation=True,
              device=dev, weights=weights)
    rec = DDPEngine(sd, 'seg', record_x0=True, **kw)
    rec.sample(dx, dn)
    own = rec.x0_trace()
    del rec
    eng = DDPEngine(sd, 'seg', force_x0=True, **kw)
    lib = eng.lib
    g = torch.Generator(device='cpu').manual_seed(3)
    # piecewise constant: 16 x 16-token blocks of one class each (what a trained model's class map looks like at this scale)
    blocks = torch.randint(0, 150, (own.shape[0], own.shape[1], wl['h'] // 16, wl['w'] // 16), generator=g, dtype=torch.uint8)
    cases = {'own decisions (synthetic weights: spatially incoherent)': own,
             'uniform random classes': torch.randint(0, 150, own.shape, generator=g, dtype=torch.uint8),
             'piecewise constant (16 x 16-token blocks)': blocks.repeat_interleave(16, 2).repeat_interleave(16, 3),
             'one class everywhere': torch.zeros_like(own)}
    out = {}
    for name, dec in cases.items():
        eng.set_x0_decisions(dec)
        eng.sample(dx, dn)
        torch.cuda.synchronize()
        _lib.check(lib.ddp_profile_begin(255))
        for _ in range(3):
            eng.sample(dx, dn)
        tot, n = C.c_float(0), C.c_int(0)
        _lib.check(lib.ddp_profile_end(C.byref(tot), C.byref(n)))
        t, k = C.c_float(0), C.c_int(0)
        lib.ddp_profile_read(10, C.byref(t), C.byref(k))
        out[name] = {'layer_tail_ms': round(t.value / max(k.value, 1), 4), 'launches': k.value,
                     'distinct_classes_per_32_tokens': round(float(torch.tensor([len(set(r.tolist())) for r in dec[0, 0].reshape(-1, 32)[:2048]],
                                                                                dtype=torch.float32).mean()), 1)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
