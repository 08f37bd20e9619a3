#!/usr/bin/env python
"""Same-box A/B of code states in ONE process (one `import torch`, one set of inputs): several builds of the library
(scripts/ab_build.sh -> ddp_amd/lib_<name>/) and / or per-call environment switches, each timed on the headline workload
with per-call-site HIP-event times (ddp_profile_begin(255)).  A variant costs ~1 s instead of the ~40 s of a bench.py run,
which is what makes measuring every kernel change affordable on a 90-GPU-minute budget.

  python scripts/ab_bench.py main=ddp_amd/lib head~1=ddp_amd/lib_HEAD_1 'lds=ddp_amd/lib:DDP_GATHER=l' 't8=ddp_amd/lib:DDP_GATHER=t'

A variant is  name=libdir[:ENV=VAL[,ENV=VAL...]].  Every variant is also checked against the first one (max-rel of the
output) so that a fast-but-wrong state is visible at once; rounds alternate the variants (box drift cancels)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddp_amd import _lib  # noqa: E402
from ddp_amd.engine import DDPEngine, PackedWeights  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402
import bench  # noqa: E402

TAGS = {2: 'prologue', 7: 'layer', 8: 'tail', 9: 'gather', 1: 'xproj', 10: 'layer_tail'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='+')
    ap.add_argument('--workload', default='ade_swin_t_k3_8x512x1024')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--profile', default='init', help="weight profile (ddp_amd/utils/synthetic.py PROFILES): 'trained_like' spreads the sampling offsets")
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS[args.workload]
    task, cx = wl['task'], wl.get('feat_channels', 256)
    cm = 1 if task == 'depth' else 256
    sd = synthetic.make_state_dict(task, wl['num_classes'], wl['num_layers'], cx, seed=2, profile=args.profile)
    weights = PackedWeights(sd, task, wl['num_layers'], dev)
    x, noise = synthetic.make_inputs(wl['batch'], wl['h'], wl['w'], wl['randsteps'], cx, cm, seed=0)
    dx, dn = x.to(dev), noise.to(dev)
    kw = dict(h=wl['h'], w=wl['w'], batch=wl['batch'], randsteps=wl['randsteps'], timesteps=wl['timesteps'],
              num_classes=wl['num_classes'], bit_scale=wl['bit_scale'], accumulation=wl['accumulation'], feat_channels=cx,
              device=dev, weights=weights)
    if task == 'bev':
        kw.update(bev_input_scope=wl['bev_input_scope'], bev_output_scope=wl['bev_output_scope'])
    variants = []
    for v in args.variants:
        name, rest = v.split('=', 1)
        libdir, _, envs = rest.partition(':')
        env = dict(e.split('=', 1) for e in envs.split(',') if e)
        path = os.path.join(ROOT, libdir, 'libddp_mi355x.so') if not libdir.endswith('.so') else libdir
        variants.append((name, path, env))
    engines, ref = {}, None
    results = {n: [] for n, _, _ in variants}
    for rnd in range(args.rounds):
        for name, path, env in variants:
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                if name not in engines:
                    engines[name] = DDPEngine(sd, task, lib_path=path, **kw)
                    engines[name].prepare()
                eng = engines[name]
                out = eng.sample(dx, dn)
                torch.cuda.synchronize()
                if rnd == 0:
                    o = out.float().cpu()
                    if ref is None:
                        ref = o
                    err = float((o - ref).abs().max() / ref.abs().max())
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    eng.sample(dx, dn, out=out)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / args.reps * 1e3
                _lib.check(eng.lib.ddp_profile_begin(255))
                eng.sample(dx, dn, out=out)
                tot, n = C.c_float(0), C.c_int(0)
                _lib.check(eng.lib.ddp_profile_end(C.byref(tot), C.byref(n)))
                per = {}
                if hasattr(eng.lib, 'ddp_profile_read'):
                    for tag, tn in TAGS.items():
                        t, k = C.c_float(0), C.c_int(0)
                        eng.lib.ddp_profile_read(tag, C.byref(t), C.byref(k))
                        if k.value:
                            per[tn] = round(t.value / k.value, 4)
                rec = dict(ms_per_batch=round(ms, 3), images_per_s=round(wl['batch'] / ms * 1e3, 2), kernels_ms=per)
                if rnd == 0:
                    rec['max_rel_vs_first'] = err
                results[name].append(rec)
                print(name, json.dumps(rec), flush=True)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
    print(json.dumps({n: round(sum(r['images_per_s'] for r in rs) / len(rs), 2) for n, rs in results.items()}))


if __name__ == '__main__':
    main()
