#!/usr/bin/env python
"""Per-CALL times of sample() under sustained load, with the package power / shader clock sampled beside them, for env-switched
variants of one workload in ONE process (round 6: the fused BEV step boundary timed 33 ms per batch in one ab_bench round and 38 in the
next while every tagged kernel took the same time - which calls are slow, and what does the clock do then?).

  python scripts/call_times.py --workload bev_fusion_k3_8x200x200 fused= unfused=DDP_TAIL_FUSED=0 --calls 40 --blocks 3

Per variant and block: the sorted per-call HIP-event times, their median / min / max, the mean power and clock over the block."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddp_amd.engine import DDPEngine, PackedWeights  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='+', help='name=ENV=VAL[,ENV=VAL...] (empty env list: the default path)')
    ap.add_argument('--workload', default='bev_fusion_k3_8x200x200')
    ap.add_argument('--calls', type=int, default=40)
    ap.add_argument('--blocks', type=int, default=3)
    ap.add_argument('--idle', type=float, default=0.0, help='seconds of idle GPU in front of every block')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS[args.workload]
    task, cx = wl['task'], wl.get('feat_channels', 256)
    sd = synthetic.make_state_dict(task, wl['num_classes'], wl['num_layers'], cx, seed=2)
    weights = PackedWeights(sd, task, wl['num_layers'], dev)
    x, noise = synthetic.make_inputs(wl['batch'], wl['h'], wl['w'], wl['randsteps'], cx, 1 if task == 'depth' else 256, seed=0)
    dx, dn = x.to(dev), noise.to(dev)
    kw = dict(h=wl['h'], w=wl['w'], batch=wl['batch'], randsteps=wl['randsteps'], timesteps=wl['timesteps'], num_classes=wl['num_classes'],
              bit_scale=wl['bit_scale'], accumulation=wl['accumulation'], feat_channels=cx, device=dev, weights=weights)
    if task == 'bev':
        kw.update(bev_input_scope=wl['bev_input_scope'], bev_output_scope=wl['bev_output_scope'])
    engines = {}
    for v in args.variants:
        name, _, envs = v.partition('=')
        env = dict(e.split('=', 1) for e in envs.split(',') if e)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        engines[name] = DDPEngine(sd, task, **kw)
        engines[name].prepare()
        engines[name].sample(dx, dn)
        for k, o in old.items():
            os.environ.pop(k, None) if o is None else os.environ.__setitem__(k, o)
    torch.cuda.synchronize()
    out = torch.empty(engines[next(iter(engines))].out_shape(), dtype=torch.float32, device=dev)
    ps = bench.PowerSampler(0, period=0.01)
    ps.start()
    res = []
    for blk in range(args.blocks):
        for name, eng in engines.items():
            if args.idle > 0:
                time.sleep(args.idle)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.calls + 1)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev[0].record()
            for i in range(args.calls):
                eng.sample(dx, dn, out=out)
                ev[i + 1].record()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.calls)]
            pw = ps.summary(t0, t1) or {}
            s = sorted(ms)
            rec = dict(variant=name, block=blk, median_ms=round(s[len(s) // 2], 3), min_ms=round(s[0], 3), max_ms=round(s[-1], 3),
                       mean_ms=round(sum(ms) / len(ms), 3), images_per_s_mean=round(wl['batch'] / (sum(ms) / len(ms)) * 1e3, 2),
                       power_w=pw.get('power_w'), sclk_mhz=pw.get('sclk_mhz'), sclk_mhz_min=pw.get('sclk_mhz_min'),
                       in_order_ms=[round(t, 2) for t in ms])
            res.append(rec)
            print(json.dumps(rec), flush=True)
    ps.stop()
    # clock trace: (seconds since the first block, W, MHz) every ~50 ms - what the power controller did over the whole run
    if ps.samples:
        t00 = ps.samples[0][0]
        print(json.dumps({'clock_trace': [[round(t - t00, 2), round(p), None if c is None else round(c)] for t, p, c in ps.samples[::5]]}))


if __name__ == '__main__':
    main()
