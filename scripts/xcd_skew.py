#!/usr/bin/env python
"""Per-XCD completion skew of ONE launch of the layer kernel: the driver's box showed the eight XCDs
at 1761 .. 1922 MHz, alternating.  The persistent grid gives every block the same number of tiles (C2: 2048 tiles / 256 CUs = 8),
so an XCD that clocks lower finishes later and the launch ends with it.  A -DDDP_LYR_STAMP build records, per wave, the 100 MHz
chip-wide clock at the start and the end of its share of the last MODE 0 launch and the XCD it ran on:

  scripts/variant_build.sh stamp -DDDP_LYR_STAMP=1 && python scripts/xcd_skew.py           # on the GPU box

Prints per XCD: blocks, mean / max block duration, when its last block ended relative to the launch's first start, and what an
ideal re-balancing could gain (launch time vs the mean over XCDs of their finish time).  Kill criterion of the experiment: if the
slowest XCD ends <= 2 % after the mean - or if the imbalance is below ONE tile of a block's 8 - a static uneven split cannot help."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddp_amd import _lib  # noqa: E402
from ddp_amd.engine import DDPEngine  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402
import bench  # noqa: E402


def main():
    path = os.path.join(ROOT, 'ddp_amd', sys.argv[1] if len(sys.argv) > 1 else 'lib_stamp', 'libddp_mi355x.so')
    lib = _lib.load(path)
    lib.ddp_debug_set_layer_stamps.argtypes = [C.c_void_p]
    lib.ddp_debug_set_layer_stamps.restype = None
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else 'ade_swin_t_k3_8x512x1024']
    sd = synthetic.make_state_dict('seg', wl['num_classes'], 6, 256, seed=2)
    eng = DDPEngine(sd, 'seg', h=wl['h'], w=wl['w'], batch=wl['batch'], randsteps=1, timesteps=wl['timesteps'], num_classes=wl['num_classes'],
                    bit_scale=0.01, accumulation=True, device=dev, lib_path=path)
    x, noise = synthetic.make_inputs(wl['batch'], wl['h'], wl['w'], 1, 256, 256, seed=0)
    dx, dn = x.to(dev), noise.to(dev)
    for _ in range(3):                                       # warm: clocks settle under the cap
        eng.sample(dx, dn)
    torch.cuda.synchronize()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    runs = []
    for rep in range(5):
        buf = torch.zeros(n_cu * 4 * 16, dtype=torch.int64, device=dev)
        lib.ddp_debug_set_layer_stamps(buf.data_ptr())
        eng.sample(dx, dn)
        torch.cuda.synchronize()
        lib.ddp_debug_set_layer_stamps(None)
        st = buf.cpu().view(n_cu, 4, 16)
        start, end, xcc = st[:, :, 8].double(), st[:, :, 14].double(), st[:, 0, 15]
        t0 = float(start.min())
        blk_end = (end.max(1).values - t0) * 10e-3          # us (100 MHz ticks)
        blk_dur = (end.max(1).values - start.min(1).values) * 10e-3
        per = {}
        for k in sorted(set(xcc.tolist())):
            m = xcc == k
            per[int(k)] = {'blocks': int(m.sum()), 'mean_block_us': round(float(blk_dur[m].mean()), 1), 'max_block_us': round(float(blk_dur[m].max()), 1),
                           'last_block_ends_us': round(float(blk_end[m].max()), 1)}
        ends = [v['last_block_ends_us'] for v in per.values()]
        launch_us = max(ends)
        runs.append({'launch_us': round(launch_us, 1), 'mean_xcd_finish_us': round(sum(ends) / len(ends), 1),
                     'slowest_xcd_after_mean': round(launch_us / (sum(ends) / len(ends)) - 1.0, 4),
                     'slowest_minus_fastest_xcd_us': round(max(ends) - min(ends), 1),
                     'block_id_mod_8_is_xcd': bool(all(int(xcc[b]) == int(xcc[b % 8]) for b in range(n_cu))), 'per_xcd': per})
    tiles = (wl['batch'] * wl['h'] * wl['w'] + 127) // 128 / n_cu
    out = {'workload': sys.argv[2] if len(sys.argv) > 2 else 'ade_swin_t_k3_8x512x1024', 'tiles_per_block': tiles,
           'one_tile_is_this_share_of_a_block': round(1.0 / tiles, 4), 'runs': runs}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
