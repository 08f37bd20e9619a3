#!/usr/bin/env python
"""Cycle accounting of the layer kernel (b3::k_layer MODE 0) with a -DDDP_LYR_STAMP build of the library:

  cd ddp_amd/csrc && for f in ddp_api ddp_gemm ddp_gemm_bf16 ddp_kernels ddp_layer_tail; do \\
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DDDP_LYR_STAMP=1 -x hip -c $f.hip -o /tmp/st_$f.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_stamp/libddp_mi355x.so /tmp/st_*.o
  python scripts/stamp_layer.py            # on the GPU box

Every wave sums s_memtime deltas per phase over its tiles; this script runs one sample of the headline workload, reads the
buffer of the LAST layer launch that projects for a next layer, and prints mean cycles per tile and the share of each
phase, next to the MFMA floor of that phase (MFMAs x 32 cycles)."""
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

PHASES = ['P0 rest (second residual fetch + stage 7)', 'P1 residual + LN0 + split', 'fc1 (32 stages)', 'GELU k-block 0 (16x, exposed)',
          'fc2 (32 stages, GELU fillers)', 'LN1 + FiLM + split + q stores', 'P3 value_proj (8 stages + stores)',
          'P3 sampling proj (3 stages + epilogues)', '-', 'tile turnaround / kernel prologue',
          'P0a tile start (issue first fragments, bias, weight frags)', 'P0b stage 0', 'P0c stages 1-5', 'P0d residual fetch + stage 6',
          '-', '-']
MFMAS = [96, 0, 3072, 0, 3072, 0, 768, 288, 0, 0, 0, 96, 480, 96, 0, 0]          # per wave and tile (slot 0 = what is left of P0: stage 7)


def main():
    path = os.path.join(ROOT, 'ddp_amd', sys.argv[1] if len(sys.argv) > 1 else 'lib_stamp', 'libddp_mi355x.so')
    lib = _lib.load(path)
    lib.ddp_debug_set_layer_stamps.argtypes = [C.c_void_p]
    lib.ddp_debug_set_layer_stamps.restype = None
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS['ade_swin_t_k3_8x512x1024']
    sd = synthetic.make_state_dict('seg', 150, 6, 256, seed=2)
    eng = DDPEngine(sd, 'seg', h=wl['h'], w=wl['w'], batch=wl['batch'], randsteps=1, timesteps=1, num_classes=150, bit_scale=0.01,
                    accumulation=True, device=dev, lib_path=path)
    x, noise = synthetic.make_inputs(wl['batch'], wl['h'], wl['w'], 1, 256, 256, seed=0)
    dx, dn = x.to(dev), noise.to(dev)
    eng.sample(dx, dn)
    torch.cuda.synchronize()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    buf = torch.zeros(n_cu * 4 * 16, dtype=torch.int64, device=dev)
    lib.ddp_debug_set_layer_stamps(buf.data_ptr())
    eng.sample(dx, dn)              # one step = 6 layer launches, summed in the buffer (5 of them project for a next layer)
    torch.cuda.synchronize()
    lib.ddp_debug_set_layer_stamps(None)
    st = buf.cpu().view(n_cu * 4, 16).double()
    st = st[st.sum(1) > 0]                                      # (builds with a smaller persistent grid leave rows empty)
    tiles = (wl['batch'] * wl['h'] * wl['w'] + 127) // 128 / (st.shape[0] / 4)
    launches = torch.tensor([6, 6, 6, 6, 6, 6, 5, 5, 1, 6, 6, 6, 6, 6, 1, 1], dtype=torch.float64)
    per_tile = st.mean(0) / tiles / launches
    tot = float(per_tile.sum())
    out = {'tiles_per_cu': tiles, 'cycles_per_tile': round(tot), 'note': 'mean over the 6 layer launches of one step (P3 over the 5 that have a next layer)', 'phases': {}}
    for name, c, m in zip(PHASES, per_tile.tolist(), MFMAS):
        if name == '-':
            continue
        out['phases'][name] = {'cycles': round(c), 'share': round(c / tot, 4), 'mfma_floor_cycles': m * 32,
                               'mfma_busy_in_phase': round(m * 32 / c, 3) if c > 0 and m else None}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
