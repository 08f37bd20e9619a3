#!/usr/bin/env python
"""What does THIS box give a matrix-core kernel under its package power cap?  (the layer kernel runs at the
1400 W cap at ~1.8 GHz of a nominal 2.4 GHz, so its roofline fraction = MFMA-busy x clock ratio is bounded by power, not by
idle issue slots.)  One process, one box, every leg with package power and shader clock sampled by amdsmi (bench.PowerSampler):

  1. the headline sampling loop (the number bench.py reports), as the reference point;
  2. the vendor's bf16 GEMM (torch.matmul -> hipBLASLt) on a large square problem and on the two FFN shapes of the decoder
     layer: what AMD's own kernels sustain on this part under the same cap;
  3. scripts/ubench/mfma_chip: v_mfma_f32_32x32x16_bf16 back to back on every SIMD of the chip, no memory at all - the
     ceiling of the matrix pipes under the cap - with zero and with random operands, one and two waves per SIMD, and with
     the LDS fragment reads and VALU fillers that surround the layer kernel's MFMAs;
(A power-cap / clock-limit sweep existed until round 5; the pool runs every job at the machine's default settings - no setter of any
kind is called from this repository - so it is gone.  Everything here only READS power and clock.)

Prints one JSON object; `gpurun -- 'python scripts/power_calibration.py > gpurun_out/<tag>/power_calibration.json'`."""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ddp_amd.engine import DDPEngine, PackedWeights  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402


def sampled(loop_body, seconds, sync, settle=0.4):
    """run loop_body() back to back for `seconds`, sampling power; returns (iterations per second, power summary)"""
    ps = bench.PowerSampler()
    loop_body()
    sync()
    ps.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        loop_body()
        n += 1
        if n % 8 == 0:
            sync()
    sync()
    t1 = time.perf_counter()
    ps.stop()
    return n / (t1 - t0), ps.summary(t0 + settle, t1)


def headline(eng, dx, dn, out, batch, seconds):
    rate, pw = sampled(lambda: eng.sample(dx, dn, out=out), seconds, torch.cuda.synchronize)
    rec = {'images_per_s': round(rate * batch, 2)}
    if pw:
        rec.update(power_w=pw['power_w'], sclk_mhz=pw['sclk_mhz'], power_cap_w=pw['power_cap_w'],
                   joules_per_image=round(pw['power_w'] / (rate * batch), 3))
    return rec


def vendor_gemm(m, n, k, seconds, dev):
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    rate, pw = sampled(lambda: torch.matmul(a, b, out=c), seconds, torch.cuda.synchronize)
    tf = rate * 2.0 * m * n * k / 1e12
    rec = {'shape_mnk': [m, n, k], 'tflops_bf16': round(tf, 1), 'frac_of_2500': round(tf / 2500.0, 4)}
    if pw:
        rec.update(power_w=pw['power_w'], sclk_mhz=pw['sclk_mhz'])
    return rec


def mfma_chip(wps, chains, pattern, seconds, ldsr=0, valu=0, dma=0, hbm=0, share=0):
    exe = os.path.join(ROOT, 'scripts', 'ubench', 'mfma_chip')
    if not os.path.exists(exe):
        return {'error': 'scripts/ubench/mfma_chip not built (hipcc --offload-arch=gfx950 -O2 mfma_chip.hip -o mfma_chip)'}
    ps = bench.PowerSampler()
    ps.start()
    t0 = time.perf_counter()
    p = subprocess.run([exe, str(wps), str(chains), str(seconds), str(pattern), str(ldsr), str(valu), str(dma), str(hbm), str(share)], capture_output=True,
                       text=True, timeout=120)
    t1 = time.perf_counter()
    ps.stop()
    try:
        rec = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        return {'error': (p.stdout + p.stderr)[-300:]}
    # the timed launches are the last `seconds` of the run
    pw = ps.summary(t1 - rec['seconds'] + 0.4, t1 - 0.05)
    if pw:
        rec.update(power_w=pw['power_w'], sclk_mhz=pw['sclk_mhz'])
        if pw['sclk_mhz']:
            rec['cycles_per_mfma'] = round(pw['sclk_mhz'] * 1e6 / rec['mfma_per_simd_per_s'], 2)
        # the part sits at its cap in every leg, so the rate of a leg IS its energy per instruction
        rec['nj_per_wave_mfma'] = round(pw['power_w'] / (rec['tflops_bf16'] * 1e12 / 32768.0) * 1e9, 2)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--components', action='store_true',
                    help='only the attribution table of VERDICT r03 next #4: the layer kernel, then the MFMA + LDS-read + VALU mix alone, '
                         '+ the LDS-DMA weight stream, + the HBM fragment streams (one box, one process, watts and MHz beside each)')
    ap.add_argument('--operand-sharing', action='store_true',
                    help='only VERDICT r04 next #6 (i): nJ per MFMA when consecutive MFMAs share their A / their B / both operands '
                         '(scripts/ubench/mfma_chip.hip, registers only)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS['ade_swin_t_k3_8x512x1024']
    sd = synthetic.make_state_dict('seg', wl['num_classes'], wl['num_layers'], 256, seed=2)
    weights = PackedWeights(sd, 'seg', wl['num_layers'], dev)
    x, noise = synthetic.make_inputs(wl['batch'], wl['h'], wl['w'], wl['randsteps'], 256, 256, seed=0)
    dx, dn = x.to(dev), noise.to(dev)
    eng = DDPEngine(sd, 'seg', h=wl['h'], w=wl['w'], batch=wl['batch'], randsteps=wl['randsteps'], timesteps=wl['timesteps'],
                    num_classes=wl['num_classes'], bit_scale=wl['bit_scale'], accumulation=wl['accumulation'], feat_channels=256,
                    device=dev, weights=weights)
    eng.prepare()
    out = eng.sample(dx, dn)
    torch.cuda.synchronize()
    res = {'headline': headline(eng, dx, dn, out, wl['batch'], args.seconds)}
    print('headline', json.dumps(res['headline']), file=sys.stderr, flush=True)
    if args.operand_sharing:
        res['operand_sharing'] = []
        for name, sh in (('rotation (A every 2nd, B every MFMA)', 0), ('A shared by 3 consecutive MFMAs', 1), ('B shared by 3 consecutive MFMAs', 2),
                         ('A and B never change', 3), ('rotation (again: drift check)', 0)):
            rec = mfma_chip(1, 2, 1, args.seconds, share=sh)
            rec['leg'] = name
            res['operand_sharing'].append(rec)
            print(name, json.dumps(rec), file=sys.stderr, flush=True)
        print(json.dumps(res))
        return
    if args.components:
        legs = [('mfma_only', (0, 0, 0, 0)), ('mix', (1, 3, 0, 0)), ('mfma+dma', (0, 0, 1, 0)), ('mix+dma', (1, 3, 1, 0)),
                ('mix+dma+hbm', (1, 3, 1, 1)), ('mix (again)', (1, 3, 0, 0))]
        # (the leg "mix + HBM streams without the weight stream" faults on the box - r04b / r04c, cause not found - and is left out)
        res['components'] = []
        for name, (l, v, d, hb) in legs:
            rec = mfma_chip(1, 2, 1, args.seconds, l, v, d, hb)
            rec['leg'] = name
            res['components'].append(rec)
            print(name, json.dumps(rec), file=sys.stderr, flush=True)
        res['vendor_bf16_gemm'] = [vendor_gemm(8192, 8192, 8192, args.seconds, dev)]
        res['headline_after'] = headline(eng, dx, dn, out, wl['batch'], args.seconds)
        print(json.dumps(res))
        return
    res['vendor_bf16_gemm'] = [vendor_gemm(8192, 8192, 8192, args.seconds, dev),
                               vendor_gemm(262144, 1024, 256, args.seconds, dev),       # fc1 of one launch (one of the 6 products)
                               vendor_gemm(262144, 256, 1024, args.seconds, dev)]      # fc2
    print('vendor', json.dumps(res['vendor_bf16_gemm']), file=sys.stderr, flush=True)
    res['mfma_chip'] = [mfma_chip(1, 2, 1, args.seconds), mfma_chip(1, 2, 0, args.seconds), mfma_chip(2, 2, 1, args.seconds),
                        mfma_chip(1, 4, 1, args.seconds)]
    # the same loop with what surrounds the layer kernel's MFMAs: A fragments read from LDS (ds_read_b128 per two MFMAs) and
    # fp32 VALU fillers behind each MFMA (the FFN runs ~3 per MFMA) - still no global memory
    res['mfma_chip_with_fillers'] = [mfma_chip(1, 2, 1, args.seconds, l, v) for l, v in ((1, 0), (0, 2), (0, 4), (1, 2), (1, 3), (2, 3), (1, 4))]
    print('mfma_chip', json.dumps(res['mfma_chip']), file=sys.stderr, flush=True)
    print('mfma_chip_with_fillers', json.dumps(res['mfma_chip_with_fillers']), file=sys.stderr, flush=True)
    res['headline_after'] = headline(eng, dx, dn, out, wl['batch'], args.seconds)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
