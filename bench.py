#!/usr/bin/env python
"""bench.py - images/s of the DDP K-step DDIM decoder loop on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 launched by
torch.distributed.run, one rank per GPU.  One "step" = one pass of the hot path over one batch =
``ddim_sample`` of B independent images (BASELINE.json configs[1]: ADE20K Swin-T decode head,
3-step DDIM, batch 8 x 512x1024 -> x (8,256,128,256), 150 classes) with inputs already resident in
HBM.  Images shard across ranks with no data-path collective (weak scaling: B images per GPU); the
only collective is the one-off RCCL broadcast of the frozen weights (outside the timed region).

Rank 0 prints ONE JSON line.  ``roofline``: the dominant kernel is the persistent layer kernel
(b3::k_layer: output_proj + LN0, FFN fc1 + GELU + fc2 + LN1 + FiLM, and the next layer's value /
sampling projections in one launch = 89 % of the loop's flops; in the default bf16x3 engine every
fp32 product is 6 bf16 MFMA products with fp32 accumulation, so the bound is the dense bf16 MFMA
peak / 6), timed live with HIP events around each of its launches in a second pass of the same
workload.  With DDP_GEMM_MODE=f32 the dominant kernel is the fp32-MFMA fc2 GEMM + LayerNorm epilogue
and the bound is the fp32 MFMA peak.  ``cpu_baseline``: the CPU oracle (a restatement of the reference's torch path,
parity-pinned to golden vectors) timed on this box's host cores on a bounded sample (single
512x1024 images of the same workload), rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# the pool's host driver only supports dmabuf IPC (RCCL across processes needs it); exported on the boxes already
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ddp_amd import _lib  # noqa: E402
from ddp_amd.engine import DDPEngine, PackedWeights  # noqa: E402
from ddp_amd.utils import synthetic  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]
    'ade_swin_t_k3_8x512x1024': dict(task='seg', batch=8, h=128, w=256, timesteps=3, randsteps=1, num_classes=150,
                                     bit_scale=0.01, accumulation=True, num_layers=6),
    # what each of 8 GPUs gets from configs[1] under STRONG scaling (the reference's own protocol is one image per call,
    # tools/test.py:214-219): 256 tiles of 128 tokens = one tile per CU of the persistent layer kernel
    'ade_swin_t_k3_1x512x1024': dict(task='seg', batch=1, h=128, w=256, timesteps=3, randsteps=1, num_classes=150,
                                     bit_scale=0.01, accumulation=True, num_layers=6),
    'ade_swin_t_k1_1x512x512': dict(task='seg', batch=1, h=128, w=128, timesteps=1, randsteps=1, num_classes=150,
                                    bit_scale=0.01, accumulation=True, num_layers=6),
    # the per-GPU shards of BASELINE.json configs[2..4] (not the headline metric: no cpu_baseline / parity leg here,
    # parity of these task variants is covered by tests/): Cityscapes Swin-L 10-step, 32x1024x2048 over 8 GPUs
    'city_swin_l_k10_4x1024x2048': dict(task='seg', batch=4, h=256, w=512, timesteps=10, randsteps=1, num_classes=19,
                                        bit_scale=0.01, accumulation=True, num_layers=6),
    # KITTI depth, 20-step, 16x352x1216 on one GPU (regression head)
    'kitti_depth_k20_16x352x1216': dict(task='depth', batch=16, h=88, w=304, timesteps=20, randsteps=1, num_classes=1,
                                        bit_scale=0.1, accumulation=False, num_layers=6),
    # nuScenes BEV map segmentation (fusion features), 3-step, 64x200x200 over 8 GPUs
    'bev_fusion_k3_8x200x200': dict(task='bev', batch=8, h=128, w=128, timesteps=3, randsteps=1, num_classes=6,
                                    bit_scale=0.01, accumulation=False, num_layers=5, feat_channels=512,
                                    bev_input_scope=[[-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8]],
                                    bev_output_scope=[[-50, 50, 0.5], [-50, 50, 0.5]]),
    # the same head with the SHIPPED sampler setting (bev/configs/nuscenes/seg/ddp-fusion-bev256d2-lss-scale001-d5-lr5e-5.yaml:5
    # randsteps: 4): 2 samples x 4 noise replicas = the token count of the workload above
    'bev_fusion_k3_r4_2x200x200': dict(task='bev', batch=2, h=128, w=128, timesteps=3, randsteps=4, num_classes=6,
                                       bit_scale=0.01, accumulation=False, num_layers=5, feat_channels=512,
                                       bev_input_scope=[[-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8]],
                                       bev_output_scope=[[-50, 50, 0.5], [-50, 50, 0.5]]),
}
# --scaling strong: the configuration's TOTAL batch (BASELINE.json configs[1..4]) is split over the ranks
TOTAL_BATCH = {'ade_swin_t_k3_8x512x1024': 8, 'city_swin_l_k10_4x1024x2048': 32, 'kitti_depth_k20_16x352x1216': 16,
               'bev_fusion_k3_8x200x200': 64}
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16, 32 cycles/SIMD)
B3_PRODUCTS = 6                    # bf16 MFMA products per fp32-equivalent product in the bf16x3 engine
TAG_FC2_LN = 7


def flops_per_token_step(num_layers, num_classes, cx=256):
    """dense contractions only, 1 MAC = 2 flop."""
    c = 256
    per_layer = 2 * c * c + 2 * c * 64 + 2 * c * 32 + 2 * c * c + 4 * c * 1024
    return 2 * c * (cx + c) + num_layers * per_layer + 2 * c * num_classes


def usable_cores():
    """host cores this process may actually use (cgroup cpu quota, else affinity / cpu_count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def plan_affinity(allowed, world, local_rank, quota=None, numa_of_rank=None, node_cpus=None):
    """The cores rank ``local_rank`` of ``world`` pins itself (and so its data-loader workers and torch's intra-op threads) to.
    The sampler has no collective, so what N ranks share is the HOST: the measurement of round 4 (``--host-leg``) has the B = 1
    rate unchanged with seven busy neighbours but 24-31 % lower once the node is oversubscribed - a rank whose launch thread is
    descheduled in favour of another rank's loader worker stalls its GPU.  Disjoint core sets rule that out (the reference's
    launcher, tools/dist_test.sh:10-20, leaves placement to the OS).
      allowed       the process's affinity mask (os.sched_getaffinity), any iterable of core ids
      quota         cgroup cpu quota in cores (cpu.max), or None: with fewer cores than the mask shows, every rank still gets a
                    DISJOINT set, but only ``quota // world`` (at least 1) of them
      numa_of_rank  optional list: NUMA node of every local rank's GPU; node_cpus: {node: iterable of core ids}.  When both are
                    known a rank only takes cores of its GPU's node, split among the ranks of that node; otherwise the mask is
                    split evenly in rank order.
    Pure function (tests/test_host_logic.py); returns a sorted list, never empty for a non-empty mask."""
    allowed = sorted(set(int(c) for c in allowed))
    world = max(1, int(world))
    if not allowed:
        return []
    pool, peers, me = allowed, world, int(local_rank)
    if numa_of_rank is not None and node_cpus is not None and len(numa_of_rank) == world:
        # the decision is the same on every rank: NUMA-aware only when EVERY node has at least one allowed core per rank of
        # its GPUs (a mixed plan would hand the same core to two ranks)
        def cores_of(node):
            return sorted(set(int(c) for c in node_cpus.get(node, ())) & set(allowed))
        nodes = set(numa_of_rank)
        if all(n is not None and n >= 0 and len(cores_of(n)) >= sum(1 for x in numa_of_rank if x == n) for n in nodes):
            node = numa_of_rank[me]
            same = [r for r in range(world) if numa_of_rank[r] == node]
            pool, peers, me = cores_of(node), len(same), same.index(me)
    per = len(pool) // peers
    if per == 0:                              # more ranks than cores: share, round-robin
        return [pool[me % len(pool)]]
    chunk = pool[me * per:(me + 1) * per]
    if quota is not None:
        chunk = chunk[:max(1, min(len(chunk), int(quota) // world))]
    return chunk


def strong_share(workload, world, call_batch):
    """--scaling strong: a rank's share of the configuration's TOTAL batch (BASELINE.json) -> (images per call, calls per step).
    C2: 8 images -> 1 per rank at 8 GPUs; C3: 32 -> 4; C4: 16 -> 2; C5: 64 -> 8.  Images are independent (the reference's sampler
    is one image per call anyway), a share larger than the workload's call batch runs as several calls."""
    if workload not in TOTAL_BATCH:
        raise SystemExit(f'bench.py: --scaling strong needs a BASELINE configuration, one of {sorted(TOTAL_BATCH)}')
    total = TOTAL_BATCH[workload]
    if total % world:
        raise SystemExit(f'bench.py: total batch {total} of {workload} does not split over {world} ranks')
    share = total // world
    b = min(share, call_batch)
    if share % b:
        raise SystemExit(f'bench.py: per-rank share {share} is not a multiple of the call batch {b}')
    return b, share // b


def gather_rank_stats(elapsed, step_ms, images_per_rank_step, steps, dev):
    """every rank's own clock and step times -> (per_rank dict, MAX elapsed over ranks) on every rank.  A straggler must be
    visible in the line, not averaged away; the contract's figure is the MAX.  Needs an initialised process group."""
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([elapsed, sum(step_ms) / len(step_ms), min(step_ms), max(step_ms)], dtype=torch.float64, device=dev)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    rows = [[float(v) for v in g.tolist()] for g in got]
    per_rank = {'elapsed_s': [round(r[0], 4) for r in rows],
                'images_per_s': [round(images_per_rank_step * steps / r[0], 2) for r in rows],
                'step_ms_mean': [round(r[1], 3) for r in rows],
                'step_ms_min_over_ranks': round(min(r[2] for r in rows), 3), 'step_ms_max_over_ranks': round(max(r[3] for r in rows), 3),
                'slowest_rank': max(range(world), key=lambda i: rows[i][0])}
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return per_rank, float(t.item())


def _gpu_numa_nodes(world, sysfs_root='', bdfs=None):
    """NUMA node of every visible GPU (sysfs), and the cores of every node; (None, None) when the box does not say.
    ``sysfs_root`` / ``bdfs``: read another tree / take the GPUs' PCI addresses from a list instead of the HIP runtime - how
    ``--dry-run-world`` and tests/test_host_logic.py walk this code against a fake 8-GPU node."""
    try:
        nodes = []
        for i in range(world):
            if bdfs is not None:
                bdf = bdfs[i].strip().lower()
            else:
                p = torch.cuda.get_device_properties(i)
                bus = getattr(p, 'pci_bus_id', None)
                if bus is None:
                    return None, None
                if isinstance(bus, str):                      # 'dddd:bb:dd.f'
                    bdf = bus.lower()
                else:                                         # torch reports domain / bus / device as integers
                    bdf = f'{int(getattr(p, "pci_domain_id", 0)):04x}:{int(bus):02x}:{int(getattr(p, "pci_device_id", 0)):02x}.0'
            nodes.append(int(open(f'{sysfs_root}/sys/bus/pci/devices/{bdf}/numa_node').read()))
        cpus = {}
        for n in set(nodes):
            if n < 0:
                return None, None
            ids = []
            for part in open(f'{sysfs_root}/sys/devices/system/node/node{n}/cpulist').read().strip().split(','):
                a, _, b = part.partition('-')
                ids += list(range(int(a), int(b or a) + 1))
            cpus[n] = ids
        return nodes, cpus
    except Exception:
        return None, None


def pin_rank(local_rank, world, sysfs_root='', bdfs=None):
    """apply plan_affinity to this process; -> what was done (for the JSON line).  Never fatal."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        quota = None
        try:
            q, p = open('/sys/fs/cgroup/cpu.max').read().split()
            if q != 'max':
                quota = max(1, int(int(q) / int(p)))
        except Exception:
            pass
        nodes, cpus = _gpu_numa_nodes(world, sysfs_root, bdfs)
        mine = plan_affinity(allowed, world, local_rank, quota, nodes, cpus)
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, len(mine)))
        return {'cores': len(mine), 'first_core': mine[0], 'last_core': mine[-1], 'mask_cores': len(allowed), 'cgroup_quota_cores': quota,
                'numa_aware': nodes is not None, 'gpu_numa_node': None if nodes is None else nodes[local_rank]}
    except Exception as e:
        return {'error': str(e)[:160]}


def require_devices(n):
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit(f'bench.py: {n} device(s) required, {have} visible (one rank per GPU; there is no CPU path)')


def relaunch_distributed(n, check_devices=True):
    """`python bench.py --gpus N` (N > 1) without a rendezvous environment: re-exec under torch.distributed.run."""
    import socket
    import subprocess
    if check_devices:
        require_devices(n)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class _LoaderImages(torch.utils.data.Dataset):
    """what a test pipeline worker does per image, roughly: decode-sized uint8 -> float, normalise, HWC -> CHW (512x1024)"""

    def __len__(self):
        return 1 << 30

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        img = torch.randint(0, 256, (512, 1024, 3), generator=g, dtype=torch.uint8)
        img = (img.float() - torch.tensor([123.675, 116.28, 103.53])) / torch.tensor([58.395, 57.12, 57.375])
        return img.permute(2, 0, 1).contiguous()


def host_leg(task, sd, weights, kw, wl, dev, seconds, ranks=8, workers_per_gpu=2):
    """Multi-GPU readiness a ONE-GPU box can measure: the sampler has no collective, so what 8 ranks share
    is the host.  (a) host time inside one ``DDPEngine.sample()`` call (the ctypes call returns after enqueueing its ~45
    launches) next to the GPU time of that call, at B = 1 (the strong-scaling shard / the reference's one-image protocol) and
    at the workload's batch; the same for ONE hipGraph launch (``DDPEngine.capture``).  (b) the B = 1 rate with the host
    busy: `ranks - 1` busy-loop processes standing in for the other ranks' Python threads, then those plus
    ``workers_per_gpu`` more per rank (their data-loader workers) and a real DataLoader feeding this rank - plain launches
    against graph replay.  Reported under "host"; never part of ``value``."""
    import subprocess
    import threading
    cx = wl.get('feat_channels', 256)
    cm = 1 if task == 'depth' else 256
    res = {'usable_cores': usable_cores(), 'ranks_modelled': ranks, 'workers_per_gpu': workers_per_gpu}
    engines = {}
    for B in sorted({1, wl['batch']}):
        eng = DDPEngine(sd, task, **dict(kw, batch=B, weights=weights))
        x, n = synthetic.make_inputs(B, wl['h'], wl['w'], wl['randsteps'], cx, cm, seed=77)
        x, n = x.to(dev), n.to(dev)
        out = torch.empty(eng.out_shape(), dtype=torch.float32, device=dev)
        eng.sample(x, n, out=out)
        graph = eng.capture(x, n)
        engines[B] = (eng, x, n, out, graph)

        def one(fn, reps=15):
            enq, tot = [], []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                enq.append(t1 - t0)
                tot.append(t2 - t0)
            enq.sort()
            tot.sort()
            return enq[len(enq) // 2] * 1e3, tot[len(tot) // 2] * 1e3
        e_ms, t_ms = one(lambda: eng.sample(x, n, out=out))
        ge_ms, gt_ms = one(lambda: graph.replay())
        res[f'b{B}'] = {'host_enqueue_ms': round(e_ms, 3), 'call_ms_idle_stream': round(t_ms, 3), 'host_share': round(e_ms / t_ms, 3),
                        'graph_host_enqueue_ms': round(ge_ms, 3), 'graph_call_ms_idle_stream': round(gt_ms, 3),
                        'graph_host_share': round(ge_ms / gt_ms, 3)}
    eng, x, n, out, graph = engines[1]

    def rate(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        while True:
            fn()
            k += 1
            if k % 8 == 0:
                torch.cuda.synchronize()
                if time.perf_counter() - t0 >= seconds:
                    break
        torch.cuda.synchronize()
        return k / (time.perf_counter() - t0)

    def measure():
        return {'plain_images_per_s': round(rate(lambda: eng.sample(x, n, out=out)), 2),
                'graph_images_per_s': round(rate(lambda: graph.replay()), 2)}
    res['b1_idle_host'] = measure()
    busy = []
    stop = threading.Event()
    try:
        for _ in range(ranks - 1):
            busy.append(subprocess.Popen([sys.executable, '-c', 'while True: pass']))
        time.sleep(0.5)
        res['b1_other_ranks_busy'] = dict(measure(), busy_processes=len(busy))
        for _ in range((ranks - 1) * workers_per_gpu):
            busy.append(subprocess.Popen([sys.executable, '-c', 'while True: pass']))
        loader = torch.utils.data.DataLoader(_LoaderImages(), batch_size=1, num_workers=workers_per_gpu)
        fed = [0]

        def feed():                                           # this rank's own data pipeline: workers + H2D copies
            for img in loader:
                img.to(dev, non_blocking=True)
                fed[0] += 1
                if stop.is_set():
                    break
        th = threading.Thread(target=feed, daemon=True)
        th.start()
        time.sleep(1.5)
        res['b1_node_oversubscribed'] = dict(measure(), busy_processes=len(busy), loader_workers=workers_per_gpu,
                                             images_fed_meanwhile=fed[0])
        stop.set()
        th.join(timeout=10)
        del loader
    finally:
        stop.set()
        for p in busy:
            p.kill()
        for p in busy:
            p.wait()
    idle = res['b1_idle_host']
    for k in ('b1_other_ranks_busy', 'b1_node_oversubscribed'):
        res[k]['plain_drop'] = round(1 - res[k]['plain_images_per_s'] / idle['plain_images_per_s'], 4)
        res[k]['graph_drop'] = round(1 - res[k]['graph_images_per_s'] / idle['graph_images_per_s'], 4)
    res['note'] = ('B = 1 = one image per call (strong-scaling shard of configs[1]; the reference harness protocol); host_share = host '
                   'time inside the call / wall time of the call on an idle stream; drops are against the idle-host rate of the same mode')
    return res


def swin_t_standin_ms(B, H, W, dev, timed):
    """time of a torch-ROCm fp32 stand-in with the dense layers of Swin-T on a (B, 3, H, W) batch -> (ms, GFLOP per image).
    NOT part of the product: the reference's backbone stays PyTorch ('end-to-end as a secondary number')."""
    import torch.nn as nn
    import torch.nn.functional as F
    depths, C0 = (2, 2, 6, 2), 96

    class Block(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.n1, self.n2 = nn.LayerNorm(c), nn.LayerNorm(c)
            self.qkv, self.proj = nn.Linear(c, 3 * c), nn.Linear(c, c)
            self.fc1, self.fc2 = nn.Linear(c, 4 * c), nn.Linear(4 * c, c)

        def forward(self, t):
            q, k, v = self.qkv(self.n1(t)).chunk(3, -1)
            t = t + self.proj(v + 0 * (q + k))            # (window attention products left out)
            return t + self.fc2(F.gelu(self.fc1(self.n2(t))))

    class StandIn(nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = nn.Conv2d(3, C0, 4, 4)
            self.stages = nn.ModuleList([nn.Sequential(*[Block(C0 << i) for _ in range(d)]) for i, d in enumerate(depths)])
            self.merge = nn.ModuleList([nn.Linear(4 * (C0 << i), 2 * (C0 << i)) for i in range(3)])

        def forward(self, img):
            t = self.embed(img)                             # (B, C, H/4, W/4)
            outs = []
            for i, st in enumerate(self.stages):
                b, c, hh, ww = t.shape
                tok = st(t.flatten(2).transpose(1, 2))
                outs.append(tok.transpose(1, 2).reshape(b, c, hh, ww))
                if i < 3:
                    g = tok.reshape(b, hh // 2, 2, ww // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(b, (hh // 2) * (ww // 2), 4 * c)
                    t = self.merge[i](g).transpose(1, 2).reshape(b, 2 * c, hh // 2, ww // 2)
            return outs

    torch.manual_seed(0)
    net = StandIn().to(dev).eval()
    img = torch.randn(B, 3, H, W, device=dev)
    macs = 3 * 16 * C0 * (H // 4) * (W // 4)
    for i, d in enumerate(depths):
        n, c = (H >> (2 + i)) * (W >> (2 + i)), C0 << i
        macs += d * 12 * c * c * n + (8 * c * c * (n // 4) if i < 3 else 0)
    with torch.no_grad():
        ms, _ = timed(lambda: net(img), reps=3)
    return ms, 2 * macs / 1e9


class PowerSampler:
    """Package power (W) and shader clock (MHz) of one GPU, sampled by a background thread while the loop runs.
    Sources, in order: the amdsmi python binding shipped with ROCm (gpu_metrics: current_socket_power / current_gfxclk, a few
    ms per sample), `rocm-smi --showpower --showclocks --json` (~0.3 s per sample; what profiles/r02y used).  The amdgpu hwmon
    files are NOT used: on the pool's boxes power1_input / freq1_input read 296 W / 2401 MHz whatever the load (r03a).
    The part runs this loop at its package power cap, so throughput is set by energy per image: the line
    carries ``joules_per_image`` next to the roofline fraction."""

    def __init__(self, index=0, period=0.02):
        self.period = period
        self.samples = []
        self.cap_w = None
        self.source = None
        self._smi = self._handle = None
        self._stop = False
        self._thread = None
        self.first_raw = None
        try:
            sys.path.insert(0, '/opt/rocm/share/amd_smi')
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self._smi, self._handle = amdsmi, hs[min(index, len(hs) - 1)]
            if self._read_amdsmi() is None:
                raise RuntimeError('amdsmi gave no power reading')
            self.source = 'amdsmi gpu_metrics (current_socket_power, current_gfxclk)'
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(self._handle)
                c = float(cap.get('power_cap', 0))
                self.cap_w = c / 1e6 if c > 1e5 else c
            except Exception:
                pass
        except Exception:
            self._smi = None
            import shutil
            self.smi_cli = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
            if os.path.exists(self.smi_cli) and self._read_cli() is not None:
                self.source = 'rocm-smi --showpower --showclocks --json'
                self.period = 0.05

    @staticmethod
    def _num(v):
        try:
            f = float(v)
            return f if 0 < f < 60000 else None          # 65535 / "N/A" = not available
        except Exception:
            return None

    def _read_amdsmi(self):
        try:
            m = self._smi.amdsmi_get_gpu_metrics_info(self._handle)
            if self.first_raw is None:
                self.first_raw = {k: m.get(k) for k in ('current_socket_power', 'average_socket_power', 'current_gfxclk', 'current_gfxclks',
                                                        'average_gfxclk_frequency') if k in m}
            pw = self._num(m.get('current_socket_power')) or self._num(m.get('average_socket_power'))
            clks = m.get('current_gfxclks')
            ck = None
            if isinstance(clks, (list, tuple)):
                vals = [self._num(c) for c in clks]
                vals = [c for c in vals if c]
                ck = sum(vals) / len(vals) if vals else None
            ck = ck or self._num(m.get('current_gfxclk')) or self._num(m.get('average_gfxclk_frequency'))
            return (pw, ck) if pw is not None else None
        except Exception:
            return None

    def _read_cli(self):
        import re
        import subprocess
        try:
            txt = subprocess.run([self.smi_cli, '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5).stdout
            card = next(iter(json.loads(txt).values()))
            if self.first_raw is None:
                self.first_raw = {k: v for k, v in card.items() if 'Power' in k or k.startswith('sclk')}
            pw = clk = None
            for k, v in card.items():
                if 'Power' in k and '(W)' in k and pw is None:
                    pw = self._num(v)
                if k.startswith('sclk clock speed'):
                    m = re.search(r'(\d+)', str(v))
                    clk = float(m.group(1)) if m else None
            return (pw, clk) if pw is not None else None
        except Exception:
            return None

    def _read(self):
        return self._read_amdsmi() if self._smi is not None else self._read_cli()

    def _run(self):
        while not self._stop:
            r = self._read()
            if r is not None:
                self.samples.append((time.perf_counter(),) + r)
            time.sleep(self.period)

    def start(self):
        if self.source is None:
            return
        import threading
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=10)

    def summary(self, t_from, t_to):
        """mean / max over the samples taken inside [t_from, t_to] (the first 0.3 s of the loop are left out by the caller:
        the power controller needs that long to settle)."""
        ss = [x for x in self.samples if t_from <= x[0] <= t_to]
        if not ss:
            return None
        pw = [x[1] for x in ss]
        ck = [x[2] for x in ss if x[2] is not None]
        return {'power_w': round(sum(pw) / len(pw), 1), 'power_w_max': round(max(pw), 1),
                'sclk_mhz': round(sum(ck) / len(ck), 0) if ck else None, 'sclk_mhz_min': round(min(ck), 0) if ck else None,
                'samples': len(ss), 'power_cap_w': self.cap_w, 'source': self.source, 'first_raw_reading': self.first_raw}


def vendor_gemm_reference(dev, index=0, seconds=1.2, n=8192):
    """What the vendor's own bf16 GEMM (torch.matmul -> hipBLASLt) sustains on THIS box, with power and clock beside it: the
    layer kernel's roofline uses the nominal 2500 TFLOP/s, which the part does not deliver under its package power cap to
    anything that keeps the matrix pipes busy (scripts/power_calibration.py).  One line of context next to
    `roofline.frac`, measured after the timed region; never part of `value`."""
    try:
        a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        torch.matmul(a, b, out=c)
        torch.cuda.synchronize()
        ps = PowerSampler(index)
        ps.start()
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(4):
                torch.matmul(a, b, out=c)
            k += 4
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        ps.stop()
        tf = k * 2.0 * n ** 3 / (t1 - t0) / 1e12
        rec = {'kernel': f'torch.matmul bf16 {n}x{n}x{n} (hipBLASLt)', 'tflops': round(tf, 1),
               'frac_of_bf16_peak': round(tf / BF16_MFMA_PEAK_TFLOPS, 4)}
        pw = ps.summary(t0 + 0.3, t1) if ps.source is not None else None
        if pw:
            rec.update(power_w=pw['power_w'], sclk_mhz=pw['sclk_mhz'])
        return rec
    except Exception as e:                                   # context only: never fail the bench line over it
        return {'error': f'{type(e).__name__}: {e}'[:160]}


def size_stream(n_sizes, sd, wl, dev):
    """The reference's test protocol on the plugin surface: one image per call, a new (h, w) almost every call.  Returns the
    streaming rate and the cost of a geometry change relative to the loop time of the same image at a fixed geometry."""
    import random
    import ddp_amd
    enc = dict(type='DetrTransformerEncoder', num_layers=wl['num_layers'],
               transformerlayers=dict(type='BaseTransformerLayer', use_time_mlp=True,
                                      attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=256, num_levels=1, num_heads=8, dropout=0.),
                                      ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, ffn_drop=0., act_cfg=dict(type='GELU')),
                                      operation_order=('self_attn', 'norm', 'ffn', 'norm')))
    model = ddp_amd.build_segmentor(dict(
        type='DDP', timesteps=wl['timesteps'], bit_scale=wl['bit_scale'], accumulation=wl['accumulation'], randsteps=wl['randsteps'],
        decode_head=dict(type='DeformableHeadWithTime', in_channels=[256], channels=256, in_index=[0], dropout_ratio=0.,
                         num_classes=wl['num_classes'], align_corners=False, num_feature_levels=1, encoder=enc,
                         positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True, offset=-0.5)),
        train_cfg=dict(), test_cfg=dict(mode='whole')))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    # ADE20K test images: Resize(img_scale=(2048, 512), keep_ratio=True) (configs/_base_/datasets/ade20k.py:23): short side 512
    # px = 128 tokens, aspect ratio 1 .. 2 either way round
    rng = random.Random(0)
    sizes = []
    while len(sizes) < n_sizes:
        lng = (int(512 * rng.uniform(1.0, 2.0)) + 3) // 4
        hw = (128, lng) if rng.random() < 0.7 else (lng, 128)
        if hw not in sizes:
            sizes.append(hw)
    g = torch.Generator(device='cpu').manual_seed(5)
    feats = [(torch.randn(1, 256, h, w, generator=g).to(dev), torch.randn(1, wl['randsteps'], 256, h, w, generator=g).to(dev)) for h, w in sizes]

    def stream():
        for x, nz in feats:
            model.ddim_sample(x, noise=nz)
    stream()                                   # first pass: engine, packed weights, workspace growth
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stream()
    torch.cuda.synchronize()
    t_stream = time.perf_counter() - t0
    # the same images with the geometry held: per size 1 untimed + 3 timed calls
    t_fixed = 0.0
    for x, nz in feats:
        model.ddim_sample(x, noise=nz)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            model.ddim_sample(x, noise=nz)
        torch.cuda.synchronize()
        t_fixed += (time.perf_counter() - t1) / 3
    eng = next(iter(model._engine_cache.values()))
    return {'sizes': n_sizes, 'tokens_min_max': [min(h * w for h, w in sizes), max(h * w for h, w in sizes)],
            'stream_ms_per_image': round(t_stream / n_sizes * 1e3, 3), 'fixed_geometry_ms_per_image': round(t_fixed / n_sizes * 1e3, 3),
            'geometry_change_ms': round((t_stream - t_fixed) / n_sizes * 1e3, 3),
            'geometry_change_overhead': round((t_stream - t_fixed) / t_fixed, 4), 'stream_images_per_s': round(n_sizes / t_stream, 2),
            'engines_built': len(model._engine_cache), 'geometry_changes': eng.geometry_changes,
            'note': 'registered DDP segmentor, ddim_sample(x) with b = 1 and a different ADE20K-like (h, w) every call; features '
                    'resident in HBM; geometry change = DDPEngine.set_geometry (positional tables + workspace carve; weights, '
                    'split planes, weight streams and LUTs are kept)'}


class _DryEngine:
    """--dry-run-world: stands where DDPEngine stands so that the N-rank CONTROL FLOW of this file (relaunch under the launcher,
    process group, weight broadcast + replica check, pinning, barriers, per-rank statistics, the line's schema) can be walked on a
    box without GPUs.  It computes nothing - a dry run's line carries ``value: null`` and ``dry_run: true``."""
    gemm, fused_layer = 'bf16x3', True

    def __init__(self, wl, batch, call_s=0.002):
        self.shape = (batch, 1 if wl['task'] == 'depth' else wl['num_classes'], wl['h'], wl['w'])
        self.call_s = call_s

    def out_shape(self):
        return self.shape

    def prepare(self):
        pass

    def sample(self, x, noise, out=None):
        time.sleep(self.call_s)
        return out


class _HostMark:
    """torch.cuda.Event's two methods on the host clock (dry run)"""

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='ade_swin_t_k3_8x512x1024', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-images', type=int, default=2, help='images timed on the CPU oracle (bounded sample)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help="weak (default): the workload's batch per GPU.  strong: the configuration's TOTAL batch "
                         '(BASELINE.json) is split over the ranks (configs[1]: 8 images -> 1 per GPU at 8 GPUs)')
    ap.add_argument('--size-stream', type=int, default=0, metavar='N',
                    help="also time the reference's own test protocol (tools/test.py:214-219): N single-image calls of the "
                         'registered DDP segmentor, every call a different ADE20K-like map size (short side 128 tokens, '
                         'keep-ratio long side); reported under "size_stream", never part of value')
    ap.add_argument('--no-power', action='store_true', help='skip the power / clock sampling leg')
    ap.add_argument('--power-seconds', type=float, default=3.0, help='length of the back-to-back loop the power leg samples')
    ap.add_argument('--force-dist', action='store_true',
                    help='run the process-group code path (RCCL init, weight broadcast, barrier, MAX all_reduce) even at world size 1')
    ap.add_argument('--host-leg', action='store_true',
                    help='also measure host enqueue time per sample() call, plain and as a hipGraph, and the B = 1 rate with the '
                         'host busy (other ranks + data loaders modelled by busy processes); reported under "host"')
    ap.add_argument('--host-seconds', type=float, default=2.0)
    ap.add_argument('--weights', choices=sorted(synthetic.PROFILES), default='init',
                    help="weight profile of the headline number (ddp_amd/utils/synthetic.py PROFILES).  'init' (default): the reference's "
                         "initialisation + small perturbations.  'trained_like': content-dependent sampling offsets of +- 2.4 px - what "
                         'real checkpoints are more likely to look like; the default line carries its rate as the "trained_like" sub-record')
    ap.add_argument('--no-trained-like', action='store_true', help='skip the trained_like sub-record of the default line')
    ap.add_argument('--no-pin', action='store_true', help='N > 1: do not pin the ranks to disjoint core sets')
    ap.add_argument('--next-rows', action='store_true',
                    help='also time the rows either side of the loop on the same batch (SURVEY.md §8 f1/f2): FPN + '
                         'MultiStageMerging neck, fused post-loop epilogue; reported under "next_rows", never part of value')
    ap.add_argument('--dry-run-world', type=int, default=0, metavar='N',
                    help='rehearse the N-rank control flow WITHOUT GPUs (CPU tensors, gloo): relaunch under torch.distributed.run, '
                         'weight broadcast + replica check, pinning plan (DDP_BENCH_FAKE_SYSFS = root of a fake sysfs tree with a '
                         'gpus.txt of PCI addresses), barriers, per-rank statistics, one JSON line with "dry_run": true and '
                         '"value": null.  No kernel runs; nothing in the line is a measurement')
    args = ap.parse_args()
    dry = args.dry_run_world > 0
    if dry:
        args.gpus = args.dry_run_world

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (the reference's tools/dist_test.sh:10-20 does the same with
        # torch.distributed.launch --nproc_per_node=$GPUS): one rank per GPU over RCCL, rank 0 prints the JSON line
        sys.exit(relaunch_distributed(args.gpus, check_devices=not dry))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to fd 1 when the process group comes up
    # (seen in profiles/r03_force_dist_rccl_world1.json), rocm libraries may do the same.  Everything anybody writes to fd 1
    # from here on goes to stderr; the line itself is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != max(args.gpus, 1):
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} '
                         f'(or run `python bench.py --gpus {args.gpus}` without a rendezvous environment)')
    if not dry:
        require_devices(world)
    dist_on = world > 1 or args.force_dist
    # N > 1: every rank (and the loader workers / intra-op threads it spawns) on its own cores, near its GPU when the box says
    # (the split is over the ranks of THIS node: LOCAL_WORLD_SIZE under torchrun, never more than the visible devices)
    local_world = max(1, min(int(os.environ.get('LOCAL_WORLD_SIZE', world)), world, (torch.cuda.device_count() or world) if not dry else world))
    fake_sysfs, fake_bdfs = '', None
    if dry and os.environ.get('DDP_BENCH_FAKE_SYSFS'):
        fake_sysfs = os.environ['DDP_BENCH_FAKE_SYSFS']
        fake_bdfs = open(os.path.join(fake_sysfs, 'gpus.txt')).read().split()
    pinned = pin_rank(local_rank, local_world, fake_sysfs, fake_bdfs) if (world > 1 and not args.no_pin) else None
    if dry:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    n_ranks_counted = 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if dry:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        # how many ranks the group REALLY has: every rank contributes a 1 to a device all_reduce (never the --gpus flag, never
        # the environment) - the line's rccl_ranks
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        n_ranks_counted = int(ones.item())

    wl = WORKLOADS[args.workload]
    B, h, w, K = wl['batch'], wl['h'], wl['w'], wl['timesteps']
    # strong scaling: a rank's share of the configuration's total batch, processed in calls of at most the workload's
    # batch (images are independent: the reference's sampler is one image per call anyway)
    calls_per_step = 1
    if args.scaling == 'strong':
        B, calls_per_step = strong_share(args.workload, world, wl['batch'])
    # frozen weights: generated on rank 0, replicated by ONE RCCL broadcast of the packed blob
    task = wl['task']
    cx = wl.get('feat_channels', 256)
    cm = 1 if task == 'depth' else 256
    sd = synthetic.make_state_dict(task, wl['num_classes'], wl['num_layers'], cx, seed=2, profile=args.weights)
    weights = PackedWeights(sd, task, wl['num_layers'], dev)
    if dist_on:
        if rank != 0:
            weights.flat.zero_()
        weights.broadcast(src=0)
        # every rank now holds rank 0's blob: compare checksums through the group (also exercises all_gather over RCCL)
        chk = weights.flat.double().sum().reshape(1)
        got = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(got, chk)
        if not all(bool(g == got[0]) for g in got):
            raise SystemExit(f'bench.py: weight replicas differ after the RCCL broadcast: {[float(g) for g in got]}')
    kw = dict(h=h, w=w, batch=B, randsteps=wl['randsteps'], timesteps=K, num_classes=wl['num_classes'],
              bit_scale=wl['bit_scale'], accumulation=wl['accumulation'], feat_channels=cx, device=dev, weights=weights)
    if task == 'bev':
        kw.update(bev_input_scope=wl['bev_input_scope'], bev_output_scope=wl['bev_output_scope'])
    eng = _DryEngine(wl, B) if dry else DDPEngine(sd, task, **kw)
    # synthetic inputs, distinct per rank (independent images), resident in HBM before timing
    if dry:                                     # (the stub never reads them: no 268-MB draws per rank on a CPU box)
        x = noise = dx = dn = torch.zeros(1)
    else:
        x, noise = synthetic.make_inputs(B, h, w, wl['randsteps'], cx, cm, seed=1000 * rank)
        dx, dn = x.to(dev), noise.to(dev)
    out = torch.empty(eng.out_shape() if not dry else (1,), dtype=torch.float32, device=dev)
    eng.prepare()
    device_sync = (lambda: None) if dry else torch.cuda.synchronize

    def barrier():
        if dist_on:
            dist.barrier()
        device_sync()

    def one_step():
        for _ in range(calls_per_step):
            eng.sample(dx, dn, out=out)

    for _ in range(args.warmup):
        one_step()
    # per-step spread (segmentation/tools/benchmark.py:80-109 reports the rate over many iterations; SURVEY §8d asks for mean and
    # variance): one HIP event per step on the stream the library launches on - an event record is stream-ordered and costs no
    # synchronisation; the contract's clock stays the barrier-to-barrier wall time around all K steps
    marks = [_HostMark() if dry else torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        one_step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    per_rank = None
    if dist_on:
        per_rank, elapsed = gather_rank_stats(elapsed, step_ms, B * calls_per_step, args.steps, dev)
    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = B * calls_per_step * world * args.steps / elapsed
    if dry and per_rank is not None:
        per_rank['images_per_s'] = None          # (clocks of a stub: the ranks' elapsed / step times stay, as plumbing evidence)

    # ---- power leg (rank 0): the same loop back to back for a couple of seconds with package power and shader clock
    # sampled beside it; also gives the SUSTAINED rate (the timed region above is a fraction of a second)
    if dry:                                      # nothing below measures anything without a GPU
        args.no_power = args.no_roofline = args.no_trained_like = args.no_cpu_baseline = True
        args.next_rows = args.host_leg = False
        args.size_stream = 0
    power = None
    if rank == 0 and not args.no_power:
        ps = PowerSampler(local_rank)
        if ps.source is not None:
            ps.start()
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            n_steps = 0
            while True:
                one_step()
                n_steps += 1
                if n_steps % 4 == 0:
                    torch.cuda.synchronize()
                    if time.perf_counter() - tp0 >= args.power_seconds:
                        break
            torch.cuda.synchronize()
            tp1 = time.perf_counter()
            ps.stop()
            power = ps.summary(tp0 + min(0.3, 0.3 * (tp1 - tp0)), tp1)
            if power is not None:
                sustained = B * calls_per_step * n_steps / (tp1 - tp0)
                power['loop_seconds'] = round(tp1 - tp0, 2)
                power['sustained_images_per_s'] = round(sustained, 2)
                power['joules_per_image'] = round(power['power_w'] / sustained, 3)
    if dist_on:
        dist.barrier()

    # ---- roofline leg: HIP events around every launch of the dominant kernel, same workload ----------
    roofline = None
    hh, wh = eng.out_shape()[-2:]
    M = B * wl['randsteps'] * hh * wh               # tokens on the decoder grid (unused by a dry run)
    if not args.no_roofline:
        lib = _lib.load()
        _lib.check(lib.ddp_profile_begin(TAG_FC2_LN))
        reps = min(args.steps, 3)
        for _ in range(reps):
            eng.sample(dx, dn, out=out)
        tot, n = C.c_float(0), C.c_int(0)
        _lib.check(lib.ddp_profile_end(C.byref(tot), C.byref(n)))
        torch.cuda.synchronize()
        avg_ms = tot.value / max(n.value, 1)
        if eng.gemm == 'bf16x3' and eng.fused_layer:
            # layer kernel, algorithmic fp32 flops per token: output_proj 2*256*256 + FFN 2*2*256*1024 every layer,
            # + next layer's value_proj 2*256*256 and sampling projection 2*256*96 for all but the last layer;
            # averaged over the L launches of a step
            L = wl['num_layers']
            full = 2 * 256 * 256 + 4 * 256 * 1024 + 2 * 256 * 256 + 2 * 256 * 96          # 1 359 872 flop per token
            per_step = n.value / float(reps * K * calls_per_step)                          # launches of this call site per step
            if abs(per_step - (L - 1)) < 1e-6:
                # the LAST layer of a step runs fused with the step's tail under its own call site (k_layer MODE 6, tag 10): every
                # launch counted here is a whole layer with the next layer's projections
                per_tok = full
                note = f'the {L - 1} launches per step that are followed by another layer; the last layer runs fused with the step tail'
            else:
                per_tok = (L * (2 * 256 * 256 + 4 * 256 * 1024) + (L - 1) * (2 * 256 * 256 + 2 * 256 * 96)) / L
                note = 'flops averaged over the L launches of a step'
            flops_launch = per_tok * M
            peak = BF16_MFMA_PEAK_TFLOPS / B3_PRODUCTS
            kernel = ('b3::k_layer<7> (persistent: output_proj+LN0, FFN fc1+GELU+fc2+LN1+FiLM, next value/sampling proj; '
                      'fp32 products as 6 bf16 MFMA products, peak = 2500 TFLOP/s dense bf16 / 6; ' + note + ')')
        elif eng.gemm == 'bf16x3':
            flops_launch = 2.0 * 256 * 1024 * M
            peak = BF16_MFMA_PEAK_TFLOPS / B3_PRODUCTS
            kernel = 'b3::k_gemm<8,EpiResLNSB,7> (FFN fc2 + residual + LayerNorm + FiLM; peak = 2500 TFLOP/s bf16 / 6)'
        else:
            flops_launch = 2.0 * 256 * 1024 * M            # fc2: (M,1024) x (256,1024)^T, algorithmic
            peak = FP32_MFMA_PEAK_TFLOPS
            kernel = 'k_gemm_tok<8,true,EpiResLNBlk,7> (FFN fc2 + residual + LayerNorm + FiLM; fp32 MFMA)'
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12
        roofline = dict(bound='mfma', kernel=kernel,
                        achieved=round(achieved, 2), peak=round(peak, 1), unit='TFLOP/s',
                        frac=round(achieved / peak, 4), traffic=None,
                        launches=n.value, avg_launch_ms=round(avg_ms, 4),
                        flops_per_launch=flops_launch)
        # HBM bytes per launch of that kernel: measured by separate rocprofv3 --pmc passes of this same command
        # (scripts/gpu_round.sh -> scripts/collect_profiles.py; FETCH_SIZE x2 on gfx950, WRITE_SIZE uncalibrated) and
        # committed under profiles/.  Only a summary taken from the SAME kernel sources (sha over csrc/ + the header) is
        # quoted; otherwise traffic stays null rather than describing another code state.
        roofline['traffic_source'] = None
        try:
            import glob
            from ddp_amd import build as _b
            cur = _b.source_hash()
            roofline['source_sha'] = cur
            if args.workload != 'ade_swin_t_k3_8x512x1024':
                raise KeyError('PMC passes exist for the headline workload only')
            key = 'k_layer<7' if 'k_layer' in kernel else ('k_gemm<8, ddp::b3::EpiResLNSB, 7>' if eng.gemm == 'bf16x3'
                                                          else 'k_gemm_tok<8, true, ddp::EpiResLNBlk, 7>')
            for summ in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_summary.json')), reverse=True):
                d = json.load(open(summ))
                if d.get('_meta', {}).get('source_sha') != cur:
                    continue
                for name, e in d.items():
                    if key in name and 'hbm_read_bytes_per_launch' in e:
                        # writes: calibrated on the profiled box by a known-byte streaming kernel when the summary carries the factor
                        wr = e.get('hbm_write_bytes_per_launch', e.get('hbm_write_bytes_per_launch_uncalibrated', 0))
                        roofline['traffic'] = int(e['hbm_read_bytes_per_launch'] + wr)
                        cal = d.get('_meta', {}).get('hbm_calibration')
                        roofline['traffic_calibration'] = ({'write_factor': cal.get('write_factor_applied'), 'read_factor': 2.0,
                                                            'read_factor_measured': cal.get('read_factor_measured')}
                                                           if cal and 'hbm_write_bytes_per_launch' in e else None)
                        roofline['traffic_source'] = os.path.basename(summ) + ' (same kernel sources, sha ' + cur + ')'
                        break
                if roofline['traffic'] is not None:
                    break
            if roofline['traffic'] is None:
                roofline['traffic_source'] = 'no committed PMC summary matches these kernel sources (sha ' + cur + ')'
        except Exception as e:
            roofline['traffic_source'] = str(e)[:120]
        # whole-loop dense-contraction rate for context
        loop_flops = flops_per_token_step(wl['num_layers'], wl['num_classes'], cx) * float(M) * K * calls_per_step
        if power is not None:
            roofline['power_w'] = power['power_w']
            roofline['sclk_mhz'] = power['sclk_mhz']
        roofline['loop_tflops'] = round(loop_flops / (ms_per_step * 1e-3) / 1e12, 2)
        roofline['loop_frac'] = round(roofline['loop_tflops'] / peak, 4)
        if rank == 0 and not args.no_power and eng.gemm == 'bf16x3':
            roofline['same_box_vendor_gemm'] = vendor_gemm_reference(dev, local_rank)

    # ---- the same workload on the trained-like weight profile: the headline runs on the reference's
    # initialisation (offsets on the per-head ring +- 0.3 px), real checkpoints are more likely to spread the offsets with the
    # content - the LDS-staged gather then serves more taps from outside its window.  Rank 0, same inputs, its own engine.
    trained = None
    if rank == 0 and task == 'seg' and args.weights == 'init' and not args.no_trained_like:
        try:
            lib = _lib.load()
            sd_t = synthetic.make_state_dict(task, wl['num_classes'], wl['num_layers'], cx, seed=2, profile='trained_like')
            eng_t = DDPEngine(sd_t, task, **dict(kw, weights=PackedWeights(sd_t, task, wl['num_layers'], dev)))
            out_t = torch.empty_like(out)

            def gather_ms(e, o):
                _lib.check(lib.ddp_profile_begin(9))
                e.sample(dx, dn, out=o)
                tt, nn = C.c_float(0), C.c_int(0)
                _lib.check(lib.ddp_profile_end(C.byref(tt), C.byref(nn)))
                return tt.value / max(nn.value, 1)
            eng_t.sample(dx, dn, out=out_t)
            torch.cuda.synchronize()
            reps_t = max(3, min(args.steps, 10))
            tt0 = time.perf_counter()
            for _ in range(reps_t):
                eng_t.sample(dx, dn, out=out_t)
            torch.cuda.synchronize()
            ms_t = (time.perf_counter() - tt0) / reps_t * 1e3
            # the init profile again, back to back in the same seconds (the timed region above ran before the power leg)
            tt0 = time.perf_counter()
            for _ in range(reps_t):
                eng.sample(dx, dn, out=out)
            torch.cuda.synchronize()
            ms_i = (time.perf_counter() - tt0) / reps_t * 1e3
            trained = {'images_per_s': round(B / ms_t * 1e3, 2), 'ms_per_step': round(ms_t, 3),
                       'init_images_per_s_back_to_back': round(B / ms_i * 1e3, 2), 'penalty': round(1.0 - ms_i / ms_t, 4),
                       'gather_ms': round(gather_ms(eng_t, out_t), 4), 'gather_ms_init': round(gather_ms(eng, out), 4),
                       'finite': bool(torch.isfinite(out_t).all()),
                       'note': "weights = synthetic.PROFILES['trained_like'] (content-dependent offsets +- 2.4 px, peaked attention, 8x "
                               'class scores); same inputs, one call per step; never part of value'}
            del eng_t, out_t
        except Exception as e:                                       # reported, never fatal for the bench line
            trained = {'error': str(e)[:200]}

    # ---- rows either side of the loop (optional, rank 0): the neck that produces x, the epilogue that consumes out --
    next_rows = None
    if args.next_rows and rank == 0 and task == 'seg':
        import ddp_amd
        from ddp_amd.engine import seg_postprocess
        inc = [96, 192, 384, 768]                                       # Swin-T stages (configs/ade/ddp_swin_t...:41)
        fpn = ddp_amd.FPN(in_channels=inc, out_channels=256, act_cfg=None, norm_cfg=dict(type='GN', num_groups=32), num_outs=4)
        msm = ddp_amd.MultiStageMerging([256] * 4, 256, kernel_size=1, norm_cfg=dict(type='GN', num_groups=32), act_cfg=None)
        fpn.load_state_dict(synthetic.make_fpn_state_dict(inc, 1))
        msm.load_state_dict(synthetic.make_neck_state_dict(1))
        fpn, msm = fpn.to(dev).eval(), msm.to(dev).eval()
        lv = [t.to(dev) for t in synthetic.make_backbone_levels(B, inc, h, w, 1)]

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps * 1e3, r
        t_fpn, f_out = timed(lambda: fpn(lv))
        t_msm, _ = timed(lambda: msm(list(f_out)))
        # the two as the segmentor runs them: one fused C entry (ddp_amd.NeckChain, the container built from the config's neck list)
        chain = ddp_amd.NeckChain(fpn, msm)
        t_neck, _ = timed(lambda: chain(lv))
        t_post, _ = timed(lambda: seg_postprocess(out, (4 * hh, 4 * wh)))
        # The backbone is out of scope and stays PyTorch-ROCm (SURVEY §8d asks for the end-to-end split beside the loop number):
        # a torch stand-in with the dense layers of Swin-T (depths 2-2-6-2, C = 96..768: LayerNorm, qkv / proj / MLP linears,
        # GELU, patch merging; the 7x7 window attention products - 1..8 % of a block's FLOPs - are left out) on the same
        # 8 x 512 x 1024 batch, fp32 as the reference runs it (rocBLAS).
        t_bb, bb_gflop = swin_t_standin_ms(B, 4 * h, 4 * w, dev, timed)
        e2e = t_bb + t_neck + ms_per_step + t_post
        next_rows = {'neck_fpn_ms': round(t_fpn, 3), 'neck_multi_stage_merging_ms': round(t_msm, 3), 'neck_fused_fpn_msm_ms': round(t_neck, 3),
                     'post_epilogue_ms': round(t_post, 3), 'loop_ms': round(ms_per_step, 3),
                     'backbone_standin_ms': round(t_bb, 3), 'backbone_standin_gflop_per_image': round(bb_gflop, 1),
                     'end_to_end_ms': round(e2e, 3), 'end_to_end_images_per_s': round(B / e2e * 1e3, 2),
                     'note': 'same batch; backbone = torch-ROCm fp32 stand-in with the dense layers of Swin-T (the backbone itself '
                             'is out of scope and stays PyTorch-ROCm); neck inputs = synthetic backbone levels (Swin-T channels)'}

    host_res = None
    if args.host_leg and rank == 0:
        host_res = host_leg(task, sd, weights, kw, wl, dev, args.host_seconds)

    stream_res = None
    if args.size_stream > 0 and rank == 0 and task == 'seg':
        stream_res = size_stream(args.size_stream, sd, wl, dev)

    # ---- CPU baseline + parity (rank 0, N=1) ---------------------------------------------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ddp_oracle as O
        cores = usable_cores()
        torch.set_num_threads(cores)
        headline = args.workload == 'ade_swin_t_k3_8x512x1024'
        n_img = max(1, min(args.cpu_images if headline else 1, B))
        r = wl['randsteps']

        def oracle_run(b, steps=K, dtype=torch.float32, accumulation=wl['accumulation'], x0_index=None, decisions=None):
            """the reference's sampler for ONE image of the batch (the reference loop is b = 1), restated on the CPU"""
            xs, ns = x[b:b + 1].to(dtype), noise[b].to(dtype)
            sdd = sd if dtype == torch.float32 else {k: v.to(dtype) for k, v in sd.items()}
            if task == 'seg':
                return O.ddim_sample_seg(xs, ns, sdd, timesteps=steps, randsteps=r, bit_scale=wl['bit_scale'], accumulation=accumulation,
                                         x0_index=x0_index, decisions=decisions)
            if task == 'depth':
                return O.sample_depth(xs, ns, sdd, timesteps=steps, randsteps=r, bit_scale=wl['bit_scale'])
            return O.ddim_sample_bev(xs, ns, sdd, timesteps=steps, randsteps=r, bit_scale=wl['bit_scale'],
                                     input_scope=tuple(map(tuple, wl['bev_input_scope'])),
                                     output_scope=tuple(map(tuple, wl['bev_output_scope'])))
        gpu_out = out.cpu()
        tcpu = 0.0
        worst, agree, bad_px = 0.0, 1.0, 0
        ref0, dec0, worst0 = None, [], 0.0
        for b in range(n_img):
            t1 = time.perf_counter()
            ref = oracle_run(b, decisions=dec0 if (b == 0 and task == 'seg') else None)
            tcpu += time.perf_counter() - t1
            rel = (gpu_out[b:b + 1] - ref).abs().amax(1) / ref.abs().max()
            worst = max(worst, float(rel.max()))
            if b == 0:
                ref0, worst0 = ref, float(rel.max())
            bad_px += int((rel > 1e-4).sum())
            if task == 'seg':
                agree = min(agree, float((gpu_out[b:b + 1].argmax(1) == ref.argmax(1)).float().mean()))
            elif task == 'bev':
                agree = min(agree, float(((gpu_out[b:b + 1] > 0.5) == (ref > 0.5)).float().mean()))
        cpu = dict(value=round(n_img / tcpu, 4), unit='images/s', cores=torch.get_num_threads(), kind='port',
                   sample=f'{n_img} x (1 image of {args.workload}: {h}x{w} map, {K}-step sampler) of the same synthetic workload, '
                          f'torch CPU fp32 oracle, {tcpu:.1f} s')
        parity = {'max_rel_vs_oracle': worst, 'images_checked': n_img, 'pixels_above_1e-4': bad_px, 'gate': 1e-3}
        if task != 'depth':
            parity['argmax_agreement' if task == 'seg' else 'thresholded_agreement'] = agree
        # (seg, one noisy map per image) the same comparison with the loop's only discontinuity taken out: the engine records the
        # class it fed back at every step (DDP_FLAG_RECORD_X0) and the oracle takes THOSE decisions instead of its own
        # argmax - what is left is arithmetic (tests/test_full_size_parity.py asserts this figure and bounds the decisions)
        if task == 'seg' and r == 1:
            try:
                engd = DDPEngine(sd, task, **dict(kw, batch=1, record_x0=True, weights=weights))
                gd = engd.sample(dx[:1].contiguous(), dn[:1].contiguous()).cpu()
                tr = engd.x0_trace()[:, :1].cpu().long()
                refd = oracle_run(0, x0_index=[tr[s] for s in range(K)])
                parity['max_rel_decisions_fed'] = float(((gd - refd).abs().amax(1) / refd.abs().max()).max())
                parity['same_bits_as_batch_call'] = bool(torch.equal(gd, gpu_out[:1]))
                del engd
            except Exception as e:
                parity['max_rel_decisions_fed'] = {'error': str(e)[:200]}
        # (seg) the yardstick for the free-running figure: the REFERENCE restated twice - grid_sample core (its CPU path) vs
        # explicit-taps core (the arithmetic of mmcv's compiled kernel, its GPU path), and fp32 vs fp64 - drifts from
        # itself by this much on the same image when each run takes its own argmax (oracle.reference_drift_seg)
        if task == 'seg' and r == 1 and K > 1:
            try:
                # (headline workload: both variants; the larger maps: the explicit-tap core only - fp64 costs minutes there)
                dr = O.reference_drift_seg(x[:1], noise[0], sd, timesteps=K, accumulation=wl['accumulation'], bit_scale=wl['bit_scale'],
                                           base=ref0, base_decisions=dec0, variants=('taps', 'fp64') if headline else ('taps',))
                parity['free_running_image0'] = worst0
                parity['reference_vs_reference'] = {'max_rel': dr['ref_vs_ref'], 'variants': dr['variants']}
                parity['free_running_within_2x_reference_drift'] = bool(worst0 <= max(1e-3, 2 * dr['ref_vs_ref']))
            except Exception as e:
                parity['reference_vs_reference'] = {'error': str(e)[:200]}
        # The K-step output feeds argmax back into the next step, so ONE near-tie pixel that rounds the other way moves
        # a ~20x20 neighbourhood by 1e-4..1e-3 (SURVEY.md §7 hard part 1) in any fp32 implementation.  The feedback-free
        # figure: single-step scores (K=1, no accumulation) of image 0 against an fp64 evaluation of the oracle, beside
        # the fp32 oracle's own distance to it, and (seg) the number of near-tie pixels of that step.
        try:
            eng1 = DDPEngine(sd, task, **dict(kw, batch=1, timesteps=1, accumulation=False, weights=weights))
            g1 = eng1.sample(dx[:1].contiguous(), dn[:1].contiguous()).cpu().double()
            r32 = oracle_run(0, 1, torch.float32, False)
            r64 = oracle_run(0, 1, torch.float64, False)
            fb = {'gpu_rms': float((g1 - r64).pow(2).mean().sqrt()), 'gpu_max': float((g1 - r64).abs().max()),
                  'cpu_fp32_rms': float((r32.double() - r64).pow(2).mean().sqrt()), 'cpu_fp32_max': float((r32.double() - r64).abs().max()),
                  'scale': float(r64.abs().max())}
            if task == 'seg':
                top = r64.topk(2, dim=1).values
                fb['near_tie_pixels_gap_below_1e-4'] = int(((top[:, 0] - top[:, 1]) < 1e-4).sum())
            parity['single_step_vs_fp64'] = fb
        except Exception as e:                                       # reported, never fatal for the bench line
            parity['single_step_vs_fp64'] = {'error': str(e)[:200]}

    # what the process group itself says (never the --gpus flag): 1 without a group
    n_ranks = dist.get_world_size() if dist_on else 1
    assert n_ranks == world == n_ranks_counted, (n_ranks, world, n_ranks_counted)
    if rank == 0:
        line = {
            'metric': 'images/s at K DDIM steps (512x1024, 150-class) per GPU and whole node' if args.workload == 'ade_swin_t_k3_8x512x1024'
                      else f'images/s at {K} DDIM steps ({args.workload})',
            'value': None if dry else round(images_per_s, 3), 'unit': 'images/s', 'n_gpus': n_ranks, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32 (bf16x3-split MFMA products, fp32 accumulate)' if eng.gemm == 'bf16x3' else 'f32',
            'data': 'synthetic' if not dry else 'none (--dry-run-world: control-flow rehearsal on CPU / gloo, no kernel ran)',
            'dry_run': dry,
            'config': {'workload': args.workload + (' (ADE20K Swin-T DDP decode head, 3-step DDIM, batch 8x512x1024 '
                                                    'per GPU; x (8,256,128,256), random-init weights)'
                                                    if args.workload == 'ade_swin_t_k3_8x512x1024' else
                                                    f' ({task} decoder, batch {B} per GPU, x ({B},{cx},{h},{w}), random-init weights)'),
                       'images_per_gpu_per_step': B * calls_per_step, 'images_per_call': B, 'ddim_steps': K, 'tokens_per_image': h * w,
                       'parallelism': f'dp{world} (independent images, weights broadcast once)', 'gemm_engine': eng.gemm},
            # ranks counted by an all_reduce of ones through the group itself (0: no process group)
            'rccl_ranks': n_ranks_counted if (dist_on and not dry) else 0, 'group_ranks': n_ranks_counted if dist_on else 0,
            'process_group': (dist.get_backend() if dist_on else None),
            'images_per_s_per_gpu': None if dry else round(images_per_s / world, 3), 'per_rank': per_rank, 'affinity': pinned,
            'weights_profile': args.weights, 'trained_like': trained,
            'step_ms': {'mean': round(sum(step_ms) / len(step_ms), 3), 'min': round(min(step_ms), 3), 'max': round(max(step_ms), 3),
                        'std': round((sum((t - sum(step_ms) / len(step_ms)) ** 2 for t in step_ms) / len(step_ms)) ** 0.5, 3),
                        'source': 'HIP events between the steps of the timed region (rank 0)'},
            'roofline': roofline, 'power': power, 'joules_per_image': (power or {}).get('joules_per_image'),
            'cpu_baseline': cpu, 'parity': parity, 'next_rows': next_rows, 'size_stream': stream_res, 'host': host_res,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    os.close(json_fd)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
