#!/usr/bin/env python
"""Run under ``python -m torch.distributed.run --nproc-per-node 1`` on a GPU box (tests/test_bench_contract_gpu.py): the
collectives of the multi-GPU layout on DEVICE tensors through RCCL - the weight broadcast of
``PackedWeights``, the all_gather of ``gather_outputs`` with a shard that needs padding logic, the MAX all-reduce bench.py
times with - at the one world size a 1-GPU box has.  RCCL executes them (``force=True``) instead of the world-size-1
short-cut.  Prints one JSON line (stdout carries RCCL's banner too: the caller takes the last line starting with '{')."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ddp_amd import parallel
    from ddp_amd.engine import DDPEngine, PackedWeights
    from ddp_amd.utils import synthetic
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    dist.init_process_group('nccl')
    dev = torch.device('cuda', torch.cuda.current_device())
    rank, world = dist.get_rank(), dist.get_world_size()
    sd = synthetic.make_state_dict('seg', 19, 6, 256, seed=2)
    pw = PackedWeights(sd, 'seg', 6, dev)
    before = pw.flat.clone()
    pw.broadcast(src=0, force=True)                                     # ncclBroadcast of the 34 MB blob
    sums = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(sums, pw.flat.double().sum().reshape(1))             # replica check as bench.py does it
    bcast_ok = bool(torch.equal(pw.flat, before)) and all(float(s) == float(sums[0]) for s in sums)
    # ragged totals: 3 images over `world` ranks; the local shard through the real sampler, gathered on device
    total, h, w = 3, 9, 13
    x, noise = synthetic.make_inputs(total, h, w, 1, 256, 256, seed=5)
    x, noise = x.to(dev), noise.to(dev)

    def make(b):
        return DDPEngine(None, 'seg', h=h, w=w, batch=b, randsteps=1, timesteps=2, num_classes=19, bit_scale=0.01, weights=pw, device=dev)
    local, (a, b) = parallel.sample_sharded(make, x, noise, total)
    full = parallel.gather_outputs(local, total, force=True)             # ncclAllGather on CUDA tensors
    whole = make(total).sample(x, noise)
    t = torch.tensor([float(rank + 1)], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = dict(backend=dist.get_backend(), world=world, broadcast_ok=bcast_ok, shard=[a, b],
               gathered_shape=list(full.shape), gathered_is_cuda=bool(full.is_cuda),
               gather_equals_whole_batch=bool(torch.equal(full, whole)), allreduce_max=float(t))
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
