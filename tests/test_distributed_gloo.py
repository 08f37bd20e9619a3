"""world_size-2 test of the multi-GPU plumbing on CPU (gloo): weight broadcast, batch sharding,
output gather.  The data path itself has no collective (images are independent)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ddp_amd import parallel
    from ddp_amd.engine import PackedWeights
    from ddp_amd.utils import synthetic
    sd = synthetic.make_state_dict('seg', 19, 2, 256, seed=11)
    pw = PackedWeights(sd, 'seg', 2, 'cpu')
    ref = pw.flat.clone()
    if rank != 0:
        pw.flat.zero_()                       # only rank 0 holds the checkpoint
    pw.broadcast(src=0)
    ok_w = torch.equal(pw.flat, ref)

    total = 5                                 # ragged split: 3 + 2
    a, b = parallel.shard_range(total, rank, world)
    x = torch.arange(total * 4, dtype=torch.float32).reshape(total, 4)

    class FakeEngine:                         # stands in for DDPEngine: per-image function of its shard
        def sample(self, xs, ns):
            return xs * 2 + ns

    out, (sa, sb) = parallel.sample_sharded(lambda n: FakeEngine(), x, torch.ones_like(x))
    full = parallel.gather_outputs(out, total)
    ok_g = torch.equal(full, x * 2 + 1) and (sa, sb) == (a, b)
    torch.save(dict(ok_w=ok_w, ok_g=ok_g, shard=(a, b)), os.path.join(tmp, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    from ddp_amd.parallel import shard_range, shard_sizes
    for total in (0, 1, 7, 8, 32, 64):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(shard_sizes(total, world)) - min(shard_sizes(total, world)) <= 1


def test_broadcast_shard_gather_gloo(tmp_path):
    world = 2
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(world)]
    assert all(r['ok_w'] and r['ok_g'] for r in res)
    assert [r['shard'] for r in res] == [(0, 3), (3, 5)]
