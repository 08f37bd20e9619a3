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


# ---- test harness: single_gpu_test / multi_gpu_test / collect_results -----------------------------
class _ToyDataset(torch.utils.data.Dataset):
    """image i = constant plane of value i; ``pre_eval`` records (index, label sum) like a dataset's per-image tuple."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return dict(img=torch.full((3, 4, 6), float(i)), idx=i)

    def pre_eval(self, preds, indices):
        return [(int(i), int(p.sum())) for p, i in zip(preds, indices)]

    def format_results(self, results, indices, prefix='f'):
        return [f'{prefix}{int(i)}' for i in indices]


def _toy_collate(batch):
    b = len(batch)
    metas = [dict(img_shape=(4, 6, 3), ori_shape=(4, 6, 3), flip=False, idx=s['idx']) for s in batch]
    return dict(img=[torch.stack([s['img'] for s in batch])], img_metas=[metas]) if b else {}


class _ToyModel(torch.nn.Module):
    """per-image 'label map' = the image's value everywhere: lets the test see which sample landed where."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.calls = []

    def forward(self, img, img_metas, return_loss=False, rescale=True):
        assert not return_loss and isinstance(img, list) and isinstance(img_metas[0], list)
        self.calls.append((img[0].shape[0], rescale))
        return [img[0][i, 0].long().numpy() for i in range(img[0].shape[0])]


def test_single_gpu_test_batched():
    sys.path.insert(0, ROOT)
    from ddp_amd import apis
    ds = _ToyDataset(7)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=_toy_collate)
    model = _ToyModel()
    res = apis.single_gpu_test(model, loader)
    assert [int(r[0, 0]) for r in res] == list(range(7)) and [c[0] for c in model.calls] == [3, 3, 1]
    assert apis.single_gpu_test(model, loader, pre_eval=True) == [(i, i * 24) for i in range(7)]
    assert apis.single_gpu_test(model, loader, format_only=True, format_args=dict(prefix='x')) == [f'x{i}' for i in range(7)]
    assert apis.collect_results(list(range(5)), 4) == [0, 1, 2, 3]           # world 1: truncation only


def _harness_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ddp_amd import apis
    ds = _ToyDataset(7)                                                      # 7 over 2 ranks: the sampler repeats one
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, sampler=sampler, collate_fn=_toy_collate)
    out = {}
    model = _ToyModel()
    r = apis.multi_gpu_test(model, loader, gpu_collect=True)
    out['gpu'] = None if r is None else [int(a[0, 0]) for a in r]
    out['rescale'] = all(c[1] is True for c in model.calls)
    r = apis.multi_gpu_test(_ToyModel(), loader, tmpdir=os.path.join(tmp, 'collect'), pre_eval=True)
    out['cpu'] = r
    r = apis.multi_gpu_test(_ToyModel(), loader, tmpdir=None, format_only=True)
    out['fmt'] = r
    torch.save(out, os.path.join(tmp, f'h{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_gpu_test_gloo(tmp_path):
    world = 2
    port = 29900 + os.getpid() % 300
    mp.spawn(_harness_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = [torch.load(os.path.join(str(tmp_path), f'h{r}.pt')) for r in range(world)]
    assert r0['gpu'] == list(range(7)) and r1['gpu'] is None and r0['rescale'] and r1['rescale']
    assert r0['cpu'] == [(i, i * 24) for i in range(7)] and r1['cpu'] is None
    assert r0['fmt'] == [f'f{i}' for i in range(7)] and r1['fmt'] is None
