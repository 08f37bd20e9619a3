"""Test harness around the sampling loop: ``single_gpu_test`` / ``multi_gpu_test`` with the call
protocol of segmentation/mmseg/apis/test.py:34-229 - ``model(return_loss=False, [rescale=True,] **data)`` per batch of a
DataLoader, per-image results appended in dataset order, optional ``dataset.pre_eval`` / ``dataset.format_results`` -
and the result collection of mmcv's ``collect_results_gpu`` / ``collect_results_cpu`` (rank-interleaved order of a
``DistributedSampler(shuffle=False)``, padding samples cut off at ``len(dataset)``, list on rank 0 and ``None`` elsewhere).

Two things differ from the reference on purpose:
  * ``samples_per_gpu > 1`` is valid (the reference is hard-wired to one image per GPU and iteration: "only
    samples_per_gpu=1 valid now", test.py:124-125,210-211): the MI355X sampler takes b >= 1 images with independent
    noise, ``DDP.simple_test`` returns one map per image and ``pre_eval`` is called per image with its own index;
  * nothing is exchanged between ranks while the loader runs - the only collective is ONE ``all_gather_object`` (RCCL
    when the process group is 'nccl', gloo in the CPU tests) or one shared-directory exchange at the very end.
"""
import os
import pickle
import shutil
import tempfile

import torch
import torch.distributed as dist


def _dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _model_device(model):
    for p in model.parameters():
        return p.device
    return torch.device('cpu')


def _to_device(data, device):
    """the loader hands over host tensors (the reference relies on MMDataParallel.scatter for this step)."""
    if torch.is_tensor(data):
        return data.to(device, non_blocking=True)
    if isinstance(data, (list, tuple)):
        return type(data)(_to_device(d, device) for d in data)
    return data


def _run_batch(model, data, device, **kwargs):
    data = dict(data)
    data['img'] = _to_device(data['img'], device)
    with torch.no_grad():
        return model(return_loss=False, **kwargs, **data)


def _post(dataset, result, batch_indices, pre_eval, format_only, format_args):
    if format_only:
        return dataset.format_results(result, indices=batch_indices, **(format_args or {}))
    if pre_eval:
        out = []
        for r, i in zip(result, batch_indices):          # per image: valid for samples_per_gpu > 1
            out.extend(dataset.pre_eval([r], indices=[i]))
        return out
    return result


def single_gpu_test(model, data_loader, pre_eval=False, format_only=False, format_args=None):
    """test.py:34-137 without the drawing / deprecated ``efficient_test`` branches.  Returns the list of per-image
    results (or pre-eval tuples / formatted file names) in dataset order."""
    assert not (pre_eval and format_only), '``pre_eval`` and ``format_only`` are mutually exclusive'
    model.eval()
    device = _model_device(model)
    dataset = data_loader.dataset
    results = []
    for batch_indices, data in zip(data_loader.batch_sampler, data_loader):
        result = _run_batch(model, data, device)
        results.extend(_post(dataset, result, batch_indices, pre_eval, format_only, format_args))
    return results


def multi_gpu_test(model, data_loader, tmpdir=None, gpu_collect=False, pre_eval=False, format_only=False,
                   format_args=None):
    """test.py:140-229: every rank walks its DistributedSampler shard, then the per-rank lists are merged."""
    assert not (pre_eval and format_only), '``pre_eval`` and ``format_only`` are mutually exclusive'
    model.eval()
    device = _model_device(model)
    dataset = data_loader.dataset
    results = []
    for batch_indices, data in zip(data_loader.batch_sampler, data_loader):
        result = _run_batch(model, data, device, rescale=True)
        results.extend(_post(dataset, result, batch_indices, pre_eval, format_only, format_args))
    return collect_results(results, len(dataset), tmpdir=tmpdir, gpu_collect=gpu_collect)


def _interleave(parts, size):
    """rank r of a DistributedSampler(shuffle=False) owns samples r, r + world, ...: zip the per-rank lists back
    together and drop the samples the sampler repeated to even out the shards."""
    ordered = []
    for group in zip(*parts):
        ordered.extend(group)
    # ragged tails (a rank that ran fewer batches): append what zip() dropped, still rank-interleaved
    n = min(len(p) for p in parts) if parts else 0
    longest = max((len(p) for p in parts), default=0)
    for i in range(n, longest):
        for p in parts:
            if i < len(p):
                ordered.append(p[i])
    return ordered[:size]


def collect_results(results, size, tmpdir=None, gpu_collect=False):
    """``gpu_collect=True``: one ``all_gather_object`` over the process group (mmcv ``collect_results_gpu``);
    otherwise every rank pickles its list into ``tmpdir`` and rank 0 reads them back (``collect_results_cpu``)."""
    rank, world = _dist_info()
    if world == 1:
        return results[:size]
    if gpu_collect:
        parts = [None] * world
        dist.all_gather_object(parts, results)
        return _interleave(parts, size) if rank == 0 else None
    # shared-directory exchange: rank 0 chooses the directory and tells the others
    holder = [tmpdir]
    if tmpdir is None:
        if rank == 0:
            holder[0] = tempfile.mkdtemp(prefix='ddp_amd_collect_')
        dist.broadcast_object_list(holder, src=0)
    tmpdir = holder[0]
    os.makedirs(tmpdir, exist_ok=True)
    with open(os.path.join(tmpdir, f'part_{rank}.pkl'), 'wb') as f:
        pickle.dump(results, f)
    dist.barrier()
    if rank != 0:
        return None
    parts = []
    for r in range(world):
        with open(os.path.join(tmpdir, f'part_{r}.pkl'), 'rb') as f:
            parts.append(pickle.load(f))
    shutil.rmtree(tmpdir, ignore_errors=True)
    return _interleave(parts, size)
