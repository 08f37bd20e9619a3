"""The bench.py contract the driver depends on, on the GPU box: under the launcher (`python -m torch.distributed.run ...`, one
rank per GPU over RCCL - here world size 1, the one device a test box has) stdout is exactly ONE JSON line with the contract's
keys.  RCCL prints a version banner to fd 1 when the process group comes up; bench.py routes everything but its own line to
stderr."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a HIP device')
def test_stdout_is_one_json_line_under_the_launcher():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29531', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--workload', 'ade_swin_t_k3_1x512x1024', '--no-cpu-baseline', '--no-power', '--force-dist']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f'stdout must be the JSON line alone, got {len(lines)} lines: {r.stdout[:500]}'
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['value'] > 0 and d['higher_is_better'] is True
    assert d['process_group'] == 'nccl' and d['rccl_ranks'] == 1           # the RCCL path really ran
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a HIP device')
def test_rccl_collectives_on_device_tensors_world1():
    """the collectives of the multi-GPU layout on device tensors under the ``nccl`` (= RCCL) backend -
    PackedWeights.broadcast, gather_outputs after the sharded sampler, the MAX all-reduce - executed by RCCL at world size 1
    (tests/rccl_world1_probe.py; the world-2 arithmetic is covered on CPU / gloo by tests/test_distributed_gloo.py)."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'tests', 'rccl_world1_probe.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['backend'] == 'nccl' and d['world'] == 1
    assert d['broadcast_ok'] and d['gathered_is_cuda'] and d['gathered_shape'] == [3, 19, 9, 13]
    assert d['gather_equals_whole_batch'] and d['shard'] == [0, 3] and d['allreduce_max'] == 1.0


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() >= 2, reason='needs a box with exactly one GPU')
def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    """GPU-side twin of tests/test_host_logic.py::test_bench_gpus_flag_is_not_silently_ignored: asking for 2 GPUs where one is
    visible must exit non-zero with the "N device(s) required" message and print no result line."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert '2 device(s) required' in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout
