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
