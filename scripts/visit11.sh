#!/bin/bash
set -u
OUT=gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python scripts/ab_bench.py 'qf32=ddp_amd/lib_HEAD_1' 'spread=ddp_amd/lib_spread' 'spread_p3=ddp_amd/lib' --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
timeout 400 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sample or head_forward or nan" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -3 | tee $OUT/pytest_fast.txt
timeout 100 python scripts/stamp_layer.py lib_stamp 2>&1 | grep -v amdgpu.ids > $OUT/stamps.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02k/stamps.json'))
print(d['cycles_per_tile'], {k[:22]: v['cycles'] for k,v in d['phases'].items()})
PY
