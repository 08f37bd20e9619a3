#!/bin/bash
# GPU visit 1 (round 2): same-box A/B of the code states that were never timed + first run of the full-size parity tests
set -u
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
ab() {  # name, lib dir, extra env
  local name=$1 lib=$2; shift 2
  for rep in 1 2; do
    env "$@" DDP_LIB_PATH=$PWD/ddp_amd/$lib/libddp_mi355x.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['roofline']['avg_launch_ms'])"
  done
}
{
ab main lib A=1
ab r01k lib_310dc67 A=1
ab eb lib_try_eb A=1
ab eb_t8 lib_try_eb DDP_GATHER_T8=1
ab main lib A=1
ab r01k lib_310dc67 A=1
ab eb lib_try_eb A=1
ab eb_t8 lib_try_eb DDP_GATHER_T8=1
} 2>&1 | tee $OUT/ab.txt
# parity of the experimental library (sampler fixtures + edge geometry) before trusting its timing
DDP_LIB_PATH=$PWD/ddp_amd/lib_try_eb/libddp_mi355x.so python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sample" 2>&1 | tail -3 | tee $OUT/pytest_eb.txt
DDP_GATHER_T8=1 DDP_LIB_PATH=$PWD/ddp_amd/lib_try_eb/libddp_mi355x.so python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sample" 2>&1 | tail -3 | tee $OUT/pytest_eb_t8.txt
python -m pytest tests/test_full_size_parity.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 | tee $OUT/pytest_full.txt
