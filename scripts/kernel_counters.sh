#!/bin/bash
# Counter passes for the two kernels of a decoder layer at the shipped state (cache behaviour, instruction mix, LDS, waits):
# the numbers DESIGN.md §3.4 quotes for the gather.  Each group in its own rocprofv3 --pmc pass (kernel-trace only).
#   gpurun --timeout 600 -- 'bash scripts/kernel_counters.sh r03s'   ->  gpurun_out/<tag>/kernel_counters.txt
set -u
TAG=${1:-rXX}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "from ddp_amd import build; print('source_sha', build.source_hash())" > $OUT/kernel_counters.txt
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-power"
cd /tmp
pass() {
  name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o ddp -- $BENCH > $OUT/$name.log 2>&1
  echo "== $name: $*" >> $OUT/kernel_counters.txt
  d=$(dirname $(find $OUT/$name -name '*counter_collection.csv' | head -1))
  python $REPO/scripts/pmc_summary.py $d >> $OUT/kernel_counters.txt 2>&1
}
pass pmc_cache TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
pass pmc_sq SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
pass pmc_sq2 SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU
cd $REPO
cat $OUT/kernel_counters.txt
exit 0
