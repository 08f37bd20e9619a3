#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, PMC passes (HBM traffic, MFMA busy).
# Usage (from the repo root):   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r01'
# Copy what should be judged from gpurun_out/<tag>/ into profiles/ afterwards (scripts/collect_profiles.py).
set -u
TAG=${1:-rXX}
QUICK=${2:-}          # "quick": headline workload only (bench, kernel stats, PMC, RCCL world-1) and the tests without C3 / C4 / C5 at full size
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "from ddp_amd import build; print(build.source_hash())" > $OUT/source_sha.txt
# the micro-benchmark binaries are git-ignored build products: a fresh container does not have them
for u in mfma_chip hbm_calib; do
  [ -x scripts/ubench/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench/$u.hip -o scripts/ubench/$u > $OUT/ubench_build_$u.log 2>&1
done
# -s: the full-size parity tests and the bf16x3 arithmetic tests print the figures DESIGN.md quotes
if [ "$QUICK" = quick ]; then
  KSEL='-k not(c3_cityscapes or c4_kitti or c5_bev)'
else
  KSEL=
fi
PYTEST_ORDER="python -m pytest tests -m gpu -q -s"
run_tests() { $PYTEST_ORDER ${KSEL:+"$KSEL"} 2>&1 | grep -v "amdgpu.ids\|^$" > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt; }
[ "$QUICK" = quick ] || run_tests
python bench.py --steps 10 --warmup 2 --next-rows --size-stream 50 --host-leg > $OUT/bench.json 2> $OUT/bench.err
cut -c1-2500 $OUT/bench.json; tail -3 $OUT/bench.err
# the strong-scaling shard of the headline configuration (one image per GPU at 8 GPUs), with kernel stats
python bench.py --workload ade_swin_t_k3_1x512x1024 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_ade_swin_t_k3_1x512x1024.json 2> $OUT/bench_b1.err
cut -c1-400 $OUT/bench_ade_swin_t_k3_1x512x1024.json
REPO=$PWD
# (--no-trained-like: the default line also times an engine with the trained_like weight profile, whose gather launches - same kernel
# name, 0.23 instead of 0.15 ms - were averaged into the init profile's kernel stats until round 5: the "bimodal" 0.145 - 0.292 ms)
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-power --no-trained-like"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o ddp -- $BENCH > $REPO/$OUT/prof_run.log 2>&1
# the rows either side of the loop (FPN, MultiStageMerging, post-loop epilogue) in their own kernel-stats pass
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_next -o ddp -- $BENCH --next-rows > $REPO/$OUT/prof_next_run.log 2>&1
# counters in their own passes (kernel-trace only, no other trace domains); FETCH_SIZE and WRITE_SIZE do not fit
# one pass ("exceeds the capabilities of the hardware" - rocprofv3 then hangs, hence the timeouts)
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_hbm_rd -o ddp -- $BENCH > $REPO/$OUT/pmc_hbm_rd.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_hbm_wr -o ddp -- $BENCH > $REPO/$OUT/pmc_hbm_wr.log 2>&1
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $REPO/$OUT/pmc_mfma -o ddp -- $BENCH > $REPO/$OUT/pmc_mfma.log 2>&1
# calibration of the two HBM counters on THIS box: known-byte streaming kernels in the library's access patterns (1 GiB each,
# scripts/ubench/hbm_calib.hip) under the same two passes -> factor = known bytes / counted bytes (collect_profiles.py, _meta.hbm_calibration)
CAL=$REPO/scripts/ubench/hbm_calib
[ -x $CAL ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $CAL $REPO/scripts/ubench/hbm_calib.hip
$CAL > $REPO/$OUT/hbm_calib.jsonl 2> $REPO/$OUT/hbm_calib.err
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_calib_wr -o cal -- $CAL > $REPO/$OUT/pmc_calib_wr.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_calib_rd -o cal -- $CAL > $REPO/$OUT/pmc_calib_rd.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_ade_swin_t_k3_1x512x1024 -o ddp -- python $REPO/bench.py --workload ade_swin_t_k3_1x512x1024 --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --no-power > $REPO/$OUT/prof_b1.log 2>&1
# the gather by position in the step, init and trained_like weight profiles (scripts/gather_by_position.py on kernel traces)
python $REPO/scripts/gather_by_position.py $(find $REPO/$OUT/prof -name '*kernel_trace.csv' | head -1) > $REPO/$OUT/gather_by_position_init.json 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_trained -o ddp -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-power --weights trained_like > $REPO/$OUT/prof_trained.log 2>&1
python $REPO/scripts/gather_by_position.py $(find $REPO/$OUT/prof_trained -name '*kernel_trace.csv' | head -1) > $REPO/$OUT/gather_by_position_trained.json 2>/dev/null
# the other BASELINE configurations (per-GPU shards): kernel stats + MFMA-busy counters each
WLS="city_swin_l_k10_4x1024x2048 kitti_depth_k20_16x352x1216 bev_fusion_k3_8x200x200"
[ "$QUICK" = quick ] && WLS=
for wl in $WLS; do
  B2="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-power --workload $wl"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_$wl -o ddp -- $B2 > $REPO/$OUT/prof_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $REPO/$OUT/pmc_mfma_$wl -o ddp -- $B2 > $REPO/$OUT/pmc_mfma_$wl.log 2>&1
done
cd $REPO
# the other BASELINE configurations as plain bench lines: throughput, roofline, CPU baseline and full-size parity of one image
for wl in $WLS; do
  timeout 400 python bench.py --workload $wl --steps 3 --warmup 1 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -1 $OUT/bench_$wl.json | cut -c1-200
done
# round 6: the fused step boundaries of the depth / BEV samplers against the round-5 launches (DDP_TAIL_FUSED=0), per-call HIP-event
# times in ONE process with power and clock beside them; the shipped BEV randsteps = 4 setting as a bench line
if [ "$QUICK" != quick ]; then
  timeout 400 python scripts/call_times.py --workload bev_fusion_k3_8x200x200 fused= unfused=DDP_TAIL_FUSED=0 --calls 30 --blocks 2 > $OUT/call_times_bev.jsonl 2> $OUT/call_times_bev.err
  timeout 400 python scripts/call_times.py --workload kitti_depth_k20_16x352x1216 fused= unfused=DDP_TAIL_FUSED=0 --calls 6 --blocks 2 > $OUT/call_times_kitti.jsonl 2> $OUT/call_times_kitti.err
  timeout 300 python bench.py --workload bev_fusion_k3_r4_2x200x200 --steps 5 --warmup 1 > $OUT/bench_bev_fusion_k3_r4_2x200x200.json 2> $OUT/bench_bev_r4.err
  python - $OUT <<'PY'
import json, sys
for wl in ('bev', 'kitti'):
    try:
        rows = [json.loads(l) for l in open(f'{sys.argv[1]}/call_times_{wl}.jsonl') if l.startswith('{"variant"')]
        for v in ('fused', 'unfused'):
            ms = [r['median_ms'] for r in rows if r['variant'] == v]
            print(wl, v, 'median ms per call', ms)
    except Exception as e:
        print(wl, e)
PY
fi
# the process-group path of bench.py (RCCL init, 34 MB weight broadcast, replica check by all_gather, barrier, MAX all_reduce)
# under the launcher at world size 1 - the multi-GPU code path with the one device this box has
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py \
    --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-power --force-dist > $OUT/force_dist_rccl_world1.json 2> $OUT/force_dist_rccl_world1.err
tail -1 $OUT/force_dist_rccl_world1.json | cut -c1-300
# strong scaling at N = 1 (what `--scaling strong --gpus 1` reports: the configuration's total batch in calls of the workload's batch)
for wl in ade_swin_t_k3_8x512x1024 city_swin_l_k10_4x1024x2048 bev_fusion_k3_8x200x200; do
  [ "$QUICK" = quick ] && [ $wl != ade_swin_t_k3_8x512x1024 ] && continue
  timeout 300 python bench.py --workload $wl --scaling strong --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-power > $OUT/${wl}_scaling_strong_n1.json 2> $OUT/strong_$wl.err
done
# FCN-head sampler: the loop-invariant kernels run once per engine, not per call (ddp_prepare_fcn)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_fcn -o fcn -- python $REPO/scripts/fcn_launch_count.py > $REPO/$OUT/fcn_calls.log 2>&1 )
grep "^call" $OUT/fcn_calls.log
# what costs what under the cap: MFMA alone, + LDS fragment reads + VALU fillers, + the LDS-DMA weight stream, + the HBM streams
timeout 300 python scripts/power_calibration.py --components --seconds 3 > $OUT/power_components.json 2> $OUT/power_components.err
# what this box gives the matrix pipes under its package power cap: vendor bf16 GEMM, MFMA-only loops
[ "$QUICK" = quick ] || timeout 300 python scripts/power_calibration.py > $OUT/power_calibration.json 2> $OUT/power_calibration.err
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -30 "$f"
python scripts/collect_profiles.py $TAG --print-only
# quick mode: the measurements first, the tests last (SKIP_TESTS=1: the caller runs the full suite itself)
[ "$QUICK" = quick ] && [ -z "${SKIP_TESTS:-}" ] && run_tests
exit 0
