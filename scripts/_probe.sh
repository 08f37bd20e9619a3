timeout 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pq -o ddp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pq.log 2>&1
head -12 gpurun_out/pq/ddp_kernel_stats.csv | cut -c1-150
