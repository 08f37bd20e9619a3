python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for v in 2; do echo "== V=$v"; DDP_GEMM_STAGGER=0 DDP_GEMM_V=$v python scripts/gemm_probe.py 2>&1 | tail -7;  DDP_GEMM_STAGGER=0 DDP_GEMM_V=$v python scripts/gemm_probe3.py 2>&1 | tail -2 | cut -c1-250; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
