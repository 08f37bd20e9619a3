python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python scripts/gemm_probe.py 2>&1 | tail -7
python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
