timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 200 python bench.py --steps 10 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity'])"
