timeout 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
echo "== bench"; timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
