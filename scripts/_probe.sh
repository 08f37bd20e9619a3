for mode in f32 bf16x3; do echo "== MODE=$mode"; DDP_GEMM_MODE=$mode python -m pytest tests -m gpu -q -x 2>&1 | tail -6; done
for mode in f32 bf16x3; do echo "== MODE=$mode"; DDP_GEMM_MODE=$mode python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
