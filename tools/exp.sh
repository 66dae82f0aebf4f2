set -e
python -m pytest tests/test_gpu_w4a16.py -x -q -m gpu -k "deferred" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w4a16', d['value'], d['ms_per_step'], d.get('ttft_ms_p50'), d.get('batch_sweep_tokens_per_s'))"
