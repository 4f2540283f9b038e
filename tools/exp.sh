timeout -k 10 200 python tools/gpass_bench.py 65536 128 2 2>&1 | grep -E "_pass|two_sweep|equal" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_towers.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -5
