timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -15 && timeout -k 10 300 python tools/microbench.py 65536 2>&1 | tail -12
