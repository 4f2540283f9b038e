timeout -k 10 600 python -m pytest tests/test_gpu_towers.py tests/test_gpu_trainer.py tests/test_gpu_edge_cases.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -8 || exit 1
echo "== fwd2"; timeout -k 10 300 python tools/microbench.py 65536 2>&1 | grep -E "fwd|full"
echo "== fwd1"; RIHIP_TOWER_FWD=1 timeout -k 10 300 python tools/microbench.py 65536 2>&1 | grep -E "fwd|full"
