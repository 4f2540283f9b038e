timeout -k 10 600 python -m pytest tests/test_gpu_towers.py tests/test_gpu_trainer.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -6 || exit 1
echo "== bwd2"; RIHIP_TOWER_BWD=2 timeout -k 10 300 python tools/microbench.py 65536 2>&1 | grep -E "bwd|full"
echo "== bwd1"; timeout -k 10 300 python tools/microbench.py 65536 2>&1 | grep -E "bwd|full"
