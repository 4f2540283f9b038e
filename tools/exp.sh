for e in 0 1 2 3; do echo "EXP=$e"; RIHIP_X6_EXP=$e timeout -k 10 200 python tools/gpass_bench.py 65536 128 2 2>&1 | grep -E "item_pass" || exit 1; done
