#!/bin/bash
# per-kernel totals of a short LambdaMART training run (tools/lambdamart_bench.py N trees)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_lm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 $R/tools/lambdamart_bench.py ${1:-10} > $O/run.log 2>&1
python3 - <<EOF
import csv, glob
f = glob.glob("$O/**/s_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>7s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
EOF
tail -1 $O/run.log
