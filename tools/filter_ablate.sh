#!/bin/bash
# duration of the bf16 filter kernel for each library build given (experiments: tools/filter_ablate.sh libA.so libB.so ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  O=$R/gpurun_out/prof_abl/$(basename $so .so)
  mkdir -p $O
  RIHIP_LIB=$R/$so rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 ${ITERS:-1} > $O/run.log 2>&1 || { echo "$so failed"; tail -3 $O/run.log; exit 1; }
  python3 - <<EOF
import csv, glob
f = glob.glob("$O/**/s_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "scan_bf16_kernel<128, 0>" in r["Name"] or "scan_bf16_wide" in r["Name"]:
        print(f"$so {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
EOF
done
