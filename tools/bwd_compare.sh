#!/bin/bash
# sampled-negative step (B = 65 536, d = hidden = 128) with the three tower-backward forms: line 1 = the default (two
# kernels, tower2.hip), line 2 = RIHIP_TOWER_BWD=5 (neither chip-filling form: the fused 64-row-tile kernel of tower.hip),
# then the per-kernel summary of the default; append `RIHIP_TOWER_BWD=4 python3 tools/sampled_step_bench.py 65536` for
# the one-kernel form (tower3.hip).  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_bwd3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/sampled_step_bench.py 65536 2>&1 | tail -1
RIHIP_TOWER_BWD=5 python3 $R/tools/sampled_step_bench.py 65536 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d $O -o b --output-format csv -- python3 $R/tools/sampled_step_bench.py 65536 > /dev/null 2>&1
python3 - <<EOF
import csv, glob
f = glob.glob("$O/**/b_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f'{r["Name"][:72]:72s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s} %')
EOF
