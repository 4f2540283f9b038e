#!/bin/bash
# last single request of tools/single_request_trace.py: kernel, duration, gap to the previous kernel (us)
R=${GRAFT_REPO_ROOT:-$(pwd)}
MODE=${1:-eager}
NB=${2:-1}
O=$R/gpurun_out/prof_single_$MODE
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o s --output-format csv -- python3 $R/tools/single_request_trace.py $MODE $NB > /dev/null 2>&1
python3 - <<EOF
import csv, glob
f = glob.glob("$O/**/s_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last request = the kernels after the last tower forward
idx = max(i for i, r in enumerate(rows) if "tower_fwd" in r["Kernel_Name"])
prev = None
tot0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s0, s1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s0 - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{name:60s} {(s1 - s0) / 1e3:7.1f} us   gap {gap:6.1f}")
    prev = s1
print(f"first start -> last end: {(prev - tot0) / 1e3:.1f} us, {len(rows) - idx} kernels")
EOF
