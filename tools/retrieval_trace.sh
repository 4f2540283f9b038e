#!/bin/bash
# kernel timeline of the LAST flat top-500 search of tools/retrieval_bench.py (4096 queries x 1M x 128)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_retr_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o s --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 4 > /dev/null 2>&1
python3 - <<EOF
import csv, glob
f = glob.glob("$O/**/s_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "scan_bf16_kernel<128, 2>" in r["Kernel_Name"])
prev = None
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s0, s1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s0 - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("rihip_index::", "").replace("void ", "")[:64]
    print(f"{name:64s} {(s1 - s0) / 1e3:8.1f} us   gap {gap:6.1f}")
    prev = s1
print(f"first start -> last end: {(prev - t0) / 1e3:.1f} us, {len(rows) - idx} kernels")
EOF
