#!/bin/bash
# idle time of the GPU inside and between consecutive batched requests (tools/single_request_trace.py eager <batch>)
R=${GRAFT_REPO_ROOT:-$(pwd)}
NB=${1:-256}
O=$R/gpurun_out/prof_gaps
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o s --output-format csv -- python3 $R/tools/single_request_trace.py eager $NB > /dev/null 2>&1
python3 - <<EOF
import csv, glob
f = glob.glob("$O/**/s_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "tower_fwd" in r["Kernel_Name"]]
for a, b in list(zip(starts, starts[1:]))[-6:]:
    seg = rows[a:b]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    span = (int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    inner = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    gaps = sorted(((int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3, seg[i + 1]["Kernel_Name"][:40]) for i in range(len(seg) - 1))[-3:]
    print(f"batch: period {span:7.1f} us, first start -> last end {inner:7.1f}, kernels busy {busy:7.1f}, gap to next batch {span - inner:6.1f}; largest inner gaps {gaps}")
EOF
