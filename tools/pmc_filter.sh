#!/bin/bash
# PMC counters of the bf16 filter kernel for one build of the library: tools/pmc_filter.sh recommendit_amd/libX.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
so=${1:-recommendit_amd/librecommendit_hip.so}
O=$R/gpurun_out/pmc_filter/$(basename $so .so)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export RIHIP_LIB=$R/$so
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/p1 -o p --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 ${ITERS:-1} > $O/p1.log 2>&1 || { tail -3 $O/p1.log; exit 1; }
echo "pass 1 done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p2 -o p --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 ${ITERS:-1} > $O/p2.log 2>&1 || { tail -3 $O/p2.log; exit 1; }
echo "pass 2 done"
python3 - <<EOF
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("$O/" + p + "/**/p_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "scan_bf16" not in k or "<128, 2>" in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); n[k] += 1
            if "End_Timestamp" in r: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k in acc:
        print(p, k[:60], "dispatches", n[k], ("avg us %.1f" % (dur[k] / n[k])) if dur[k] else "")
        for c, v in sorted(acc[k].items()): print("   %-34s %.4g per dispatch" % (c, v / n[k]))
EOF
