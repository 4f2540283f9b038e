#!/bin/bash
# PMC passes over the brute-force retrieval leg (scan_bf16 filter, rerank, finalize).  Run on the GPU box from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_retr
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/p1 -o p --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 3 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p2 -o p --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 3 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p3 -o p --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 3 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p4 -o p --output-format csv -- python3 $R/tools/retrieval_bench.py 4096 3 > $O/p4.log 2>&1
ls $O/*
