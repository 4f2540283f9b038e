#!/bin/bash
# PMC passes over the serve leg (IVF list-major scan, finalize, GBDT walk).  Run on the GPU box from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_serve
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/p1 -o p --output-format csv -- python3 $R/tools/serve_bench.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p2 -o p --output-format csv -- python3 $R/tools/serve_bench.py > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p3 -o p --output-format csv -- python3 $R/tools/serve_bench.py > $O/p3.log 2>&1
ls $O/p1 $O/p2 $O/p3
