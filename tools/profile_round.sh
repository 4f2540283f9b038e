#!/bin/bash
# Round profile (run on the GPU box from the repo root: bash tools/profile_round.sh [tag] [A|B|all]); outputs under
# gpurun_out/prof_<tag>/, summarised into profiles/ by tools/summarize_profiles.py.
#   stats   : rocprofv3 --kernel-trace --stats of the default bench command
#   fetch/write : HBM traffic of the headline step (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
#   pmc_*   : SQ wave-state / MFMA-busy / LDS / traffic counters of the headline, tower, retrieval and serve kernels
# Every rocprofv3 line runs the python program directly (no env/bash hop).  The counter passes use --kernel-trace --pmc
# (kernel dispatch records are what carries the counters) and never add -s/-r or the hip/hsa/memory-copy/marker domains.
set -e
TAG=${1:-r03}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
if [ $PART != B ]; then
python3 $R/bench.py > $O/bench.json 2> $O/bench.log
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_stats.json 2> $O/bench_stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > $O/fetch.json 2> $O/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > $O/write.json 2> $O/write.log
rocprofv3 --kernel-trace --pmc $SQ1 -d $O/head1 -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2> $O/head1.log
rocprofv3 --kernel-trace --pmc $SQ2 -d $O/head2 -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2> $O/head2.log
fi
if [ $PART != A ]; then
for leg in sampled_step retrieval serve; do
  if [ $leg = sampled_step ]; then ARGS="65536"; else ARGS=""; fi
  rocprofv3 --kernel-trace --pmc $SQ1 -d $O/${leg}1 -o p --output-format csv -- python3 $R/tools/${leg}_bench.py $ARGS > $O/${leg}1.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 -d $O/${leg}2 -o p --output-format csv -- python3 $R/tools/${leg}_bench.py $ARGS > $O/${leg}2.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${leg}3 -o p --output-format csv -- python3 $R/tools/${leg}_bench.py $ARGS > $O/${leg}3.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${leg}4 -o p --output-format csv -- python3 $R/tools/${leg}_bench.py $ARGS > $O/${leg}4.log 2>&1
done
for w in 2 4 8; do python3 $R/tools/rank_shape_bench.py $w; done > $O/rank_shape_bench.log 2>&1
python3 $R/tools/gpass_bench.py 65536 128 > $O/gpass_bench.log 2>&1
python3 $R/tools/lambdamart_bench.py 20 > $O/lambdamart_bench.log 2>&1
fi
rm -f $O/*/*_agent_info.csv $O/*/*kernel_trace.csv $O/stats/b_domain_stats.csv
ls $O
