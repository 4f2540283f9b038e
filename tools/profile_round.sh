#!/bin/bash
# Round profile: kernel stats of the default bench + HBM-traffic PMC passes of the headline step.
# Run on the GPU box from the repo root:  bash tools/profile_round.sh  (outputs under gpurun_out/prof_round/)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_stats.json 2> $O/bench_stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > $O/fetch.json 2> $O/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > $O/write.json 2> $O/write.log
ls $O/stats $O/fetch $O/write
