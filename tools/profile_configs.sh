#!/bin/bash
# rocprofv3 kernel statistics of the other bench.py workloads (--config 1, 2, 4) on the GPU box.
# usage: bash tools/profile_configs.sh <tag>      (run from the repo root; writes gpurun_out/prof_<tag>/)
set -e
TAG=${1:-run}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in 1 2 4; do
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$C -o cfg$C -- \
        python3 $R/bench.py --config $C --steps 20 --warmup 3 > $O/bench_config$C.json 2> $O/err_config$C.log
    cp $(find $O/stats_c$C -name '*kernel_stats.csv' | head -n 1) $O/kernel_stats_config$C.csv
done
cd $R
cat $O/bench_config*.json
