#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace statistics, then counter passes (one per run).
# usage: bash tools/profile_round.sh <tag>      (run from the repo root; writes gpurun_out/prof_<tag>/)
set -e
TAG=${1:-run}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-general-lattice"
S="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-lattice"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- $B > $O/bench_stats.json 2> $O/err_stats.log
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o $TAG -- $S > $O/bench_sq.json 2> $O/err_sq.log
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -o $TAG -- $S > $O/bench_sq2.json 2> $O/err_sq2.log
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o $TAG -- $S > $O/bench_fetch.json 2> $O/err_fetch.log
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o $TAG -- $S > $O/bench_write.json 2> $O/err_write.log
cd $R
python3 tools/pmc_summary.py $O/stats $O/pmc_sq $O/pmc_sq2 $O/pmc_fetch $O/pmc_write > $O/summary.json
python3 tools/pmc_traffic.py $O/summary.json $O/bench_fetch.json > $O/pmc_traffic.json
cat $O/bench_stats.json
