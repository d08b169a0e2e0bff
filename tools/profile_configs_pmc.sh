#!/bin/bash
# rocprofv3 kernel statistics and SQ counter passes (one per run) of bench.py --config 2 and 4 on
# the GPU box: the reference's own 8 x 21 lattice and the calibration loss matrix.
# usage: bash tools/profile_configs_pmc.sh <tag>   (run from the repo root; writes gpurun_out/prof_<tag>/)
set -e
TAG=${1:-run}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in 2 4; do
    B="python3 $R/bench.py --config $C --steps 20 --warmup 3"
    S="python3 $R/bench.py --config $C --steps 3 --warmup 1"
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$C -o cfg$C -- $B > $O/bench_config$C.json 2> $O/err_config$C.log
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_c$C -o cfg$C -- $S > $O/bench_sq_c$C.json 2> $O/err_sq_c$C.log
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2_c$C -o cfg$C -- $S > $O/bench_sq2_c$C.json 2> $O/err_sq2_c$C.log
done
cd $R
python3 tools/pmc_summary.py $O/stats_c2 $O/pmc_sq_c2 $O/pmc_sq2_c2 > $O/summary_config2.json
python3 tools/pmc_summary.py $O/stats_c4 $O/pmc_sq_c4 $O/pmc_sq2_c4 > $O/summary_config4.json
cat $O/bench_config2.json $O/bench_config4.json
