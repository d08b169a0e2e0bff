#!/bin/bash
# rocprofv3 kernel statistics + SQ counters of the calibration loss-matrix kernel alone (GPU box).
# usage: bash tools/profile_stage1.sh <tag>
set -e
TAG=${1:-run}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/gpu_time_stage1.py"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s1 -- $P > $O/out_stats.txt 2> $O/err_stats.log
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o s1 -- $P > $O/out_sq.txt 2> $O/err_sq.log
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc_sq2 -o s1 -- $P > $O/out_sq2.txt 2> $O/err_sq2.log
cd $R
python3 tools/pmc_summary.py $O/stats $O/pmc_sq $O/pmc_sq2 > $O/summary.json
cat $O/out_stats.txt
