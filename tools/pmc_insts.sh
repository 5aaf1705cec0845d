#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_insts.sh <tag> <lib.so> [script]  -- instruction counts by kind and wait cycles of one render
set -u
OUT=gpurun_out/$1
export RT_HIP_LIB=$GRAFT_REPO_ROOT/$2
SCRIPT=${3:-tools/run_c3_once.py}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmcB -o p -- python3 $SCRIPT > $OUT/runB.txt 2> $OUT/pmcB.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmcC -o p -- python3 $SCRIPT > $OUT/runC.txt 2> $OUT/pmcC.err || exit 1
python3 tools/pmc_ab_sum.py $1
