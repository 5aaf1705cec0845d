#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_ab.sh <tag> <lib.so> [script = tools/run_c4_once.py]
# Two counter passes (memory side, wave side) of one render with an alternative build; tools/pmc_ab_sum.py prints them.
set -u
OUT=gpurun_out/$1
export RT_HIP_LIB=$GRAFT_REPO_ROOT/$2
SCRIPT=${3:-tools/run_c4_once.py}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmcA -o p -- python3 $SCRIPT > $OUT/runA.txt 2> $OUT/pmcA.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmcB -o p -- python3 $SCRIPT > $OUT/runB.txt 2> $OUT/pmcB.err || exit 1
python3 tools/pmc_ab_sum.py $1
