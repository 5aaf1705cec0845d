#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_c4.sh <tag>   -- PMC passes for the C4 (triangle mesh) render
set -u
TAG=${1:-c4prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/run_c4_once.py > $OUT/run.txt 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python3 tools/run_c4_once.py > /dev/null 2> $OUT/pmc1.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc2 -o p -- python3 tools/run_c4_once.py > /dev/null 2> $OUT/pmc2.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o p -- python3 tools/run_c4_once.py > /dev/null 2> $OUT/pmc3.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc4 -o p -- python3 tools/run_c4_once.py > /dev/null 2> $OUT/pmc4.err || exit 1
cat $OUT/run.txt
