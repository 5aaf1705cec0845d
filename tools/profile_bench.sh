#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [workload = c3]
# Writes gpurun_out/<tag>/: bench JSON, rocprofv3 kernel stats, PMC passes (each its own run).
set -u
TAG=${1:-prof}
WL=${2:-c3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -u bench.py --workload $WL --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-one-eighth > $OUT/bench_traced.json 2> $OUT/trace.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc1 -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-one-eighth > /dev/null 2> $OUT/pmc1.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-one-eighth > /dev/null 2> $OUT/pmc2.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-one-eighth > /dev/null 2> $OUT/pmc3.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc5 -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-one-eighth > /dev/null 2> $OUT/pmc5.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-one-eighth > /dev/null 2> $OUT/pmc4.err || exit 1
cat $OUT/bench.json
