#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_quick.sh <tag> [lib.so]  -- two PMC passes of one C3 step, optionally with another build
set -u
OUT=gpurun_out/${1:-pf}
[ -n "${2:-}" ] && export RT_HIP_LIB=$GRAFT_REPO_ROOT/$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc1 -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc1.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc2 -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc2.err || exit 1
echo done
