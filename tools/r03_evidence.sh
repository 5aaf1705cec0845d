#!/bin/bash
# Round-3 evidence run (GPU box, repo root): GPU tests, bench + rocprofv3 passes of C2/C3/C4, whole-image parity.
set -u
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/r03/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for wl in c3 c2 c4; do
  tools/profile_bench.sh r03_$wl $wl > gpurun_out/r03/profile_$wl.log 2>&1 || { echo "profile $wl failed"; tail -5 gpurun_out/r03/profile_$wl.log; exit 1; }
  echo "profiled $wl"
done
for wl in c2 c3 c4; do
  timeout -k 10 900 python -u tools/full_parity.py $wl > gpurun_out/r03/full_parity_$wl.log 2>&1 || { echo "full parity $wl failed"; tail -5 gpurun_out/r03/full_parity_$wl.log; exit 1; }
  tail -2 gpurun_out/r03/full_parity_$wl.log
done
