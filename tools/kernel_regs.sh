#!/bin/bash
# Usage: tools/kernel_regs.sh [extra hipcc flags]  -- VGPRs / scratch / occupancy / LDS of every trace kernel instantiation (CPU only)
cd "$(dirname "$0")/../raytrace_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -x hip rt_kernels.hip -c -o /tmp/rk_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rec = {}; continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
    if m and cur:
        rec[m.group(1).split(" [")[0]] = int(m.group(2))
        if m.group(1).startswith("LDS"):
            if "rt_trace" in cur:
                t = re.search(r"rt_trace_(\w+?)_kernelI(.*?)EEv", cur)
                print("%-8s %-40s" % (t.group(1), t.group(2)), rec)
            cur = None
'
