"""Sums the counters tools/pmc_ab.sh collected for the trace kernel: python tools/pmc_ab_sum.py <tag> -> one JSON line."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
agg = collections.defaultdict(float)
for f in sorted(glob.glob(os.path.join("gpurun_out", tag, "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "rt_trace_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
d = dict(agg)
if "TCC_MISS_sum" in d:
    d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    d["l1_hit_rate"] = 1.0 - d["TCP_TCC_READ_REQ_sum"] / max(1.0, d["TCP_TOTAL_CACHE_ACCESSES_sum"])
if "SQ_WAIT_ANY" in d:
    d["wait_any_over_wave_cycles"] = d["SQ_WAIT_ANY"] / max(1.0, d["SQ_WAVE_CYCLES"])
    d["lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64 * d["SQ_INSTS_VALU"])
run = open(os.path.join("gpurun_out", tag, "runB.txt")).read().strip().splitlines()[-1]
print(json.dumps({"tag": tag, "run": run, "counters": d}))
