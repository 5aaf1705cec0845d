"""Summarise gpurun_out/<tag>/ (tools/profile_bench.sh) into profiles/<name>_{kernel_stats.csv,bench.json,pmc.json}."""
import collections, csv, glob, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), "profiles/%s_bench.json" % name)
ks = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], "profiles/%s_kernel_stats.csv" % name)
bench = json.load(open(os.path.join(src, "bench.json")))
agg = collections.defaultdict(float)
info = {}
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "rt_trace_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            info = dict(kernel=r["Kernel_Name"][:90], VGPR_Count=r["VGPR_Count"], SGPR_Count=r["SGPR_Count"],
                        LDS_Block_Size=r["LDS_Block_Size"], Grid_Size=r["Grid_Size"], Workgroup_Size=r["Workgroup_Size"])
rays = bench["roofline"]["rays_per_launch"]
d = {}
if "SQ_INSTS_VALU" in agg:
    d["valu_wave_insts_per_64_rays"] = agg["SQ_INSTS_VALU"] * 64 / rays
    d["valu_lane_utilisation"] = agg["SQ_THREAD_CYCLES_VALU"] / (agg["SQ_INSTS_VALU"] * 64)
    d["salu_per_valu"] = agg["SQ_INSTS_SALU"] / agg["SQ_INSTS_VALU"]
    # executed VALU work: active lanes summed over every VALU instruction of one launch (SURVEY 8(d) "roofline.executed")
    d["valu_lane_ops_per_launch"] = agg["SQ_THREAD_CYCLES_VALU"]
    d["valu_wave_insts_per_launch"] = agg["SQ_INSTS_VALU"]
if "FETCH_SIZE" in agg:
    # MI355X_MICROARCH.md: FETCH_SIZE (KB) reports half the bytes of wide coalesced reads on gfx950; this kernel's
    # HBM reads are the 12-byte-per-lane accum read-modify-write, so the factor is uncalibrated: both given.
    d["hbm_read_bytes_counter"] = agg["FETCH_SIZE"] * 1024
    d["hbm_read_bytes_x2_correction"] = agg["FETCH_SIZE"] * 2048
if "WRITE_SIZE" in agg:
    d["hbm_write_bytes"] = agg["WRITE_SIZE"] * 1024
if "TCC_HIT_sum" in agg:
    d["l2_hit_rate"] = agg["TCC_HIT_sum"] / max(1.0, agg["TCC_HIT_sum"] + agg["TCC_MISS_sum"])
if "SQ_LDS_BANK_CONFLICT" in agg:
    d["lds_conflict_fraction"] = agg["SQ_LDS_BANK_CONFLICT"] / max(1.0, agg["SQ_LDS_IDX_ACTIVE"])
if "SQ_WAIT_INST_ANY" in agg:
    d["wait_any_over_wave_cycles"] = agg["SQ_WAIT_ANY"] / max(1.0, agg.get("SQ_WAVE_CYCLES", 0) or 1)
if "GRBM_GUI_ACTIVE" in agg:
    d["effective_clock_ghz"] = agg["GRBM_GUI_ACTIVE"] / 8 / (bench["roofline"]["kernel_ms"] * 1e-3) / 1e9
if "TCP_TOTAL_CACHE_ACCESSES_sum" in agg and "GRBM_GUI_ACTIVE" in agg:
    # the vector L1 (TCP) serves about 1.93 divergent lane-requests per clock and CU whatever their width (4, 8 or 16
    # bytes; tools/ubench/gather_rate.hip, L1-resident window, 16 waves per CU): the bound of a per-lane tree walk
    cycles = agg["GRBM_GUI_ACTIVE"] / 8
    d["l1_requests_per_clk_per_cu"] = agg["TCP_TOTAL_CACHE_ACCESSES_sum"] / cycles / 256
    d["l1_requests_frac_of_measured_gather_peak"] = d["l1_requests_per_clk_per_cu"] / 1.93
    d["l1_hit_rate"] = 1.0 - agg["TCP_TCC_READ_REQ_sum"] / agg["TCP_TOTAL_CACHE_ACCESSES_sum"]
    d["l1_requests_per_ray"] = agg["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays
if "SQ_INSTS_VALU" in agg and "GRBM_GUI_ACTIVE" in agg:
    # What the 1 024 SIMDs issue, against what they were MEASURED to issue (tools/ubench/valu_rate.hip, 8 waves per
    # SIMD, independent instructions): v_fma_f32 1.74 ns per wave-instruction and SIMD, v_cmp + v_cndmask 1.67,
    # v_mul_f32 1.21, v_pk_fma_f32 2.12.  The nominal rate (2 cycles on a SIMD-32 = 0.83 ns) was not reached by any
    # of them; 1.7 ns is the yardstick for this instruction mix.
    d["valu_ns_per_inst_per_simd"] = bench["roofline"]["kernel_ms"] * 1e6 * 1024 / agg["SQ_INSTS_VALU"]
    d["valu_issue_vs_measured_rate"] = 1.7 / d["valu_ns_per_inst_per_simd"]
    # Round 3: SIMD cycles per instruction over ALL instructions (vector, scalar, branch, LDS, memory), for comparison with
    # tools/ubench/salu_mix.hip's 2.5-2.9 cycles per instruction of waves in lockstep.  Not the bound of these kernels
    # (profiles/r03_asm_step_experiment.json: 4.4 % fewer instructions at the same VALU count, same time): the VALU pipe is.
    if all(k in agg for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH")):
        alli = (agg["SQ_INSTS_VALU"] + agg["SQ_INSTS_SALU"] + agg["SQ_INSTS_LDS"] + agg["SQ_INSTS_VMEM_RD"] + agg["SQ_INSTS_VMEM_WR"] +
                agg["SQ_INSTS_SMEM"] + agg["SQ_INSTS_BRANCH"])
        d["insts_all_per_launch"] = alli
        d["insts_all_per_64_rays"] = alli * 64 / rays
        d["simd_cycles_per_inst_all"] = (agg["GRBM_GUI_ACTIVE"] / 8) * 1024 / alli
        d["lockstep_ubench_cycles_per_inst"] = [2.5, 2.9]
        d["simd_cycles_per_valu_inst"] = (agg["GRBM_GUI_ACTIVE"] / 8) * 1024 / agg["SQ_INSTS_VALU"]
# Only what THIS run's bench line measured itself goes into the summary: the line's traffic / executed / l1_gather blocks
# are cross-references to an EARLIER committed counter pass (bench.py marks them "measured_in_run": false) and must not
# be carried into a new profile.
roof = {k: v for k, v in bench["roofline"].items() if k not in ("traffic", "traffic_source", "executed", "l1_gather",
                                                                "counters_measured_in_run", "counters_profile_git")}
try:
    import subprocess
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "raytrace_amd"], capture_output=True, text=True,
                                timeout=10).stdout.strip())
    git = git + ("+dirty" if dirty else "")
except Exception:
    git = "unknown"
out = {"workload": bench["config"]["workload"], "git": git, "dispatch": info, "counters_one_launch": dict(agg), "derived": d,
       "bench": {k: bench[k] for k in ("value", "ms_per_step")}, "roofline": roof}
json.dump(out, open("profiles/%s_pmc.json" % name, "w"), indent=1)
if len(sys.argv) > 3:
    shutil.copy("profiles/%s_pmc.json" % name, "profiles/latest_%s_pmc.json" % sys.argv[3])
print(json.dumps(d, indent=1))
