"""Summarises the rocprofv3 --pmc passes of profiles/tools/pmc_mfma.sh: median counter value per (job, kernel), derived ratios."""
import collections, csv, glob, json, re, statistics, sys
tag = sys.argv[1]
out = {"note": "rocprofv3 --pmc, one counter group per run with --kernel-trace only; job = profiles/tools/kbench.py <what> fp16 cin cout n iters batch "
               "(KB_STATS=1: forward statistics fused); values are medians over the launches of the job's dominant kernel(s); SQ_*_CYCLES in the "
               "units the guide states (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES "
               "and SQ_BUSY_CYCLES in cycles summed over SEs/XCDs as rocprofv3 reports them); FETCH_SIZE / WRITE_SIZE in KiB, FETCH x2 on gfx950"}
import hashlib
out["samples_per_launch"] = 8
out["kernel_source_sha16"] = {f: hashlib.sha256(open("dg_tta_amd/csrc/" + f, "rb").read()).hexdigest()[:16] for f in ("conv_ring.hip", "conv_rows.hip", "conv_mfma.hip", "conv_wgrad.hip", "conv_wgrad_ring.hip")}
jobs = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    job = re.search(rf"{tag}_pmc_(.+)_\d+/", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if not any(s in k for s in ("conv3_", "wgrad", "conv_ring")):
            continue
        kn = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        jobs[job][kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/**/*kernel_trace.csv", recursive=True):
    job = re.search(rf"{tag}_pmc_(.+)_\d+/", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if not any(s in k for s in ("conv3_", "wgrad", "conv_ring")):
            continue
        kn = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        jobs[job][kn]["duration_us_under_pmc"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for job, ks in jobs.items():
    out[job] = {}
    for kn, cs in ks.items():
        d = {c: statistics.median(v) for c, v in cs.items()}
        d["launches"] = max(len(v) for v in cs.values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
            d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if c in d and d.get("SQ_WAVE_CYCLES"):
                d[c + "_over_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
        out[job][kn] = d
print(json.dumps(out, indent=1))
