"""Summarises the rocprofv3 --pmc passes of profiles/tools/pmc_run.sh into profiles/r01_pmc_summary.json format."""
import csv, glob, hashlib, json, statistics
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (with --kernel-trace only), command: "
               "KB_STATS=1 python profiles/tools/kbench.py {conv|wgrad} bf16 32 32 128 12 = the 128^3 32->32 layer alone; counter "
               "values in KiB as reported; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for wide coalesced "
               "reads, WRITE_SIZE exact; per launch (median over launches)"}
# bench.py only quotes this summary while the kernel source it was measured on is the one in the tree
out["kernel_source_sha16"] = hashlib.sha256(open("dg_tta_amd/csrc/conv_rows.hip", "rb").read()).hexdigest()[:16]
for what, pat in (("conv", "conv3_rows_kernel"), ("wgrad", "conv3_wgrad_tr_kernel"), ("wgrad_reduce", "wgrad_reduce_kernel")):
    ent = {"kernel": pat}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        src = "wgrad" if what == "wgrad_reduce" else what
        files = glob.glob(f"gpurun_out/pmc_{src}_{ctr}/**/*counter_collection.csv", recursive=True)
        vals = []
        for f in files:
            for r in csv.DictReader(open(f)):
                if pat in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr:
                    vals.append(float(r["Counter_Value"]))
        if vals:
            ent[ctr + "_KiB"] = {"n": len(vals), "median": statistics.median(vals), "min": min(vals), "max": max(vals)}
    if "FETCH_SIZE_KiB" in ent and "WRITE_SIZE_KiB" in ent:
        ent["fetch_bytes_corrected_median"] = ent["FETCH_SIZE_KiB"]["median"] * 1024 * 2
        ent["write_bytes_median"] = ent["WRITE_SIZE_KiB"]["median"] * 1024
    out[what + "_128cube_32to32"] = ent
print(json.dumps(out, indent=1))
