"""Summary of profiles/tools/epoch_profile.sh: HBM bytes and per-kernel-name time of ONE steady-state TTA epoch =
(3-epoch run - 1-epoch run) / 2 of the same bench.py command.  FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them,
FETCH x 2 on gfx950 (MI355X_MICROARCH.md, HBM: 128-byte requests tallied at 64 bytes)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from dg_tta_amd.build import source_sha16
tag, dts = sys.argv[1], sys.argv[2:]
SHA = source_sha16()          # the kernel sources this run measured (a summary must say what it was measured on)
# weight gradients that the ring sweep takes (conv_wgrad_ring.hip): the 13 plain stride-1 3x3x3 layers at >= 16^3, forward GFLOP
# per 128^3 sample from SURVEY.md 8d (a weight gradient costs what the forward costs), x 32 sample passes per epoch
WRING_GFLOP = [43.49, 115.96, 57.98, 28.99, 14.50, 28.99, 14.50, 57.98, 28.99, 115.96, 57.98, 231.93, 115.96]
# 16-bit storage since round 6: the three 16^3 layers (enc.3.1, dec.1.0, dec.1.1) run conv3_wgrad_flat_kernel; the six split
# launches of an fp32 weight gradient keep the sweep for them
WRING_GFLOP_16 = [43.49, 115.96, 57.98, 28.99, 57.98, 28.99, 115.96, 57.98, 231.93, 115.96]


def short(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def pmc_sum(d, counter):
    tot, n = 0.0, 0
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
                n += 1
    return tot, n


def pmc_by_name(d, counter):
    """Counter sums per kernel name (KiB as rocprofv3 reports FETCH_SIZE / WRITE_SIZE)."""
    t = collections.Counter()
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                t[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return t


def by_name(d):
    t, c = collections.Counter(), collections.Counter()
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            c[k] += 1
    return t, c


prev = {}
for cand in (f"gpurun_out/{tag}_epoch_profile.json", f"profiles/{tag}_epoch_profile.json"):      # (gpurun_out/ does not travel to the box)
    if os.path.exists(cand):
        try:
            prev = json.load(open(cand))
            break
        except Exception:
            prev = {}
out = {}
for dt in dts:
    f1, n1 = pmc_sum(f"gpurun_out/{tag}_ep_{dt}_fetch_1", "FETCH_SIZE")
    f3, n3 = pmc_sum(f"gpurun_out/{tag}_ep_{dt}_fetch_3", "FETCH_SIZE")
    w1, _ = pmc_sum(f"gpurun_out/{tag}_ep_{dt}_write_1", "WRITE_SIZE")
    w3, _ = pmc_sum(f"gpurun_out/{tag}_ep_{dt}_write_3", "WRITE_SIZE")
    t1, c1 = by_name(f"gpurun_out/{tag}_ep_{dt}_stats_1")
    t3, c3 = by_name(f"gpurun_out/{tag}_ep_{dt}_stats_3")
    per = {k: (t3[k] - t1[k]) / 2 for k in t3}
    cnt = {k: (c3[k] - c1[k]) / 2 for k in c3}
    total = sum(per.values())
    top = sorted(per.items(), key=lambda kv: -kv[1])[:12]
    fetch_b, write_b = (f3 - f1) / 2 * 1024 * 2, (w3 - w1) / 2 * 1024
    # where the bytes go: per kernel name, fetched (x 2 on gfx950) + written, one steady-state epoch
    fn1, fn3 = pmc_by_name(f"gpurun_out/{tag}_ep_{dt}_fetch_1", "FETCH_SIZE"), pmc_by_name(f"gpurun_out/{tag}_ep_{dt}_fetch_3", "FETCH_SIZE")
    wn1, wn3 = pmc_by_name(f"gpurun_out/{tag}_ep_{dt}_write_1", "WRITE_SIZE"), pmc_by_name(f"gpurun_out/{tag}_ep_{dt}_write_3", "WRITE_SIZE")
    by_kernel = {k: ((fn3[k] - fn1[k]) / 2 * 1024 * 2, (wn3[k] - wn1[k]) / 2 * 1024) for k in set(fn3) | set(wn3)}
    hbm_top = sorted(by_kernel.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:16]
    old = prev.get("fp32" if dt == "fp32" else "16bit", {})
    measured_on = {"time": SHA if top else None, "traffic": SHA if n3 else None}
    if n3 == 0 and old:          # the PMC passes were not re-run: keep the recorded traffic UNDER ITS OWN STAMP
        fetch_b, write_b = old.get("fetch_bytes_per_epoch", 0.0), old.get("write_bytes_per_epoch", 0.0)
        n1, n3 = 0, 2 * old.get("dispatches_per_epoch", 0)
        measured_on["traffic"] = (old.get("measured_on") or {}).get("traffic", "unrecorded (before round 6)")
        measured_on["traffic_carried_over_from"] = "the previous summary of this tag"
    ent = {"size": 128, "accum": 16, "storage": dt,
           "how": "(3-epoch run - 1-epoch run) / 2 of `bench.py --dtype %s --warmup 0 --weights he --no-fp32 --no-cpu-baseline "
                  "--inference-size 0` under rocprofv3: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace "
                  "only (FETCH x 2 on gfx950), --kernel-trace --stats with DGTTA_WGRAD_STREAM=0 DGTTA_PIPELINE_PREP=0 (one stream, so that durations add up) for the per-name time" % dt,
           "dispatches_per_epoch": (n3 - n1) / 2,
           "fetch_bytes_per_epoch": fetch_b, "write_bytes_per_epoch": write_b, "hbm_bytes_per_epoch": fetch_b + write_b,
           "hbm_bytes_by_kernel": [{"kernel": k, "fetch_gb": round(v[0] / 1e9, 2), "write_gb": round(v[1] / 1e9, 2)} for k, v in hbm_top] or
                                  old.get("hbm_bytes_by_kernel", []),
           "hbm_bytes_by_kernel_note": "FETCH_SIZE x 2 holds for strided 16-byte reads too (profiles/tools/fetch_calib.sh): the memory side moves whole "
                                       "128-byte lines.  conv_s2_regs_kernel<2, 2> reads the skip half of the [voxel][64] concat buffer - 64 used "
                                       "bytes per line - so half of its fetched bytes are over-fetch (profiles/r05_ab.txt)",
           "kernel_ms_per_epoch": round(total, 2),
           "top_kernels_ms_per_epoch": [{"kernel": k, "ms": round(v, 3), "launches": cnt[k], "share": round(v / total, 4)} for k, v in top]}
    if not top and old:
        ent["kernel_ms_per_epoch"], ent["top_kernels_ms_per_epoch"] = old.get("kernel_ms_per_epoch"), old.get("top_kernels_ms_per_epoch", [])
        ent["largest_consumer"] = old.get("largest_consumer")
        measured_on["time"] = (old.get("measured_on") or {}).get("time", "unrecorded (before round 6)")
        measured_on["time_carried_over_from"] = "the previous summary of this tag"
    ent["measured_on"] = measured_on
    if top:
        k, v = top[0]
        lc = {"kernel": k, "ms_per_epoch": round(v, 2), "share_of_kernel_time": round(v / total, 4), "launches_per_epoch": cnt[k]}
        if "wgrad_ring" in k:
            # the family: every instantiation of the sweep (the first layer runs the two-taps-per-MFMA form under its own name)
            fam = {kk: vv for kk, vv in per.items() if "conv3_wgrad_ring_kernel" in kk}
            v = sum(fam.values())
            lc.update(kernel="conv3_wgrad_ring_kernel (all instantiations)", ms_per_epoch=round(v, 2), share_of_kernel_time=round(v / total, 4),
                      launches_per_epoch=sum(cnt[kk] for kk in fam), instantiations={kk: round(vv, 2) for kk, vv in fam.items()})
            gl = WRING_GFLOP if dt == "fp32" else WRING_GFLOP_16
            tf = sum(gl) * 32 / 1e3
            lc.update(tflop_per_epoch=round(tf, 2), achieved_tflops=round(tf / (v * 1e-3), 1),
                      frac_of_peak=round(tf / (v * 1e-3) / (157.3 if dt == "fp32" else 2500.0), 4),
                      flop_basis=f"{len(gl)} plain stride-1 3x3x3 layers at >= {16 if dt == 'fp32' else 32}^3 (SURVEY.md 8d per-layer GFLOP) x 32 sample passes")
        ent["largest_consumer"] = lc
    out["fp32" if dt == "fp32" else "16bit"] = ent
print(json.dumps(out, indent=1))
