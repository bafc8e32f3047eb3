"""Per-epoch kernel table from the committed one-stream rocprofv3 --stats CSVs of epoch_profile.sh: (3-epoch run - 1-epoch run) / 2 per
kernel name, plus the HBM bytes by kernel name from <tag>_epoch_profile.json.  usage: top_kernels_table.py <tag> <dtype> [<dtype> ...]
(reads profiles/<tag>_epoch_<dtype>_{1,3}ep_kernel_stats.csv, profiles/<tag>_epoch_profile.json; prints the table)"""
import collections, csv, json, sys
tag, dts = sys.argv[1], sys.argv[2:]


def load(f):
    t = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        t[r["Name"]] = (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]))
    return t


prof = json.load(open(f"profiles/{tag}_epoch_profile.json"))
for dt in dts:
    a, b = load(f"profiles/{tag}_epoch_{dt}_1ep_kernel_stats.csv"), load(f"profiles/{tag}_epoch_{dt}_3ep_kernel_stats.csv")
    rows = sorted((((t3 - a.get(k, (0, 0))[0]) / 2, (c3 - a.get(k, (0, 0))[1]) / 2, k) for k, (t3, c3) in b.items()), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"== {dt}: one steady-state epoch = (3-epoch run - 1-epoch run) / 2 of rocprofv3 --kernel-trace --stats on ONE stream "
          f"(profiles/{tag}_epoch_{dt}_{{1,3}}ep_kernel_stats.csv): {tot:.2f} ms of kernels")
    for t, c, k in rows[:42]:
        n = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:86]
        print(f"{n:86s} launches {c:6.1f} {t:9.3f} ms  avg {t / max(c, 1e-9) * 1e3:8.1f} us {100 * t / tot:6.2f} %")
    fam = {"InstanceNorm apply / reduce / finalize": ("in_apply_vec", "chan_reduce_vec", "in_stats_finalize", "in_bwd_finalize"),
           "D-ring conv forward / data gradient": ("conv3_ring_kernel",), "weight-gradient ring sweep": ("conv3_wgrad_ring_kernel",), "weight gradient of the small planes (flat runs)": ("conv3_wgrad_flat_kernel",),
           "row-reuse conv": ("conv3_rows_kernel",), "generic MFMA conv": ("conv3_mfma_kernel",), "fused head + warp": ("head_warp_",),
           "MIND + GIN + noise + image warps": ("mind_", "gin_chain", "distribution_elementwise", "warp_fwd_kernel"),
           "stride-2 / transposed conv (forward, data and weight gradients)": ("conv_s2_regs", "convT_", "conv3_wgrad_tr_s2x", "conv3_wgrad_tr_kernel", "conv3_wgrad_tr8"),
           "loss": ("softdice_",), "slab reductions": ("wgrad_reduce", "pointwise_wgrad")}
    print("-- by family (ms per epoch):")
    for name, pats in fam.items():
        v = sum(t for t, c, k in rows if any(p in k for p in pats))
        print(f"   {name:70s} {v:8.2f}")
    e = prof.get("fp32" if dt == "fp32" else "16bit", {})
    if e.get("hbm_bytes_by_kernel"):
        print(f"-- HBM bytes of the same epoch by kernel name (rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE, separate passes): "
              f"{e['hbm_bytes_per_epoch'] / 1e9:.1f} GB in all (fetched {e['fetch_bytes_per_epoch'] / 1e9:.1f}, written {e['write_bytes_per_epoch'] / 1e9:.1f})")
        for h in e["hbm_bytes_by_kernel"]:
            print(f"{h['kernel'][:86]:86s} fetched {h['fetch_gb']:8.2f} GB  written {h['write_gb']:8.2f} GB")
    print()
