"""Per-dispatch rows behind bench.py's `roofline.avg_ms`: every launch of the kernel instantiation that
runs the forward of the two 128^3 32 -> 32 blocks (enc.0.1 and dec.3.1: conv3_ring_kernel<T16, NT = false, ABL = 0, GST = false,
KH = 1> - the data gradients of these layers run the GST instantiation, the 64-channel layers KH = 2, so this name is ONE shape at
training batch 8 plus the batch-1 launches of the evaluation pass) in the kernel trace of the 3-epoch bench run of
profiles/tools/epoch_profile.sh.  stdout: csv rows; stderr: the summary (mean / median of the full-grid launches)."""
import csv, glob, statistics, sys
tag = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/{tag}_ep_bf16_stats_3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3_ring_kernel<unsigned short, false, 0, false, 1>" in k:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")),
                         r.get("LDS_Block_Size", ""), r.get("Dispatch_Id", "")))
rows.sort()
# One-stream order of this instantiation inside a TTA epoch (4 network passes of 8 samples, then the evaluation pass of 1):
#   per training pass: forward enc.0.1, forward dec.3.1, data gradient of dec.3.0 (32 channels of dy in, 64 out: two channel
#   blocks, ~2.3x as long); then the two forward launches of the evaluation pass.  14 launches per epoch.
PATTERN = ["fwd_enc01_batch8", "fwd_dec31_batch8", "dgrad_dec30_batch8"] * 4 + ["fwd_enc01_eval_batch1", "fwd_dec31_eval_batch1"]
print("dispatch_id,epoch,start_ns,end_ns,duration_us,grid_size,workgroup_size,lds_bytes,class")
by = {}
for n, (s, e, g, w, l, i) in enumerate(rows):
    ep, cls = n // len(PATTERN), PATTERN[n % len(PATTERN)]
    by.setdefault((cls, ep), []).append((e - s) / 1e3)
    print(f"{i},{ep},{s},{e},{(e - s) / 1e3:.1f},{g},{w},{l},{cls}")
neps = (len(rows) + len(PATTERN) - 1) // len(PATTERN)
print(f"{len(rows)} launches of conv3_ring_kernel<bf16, NT=false, ABL=0, GST=false, KH=1> in {neps} epochs (one stream)", file=sys.stderr)
for cls in ("fwd_enc01_batch8", "fwd_dec31_batch8", "dgrad_dec30_batch8", "fwd_dec31_eval_batch1"):
    for ep in range(neps):
        v = by.get((cls, ep), [])
        if v:
            print(f"  epoch {ep} {cls}: n {len(v)}, mean {statistics.mean(v):.1f} us, min {min(v):.1f}, max {max(v):.1f}", file=sys.stderr)
steady = [x for ep in range(1, neps) for x in by.get(("fwd_dec31_batch8", ep), [])]
if steady:
    m = statistics.mean(steady)
    print(f"dec.3.1 forward, 8 samples per launch, epochs 1.. (the first epoch ramps the clock): mean {m:.1f} us over {len(steady)} launches "
          f"-> 927.7 GFLOP / {m:.1f} us = {927.7 / m * 1e3:.0f} TFLOP/s = {927.7 / m * 1e3 / 2500:.4f} of 2.5 PF   (bench.py's roofline.avg_ms "
          f"is the same population timed with events inside the timed region)", file=sys.stderr)
