"""Per-dispatch rows behind bench.py's `roofline.forward`: every launch of the kernel instantiation
that runs the forward of the two 128^3 32 -> 32 blocks (enc.0.1 and dec.3.1: conv3_ring_kernel<T16, NT = false, ABL = 0, GST = false,
KH = 1>) in the ONE-STREAM kernel trace of the 3-epoch bench run of profiles/tools/epoch_profile.sh.  Since round 5's split concat
gradient the same instantiation also runs the two halves of dec.3.0's data gradient, so one epoch holds 18 launches:
4 training passes x (forward enc.0.1, forward dec.3.1, data gradient half `up`, data gradient half `skip`), all with 8 samples,
and the evaluation pass's two forward launches with ONE sample (the persistent grid is the same size, so the grid does not tell
them apart).  Rows are classified by their position in that sequence AND checked against their class: a row shorter than 0.3 x the
class median is re-labelled `batch1_or_misplaced` and kept out of every mean (round 5's summary averaged six ~101-us evaluation
launches into the batch-8 class).  stdout: csv rows; stderr: the summary.  usage: dec31_dispatches.py <tag> [dtype=fp16]"""
import csv, glob, statistics, sys
tag, dt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "fp16")
rows, wrows = [], []
for f in glob.glob(f"gpurun_out/{tag}_ep_{dt}_stats_3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        rec = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "")),
               r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("LDS_Block_Size", ""), r.get("Dispatch_Id", ""))
        if "conv3_ring_kernel<" in k and ", false, 0, false, 1>" in k:
            rows.append(rec)
        elif "conv3_wgrad_ring_kernel<" in k and ", false, false, false, true>" in k:      # (the one-tap form: every layer but the first)
            wrows.append(rec)
rows.sort()
wrows.sort()
PASS = ["fwd_enc01_batch8", "fwd_dec31_batch8", "dgrad_dec30_up_half_batch8", "dgrad_dec30_skip_half_batch8"]
# one-stream order: the two forward launches of a pass, later (in its backward) the two data-gradient halves; after the 4 passes
# the evaluation forward
PATTERN = (PASS[:2] + PASS[2:]) * 4 + ["fwd_enc01_eval_batch1", "fwd_dec31_eval_batch1"]
cls_of = [PATTERN[n % len(PATTERN)] for n in range(len(rows))]
dur = [(e - s) / 1e3 for s, e, *_ in rows]
med = {}
for c in set(cls_of):
    v = [d for d, cc in zip(dur, cls_of) if cc == c]
    med[c] = statistics.median(v) if v else 0.0
print("dispatch_id,epoch,start_ns,end_ns,duration_us,grid_size,workgroup_size,lds_bytes,class")
by = {}
for n, ((s, e, g, w, l, i), c, d) in enumerate(zip(rows, cls_of, dur)):
    ep = n // len(PATTERN)
    if c.endswith("batch8") and d < 0.3 * med[c]:
        c = "batch1_or_misplaced"
    by.setdefault((c, ep), []).append(d)
    print(f"{i},{ep},{s},{e},{d:.1f},{g},{w},{l},{c}")
neps = (len(rows) + len(PATTERN) - 1) // len(PATTERN)
print(f"{len(rows)} launches of conv3_ring_kernel<{dt}, NT=false, ABL=0, GST=false, KH=1> in {neps} epochs of {len(PATTERN)} (one stream)", file=sys.stderr)
for c in PASS + ["fwd_dec31_eval_batch1", "batch1_or_misplaced"]:
    for ep in range(neps):
        v = by.get((c, ep), [])
        if v:
            print(f"  epoch {ep} {c}: n {len(v)}, mean {statistics.mean(v):.1f} us, min {min(v):.1f}, max {max(v):.1f}", file=sys.stderr)
steady = [x for ep in range(1, neps) for x in by.get(("fwd_dec31_batch8", ep), [])]
if steady:
    m = statistics.mean(steady)
    print(f"dec.3.1 forward, 8 samples per launch, epochs 1.. (the first epoch ramps the clock): mean {m:.1f} us over {len(steady)} launches "
          f"-> 927.7 GFLOP / {m:.1f} us = {927.7 / m * 1e3:.0f} TFLOP/s = {927.7 / m * 1e3 / 2500:.4f} of 2.5 PF (profiler on, one stream; "
          f"bench.py's roofline.forward is the same population timed with events, profiler off)", file=sys.stderr)

# ---- the weight-gradient sweep behind `roofline` (top level): PER_PASS launches per training pass in backward order (12 until the
# 16^3 level's three layers moved to conv3_wgrad_flat_kernel, 9 since), the FIRST of each pass is block dec.3.1 (128^3, 32 -> 32,
# 8 samples) - the shape bench.py probes; its slab reduction (wgrad_reduce_kernel<8>, ~20 us) is a launch of its own here and
# inside the probe's event pair there
PER_PASS = len(wrows) // (4 * neps) if neps and len(wrows) % (4 * neps) == 0 else 12
wd = [(e - s) / 1e3 for s, e, *_ in wrows]
print("dispatch_id,epoch,start_ns,end_ns,duration_us,grid_size,workgroup_size,lds_bytes,class")
first = []
for n, ((s_, e_, g, w, l, i), d) in enumerate(zip(wrows, wd)):
    c = "wgrad_dec31_batch8" if n % PER_PASS == 0 else f"wgrad_other_layer_{n % PER_PASS}"
    ep = n // (4 * PER_PASS)
    if n % PER_PASS == 0:
        first.append((ep, d))
    print(f"{i},{ep},{s_},{e_},{d:.1f},{g},{w},{l},{c}")
print(f"{len(wrows)} launches of conv3_wgrad_ring_kernel<{dt}, one tap per MFMA, operand reuse> ({PER_PASS} per training pass, 4 passes per epoch)", file=sys.stderr)
steady = [d for ep, d in first if ep >= 1]
if steady and len(wrows) % PER_PASS == 0:
    m = statistics.mean(steady)
    print(f"dec.3.1 weight gradient (first sweep of each backward), 8 samples per launch, epochs 1..: mean {m:.1f} us over {len(steady)} launches, min {min(steady):.1f}, "
          f"max {max(steady):.1f} -> 927.7 GFLOP / {m:.1f} us = {927.7 / m * 1e3:.0f} TFLOP/s = {927.7 / m * 1e3 / 2500:.4f} of 2.5 PF (profiler on, one stream, "
          f"the sweep without its ~20-us slab reduction; bench.py's roofline is the same launch + reduction timed with events, profiler off)", file=sys.stderr)
elif wrows:
    print(f"(weight-gradient launches not a multiple of {PER_PASS}: no per-layer classification)", file=sys.stderr)
