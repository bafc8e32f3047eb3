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
print("dispatch_id,start_ns,end_ns,duration_us,grid_size,workgroup_size,lds_bytes,class")
full = []
gmax = max((int(r[2]) for r in rows if str(r[2]).isdigit()), default=0)
for s, e, g, w, l, i in rows:
    cls = "train_batch8" if str(g).isdigit() and int(g) == gmax else "eval_batch1"
    if cls == "train_batch8":
        full.append((e - s) / 1e3)
    print(f"{i},{s},{e},{(e - s) / 1e3:.1f},{g},{w},{l},{cls}")
if full:
    print(f"{len(full)} full-grid launches (8 samples, 128^3, 32 -> 32, forward with statistics): mean {statistics.mean(full):.1f} us, median "
          f"{statistics.median(full):.1f} us, min {min(full):.1f}, max {max(full):.1f}; FLOP per launch 927.7 G -> "
          f"{927.7 / statistics.mean(full) * 1e3 / 1e3:.1f} TFLOP/s mean = {927.7 / statistics.mean(full) / 2500 * 1e3:.4f} of 2.5 PF",
          file=sys.stderr)
