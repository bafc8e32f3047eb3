"""Idle gaps of the GPU in a rocprofv3 kernel trace: python3 profiles/tools/gaps.py <kernel_trace.csv> [epochs]
Prints the union-busy time, the idle time in gaps <= 1 ms (between epochs the host reads results back: longer gaps are the
bench's own bookkeeping) and the kernel pairs around the gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nep = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]) for r in rows)
ce, last, busy, cs = ev[0][1], ev[0][2], 0, ev[0][0]
gaps = []
for s, e, n in ev[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, last, n))
        cs, ce, last = s, e, n
    elif e > ce:
        ce, last = e, n
busy += ce - cs
small = [g for g in gaps if g[0] <= 1e6]
print(f"busy {busy/1e6/nep:.2f} ms per epoch; {len(small)/nep:.0f} gaps <= 1 ms per epoch, {sum(g[0] for g in small)/1e6/nep:.2f} ms per epoch; "
      f"{len(gaps)-len(small)} longer gaps, {sum(g[0] for g in gaps if g[0] > 1e6)/1e6:.1f} ms in all")
t, c = collections.Counter(), collections.Counter()
for d, a, b in small:
    t[(a, b)] += d
    c[(a, b)] += 1
for k, v in t.most_common(12):
    print(f"{v/1e3/nep:8.1f} us per epoch {c[k]/nep:6.1f} x  {k[0]}  ->  {k[1]}")
