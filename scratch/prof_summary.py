import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(f'{d}/*/*kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel time s:", tot / 1e9)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:58]
    print(f"{n:58s} calls {r['Calls']:>6s} tot {int(r['TotalDurationNs'])/1e6:8.1f} ms avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%")
