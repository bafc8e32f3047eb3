"""Top rows of a rocprofv3 --stats kernel_stats.csv found under a directory. usage: stats_table.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel s {tot / 1e9:.4f}")
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:80]
    print(f"{name:80s} calls {r['Calls']:>6s} tot {int(r['TotalDurationNs']) / 1e6:9.2f} ms avg {float(r['AverageNs']) / 1e3:9.1f} us {r['Percentage']:>6s}%")
