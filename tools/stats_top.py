"""print the top rows of a rocprofv3 kernel_stats csv: python tools/stats_top.py <csv> [n_forwards]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    n = r["Name"].replace("(anonymous namespace)::", "")[:64]
    print(f"{n:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:7.1f} pct={r['Percentage']}")
print("sum of kernel time per forward (us):", round(tot / nf / 1e3, 1), " launches per forward:", sum(int(r["Calls"]) for r in rows) / nf)
