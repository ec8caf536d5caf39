"""print a rocprofv3 kernel_stats csv as ms per training step: python scripts/show_stats.py <csv> <steps incl. warm-up>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:100]:100s} calls/step={float(r['Calls'])/n:6.1f} ms/step={float(r['TotalDurationNs'])/n/1e6:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
print("sum of kernel time per step (ms):", tot / n / 1e6)
