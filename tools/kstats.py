"""Print a rocprofv3 kernel_stats.csv sorted by total time: python tools/kstats.py FILE [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:9.1f} "
          f"pct={r['Percentage']}")
