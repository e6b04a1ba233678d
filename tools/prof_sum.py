"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-step kernel time by kernel (tools/prof_sum.py <stats.csv> <steps+warmup>)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{r['Name'][:64]:64s} {float(r['TotalDurationNs'])/n/1e6:7.2f} ms/step  avg {float(r['AverageNs'])/1e3:7.1f} us  x{int(r['Calls'])/n:5.0f}")
print(f"TOTAL {tot/n/1e6:.2f} ms/step")
