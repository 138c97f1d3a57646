"""Summarise a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [steps]"""
import csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms (%.2f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-84s calls %5s total %9.2f ms avg %8.3f ms %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
