"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run: python tools/kgaps.py <dir> [tail_fraction]
Takes the last <tail_fraction> of the trace (the timed steps), sorts by start, reports busy / idle time and the gap
histogram, and which kernels the largest gaps precede."""
import csv, glob, re, sys
d = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|ru3d_f16::|void ", "", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", name)[:50]))
rows.sort()
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps.append((s1 - e0, n0, n1))
idle = sum(max(g, 0) for g, _, _ in gaps)
over = sum(-min(g, 0) for g, _, _ in gaps)
print("kernels %d  span %.3f ms  busy %.3f ms  idle %.3f ms (%.1f%%)  overlap %.3f ms" % (len(rows), span / 1e6, busy / 1e6, idle / 1e6, 100.0 * idle / span, over / 1e6))
edges = [0, 500, 1000, 2000, 3000, 5000, 10000, 50000, 10**9]
for lo, hi in zip(edges, edges[1:]):
    sel = [g for g, _, _ in gaps if lo <= g < hi]
    print("  gap %6d..%-9d ns: %5d gaps, %.3f ms" % (lo, hi, len(sel), sum(sel) / 1e6))
agg = {}
for g, n0, n1 in gaps:
    a = agg.setdefault((n0, n1), [0, 0]); a[0] += 1; a[1] += max(g, 0)
print("largest idle by (previous -> next):")
for (n0, n1), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-50s -> %-50s n %4d  avg %7.2f us  total %.3f ms" % (n0, n1, c, t / c / 1e3, t / 1e6))
