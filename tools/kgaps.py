"""Idle time between kernels of a rocprofv3 --kernel-trace run: python tools/kgaps.py <dir> [steps]
Sorts the dispatches by start time, prints wall span, union-busy time, and the idle gaps attributed to the kernel that
FOLLOWS each gap (the one whose launch / dependency wait the gap is), so tiny dependent launches show their real cost:
duration + the bubble in front of them."""
import csv, glob, re, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|ru3d_f16::|void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)[:40]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# drop the first third (warm-up, allocator growth) - keep the dispatches of the last `steps` steps if the caller
# profiled only timed steps; otherwise everything
busy = 0; cur_end = rows[0][0]; gaps = {}; durs = {}; biggest = []
for s, e, n in rows:
    a = durs.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    if s > cur_end:
        g = (s - cur_end) / 1e3
        if g < 200:                      # host-side pauses between steps / syncs are not launch bubbles
            ga = gaps.setdefault(n, [0, 0.0]); ga[0] += 1; ga[1] += g
        else:
            biggest.append((g, n))
        busy += e - s; cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end; cur_end = e
span = (rows[-1][1] - rows[0][0]) / 1e6
gap_tot = sum(v[1] for v in gaps.values()) / 1e3
print("dispatches %d, span %.2f ms, union busy %.2f ms, bubbles < 200 us: %.2f ms (%.3f ms/step), long pauses: %d (%.2f ms)"
      % (len(rows), span, busy / 1e6, gap_tot, gap_tot / steps, len(biggest), sum(g for g, _ in biggest) / 1e3))
print("%-42s %8s %10s %10s %12s" % ("kernel (follows the bubble)", "calls/st", "dur us", "bubble us", "bubble ms/st"))
for n, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    dn = durs[n]
    print("%-42s %8.1f %10.1f %10.2f %12.3f" % (n, dn[0] / steps, dn[1] / dn[0], g / c, g / 1e3 / steps))
