"""Timeline view of a rocprofv3 --kernel-trace run (graph replays or eager): python tools/ktimeline.py <dir> <steps> [gaps]
For the last <steps> steps (a step ends with the Adam kernel): wall time, union of busy time, idle time, time with two or
more kernels in flight, per-queue busy time, and the largest idle gaps with the kernels either side."""
import csv, glob, re, sys
d = sys.argv[1]; steps = int(sys.argv[2]); ngaps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|ru3d_f16::|void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*$", "", name)[:48]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith("adam_multi")]
if len(ends) < steps + 1:
    sys.exit("not enough steps in the trace (%d Adam launches)" % len(ends))
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
win = rows[lo:hi]
t0, t1 = win[0][0], max(r[1] for r in win)
ev = sorted([(r[0], 1) for r in win] + [(r[1], -1) for r in win])
busy = multi = 0; depth = 0; last = t0
for t, dlt in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += dlt; last = t
wall = t1 - t0
print("steps %d  kernels/step %.1f  wall %.3f ms/step  busy %.3f  idle %.3f  >=2 kernels in flight %.3f  sum of durations %.3f" % (
    steps, len(win) / steps, wall / 1e6 / steps, busy / 1e6 / steps, (wall - busy) / 1e6 / steps, multi / 1e6 / steps,
    sum(r[1] - r[0] for r in win) / 1e6 / steps))
short = [r for r in win if r[1] - r[0] < 12000]
print("kernels shorter than 12 us: %.1f per step, %.3f ms per step" % (len(short) / steps, sum(r[1] - r[0] for r in short) / 1e6 / steps))
q = {}
for r in win: q[r[3]] = q.get(r[3], 0) + r[1] - r[0]
print("busy per queue (ms/step):", {k: round(v / 1e6 / steps, 3) for k, v in q.items()})
# idle gaps
gaps = []; cur_end = win[0][1]; cur_name = "q%s %s" % (win[0][3], win[0][2])
for r in win[1:]:
    if r[0] > cur_end: gaps.append((r[0] - cur_end, cur_name, "q%s %s" % (r[3], r[2])))
    if r[1] > cur_end: cur_end, cur_name = r[1], "q%s %s" % (r[3], r[2])
hist = {}
for g, a, b in gaps:
    k = "<2us" if g < 2000 else ("2-5us" if g < 5000 else ("5-10us" if g < 10000 else ">=10us"))
    h = hist.setdefault(k, [0, 0]); h[0] += 1; h[1] += g
print("idle gaps per step:", {k: (round(v[0] / steps, 1), "%.3f ms" % (v[1] / 1e6 / steps)) for k, v in hist.items()})
for g, a, b in sorted(gaps, reverse=True)[:ngaps]:
    print("  %7.1f us  after %-52s before %s" % (g / 1e3, a, b))
if len(sys.argv) > 4:      # dump a slice of the last step: start index, count
    a0, n0 = int(sys.argv[4]), int(sys.argv[5])
    last = rows[ends[-2] + 1:ends[-1] + 1]
    for r in last[a0:a0 + n0]:
        print("  +%9.1f us  dur %7.1f  q%s  %s" % ((r[0] - last[0][0]) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
