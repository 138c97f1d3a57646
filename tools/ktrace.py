"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace run: python tools/ktrace.py <dir> [steps] [rows]"""
import csv, glob, re, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0; nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 80
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|ru3d_f16::|void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*$", "", name)[:60]
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    wg = r.get("Workgroup_Size_X", "?")
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = (name, grid, wg)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += t
tot = sum(a[1] for a in agg.values())
print("total %.2f ms/step over %g steps" % (tot / 1e3 / steps, steps))
for (name, grid, wg), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:nrows]:
    print("%-60s grid %-18s wg %-4s calls/step %5.1f  us/call %8.1f  ms/step %6.3f  %4.1f%%" % (name, grid, wg, n / steps, t / n, t / 1e3 / steps, 100 * t / tot))
