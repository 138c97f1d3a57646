"""Whole-step HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the same bench command:
    python tools/pmc_step.py <fetch_dir> <write_dir> <steps incl. warm-up> <out.json>
gfx950 corrections of MI355X_MICROARCH.md: both counters are in KiB; FETCH_SIZE reports half the bytes of 16-byte-per-lane
streaming reads (x2), WRITE_SIZE is exact for 16-byte streaming stores."""
import collections, csv, glob, json, re, sys

def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|ru3d_f16::|void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*$", "", name)[:70]
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3])
rows = []
for k in sorted(set(fetch) | set(write)):
    fb = 2.0 * fetch.get(k, [0, 0.0])[1] * 1024.0
    wb = write.get(k, [0, 0.0])[1] * 1024.0
    rows.append((k, fetch.get(k, write.get(k))[0], fb, wb))
tot_f = sum(r[2] for r in rows) / steps
tot_w = sum(r[3] for r in rows) / steps
print("whole step: fetch %.2f GB + write %.2f GB = %.2f GB per step (FETCH_SIZE x2, KiB -> bytes; %g steps incl. warm-up)" %
      (tot_f / 1e9, tot_w / 1e9, (tot_f + tot_w) / 1e9, steps))
for k, n, fb, wb in sorted(rows, key=lambda r: -(r[2] + r[3])):
    print("%-70s launches/step %6.1f  fetch %8.1f MB  write %8.1f MB  per step" % (k, n / steps, fb / steps / 1e6, wb / steps / 1e6))
json.dump({"fetch_bytes_per_step": tot_f, "write_bytes_per_step": tot_w, "hbm_bytes_per_step": tot_f + tot_w,
           "correction": "FETCH_SIZE x2 (gfx950 16-B/lane streaming reads), WRITE_SIZE exact, KiB -> bytes",
           "steps_profiled": steps}, open(sys.argv[4], "w"), indent=1)
