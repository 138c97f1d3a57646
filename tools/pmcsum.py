import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|void ", "", r["Kernel_Name"])
    k = re.sub(r"\(.*$", "", k)[:52] + " grid " + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("WRITE_SIZE", [0]))):
    if any(s in k for s in ("convt", "gather", "direct", "head_", "stem_", "in_lrelu", "reduce2")):
        print("%-70s n=%3d  %s" % (k, len(next(iter(d.values()))), "  ".join("%s %.0f MB" % (c, sum(v) / len(v) * 1024 / 1e6 * (2 if c == "FETCH_SIZE" else 1)) for c, v in sorted(d.items()))))
