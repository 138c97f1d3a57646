"""Parse two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `tools/kbench.py 1` into a per-launch HBM
traffic figure for the roofline kernel, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reads 1/2
of the bytes of a 16-B/lane streaming read; WRITE_SIZE is exact; both are in KiB).
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys

def mean_counter(d, counter, kernel_subs):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in kernel_subs)]
    return sum(vals) / max(len(vals), 1), len(vals)

out = {}
# kbench.py 3 runs three shapes; the 32->32 one is the only user of the <false, false> sliding instantiation with a
# 65536-thread grid, the averages over the other kernels mix shapes and are reported as such
for name, sub in (("conv3d k3 s1 32->32 on 2x128^3", ("conv3_s1_slide32_kernel<false, false, false, false>",)),
                  ("conv3d k3 s1 64->32 on 2x128^3 (sliding 64-channel kernel, Cout = 32 form)", ("conv3_s1_slide64_kernel<false, false, 2, false>",)),
                  ("conv3d k3 s1 64->64 on 2x64^3 (sliding 64-channel kernel)", ("conv3_s1_slide64_kernel<false, false, 1, false>",)),
                  ("wgrad k3 s1 sliding kernel (32->32, 64->32 @128^3, 64->64 @64^3 mixed)", ("wgrad3_s1_slide_kernel",))):
    fetch, n1 = mean_counter(sys.argv[1], "FETCH_SIZE", sub)
    write, n2 = mean_counter(sys.argv[2], "WRITE_SIZE", sub)
    out[name] = {"fetch_size_kib_raw": fetch, "write_size_kib": write, "launches": [n1, n2],
                 "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                 "correction": "FETCH_SIZE x2 (gfx950 16-B/lane streaming reads), WRITE_SIZE exact, KiB -> bytes"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
