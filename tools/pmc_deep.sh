#!/bin/bash
# usage (GPU box, repo root): tools/pmc_deep.sh <tag>  -> where the deep-level kernels' bytes come from: L2 hit / miss,
# memory-side fetch, vector-memory and MFMA busy cycles of the 3x3x3 convs / weight gradients at 32^3, 16^3, 8^3
# (tools/kbench_deep.py).  Counters in passes of their own (no trace domains beside --pmc).
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmcdeep_${tag}_l2 -- python3 tools/kbench_deep.py > gpurun_out/pmcdeep_${tag}_l2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcdeep_${tag}_fetch -- python3 tools/kbench_deep.py > gpurun_out/pmcdeep_${tag}_fetch.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmcdeep_${tag}_sq -- python3 tools/kbench_deep.py > gpurun_out/pmcdeep_${tag}_sq.log 2>&1
python3 - <<PY
import csv, glob, collections, re
out = []
for part in ("l2", "fetch", "sq"):
    fs = glob.glob("gpurun_out/pmcdeep_${tag}_%s/**/*counter_collection.csv" % part, recursive=True)
    if not fs:
        out.append("(no counter file for pass %s)" % part); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|void ", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k)[:52] + " grid " + r.get("Grid_Size", "?")
        if "conv3" in k or "wgrad3" in k or "ksplit" in k or "wgrad_reduce" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append("== pass %s" % part)
    for k, d in sorted(agg.items()):
        out.append(k)
        for c, v in sorted(d.items()):
            out.append("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            h, m = sum(d["TCC_HIT_sum"]) / len(d["TCC_HIT_sum"]), sum(d["TCC_MISS_sum"]) / len(d["TCC_MISS_sum"])
            out.append("   L2 hit rate                  %16.3f" % (h / max(h + m, 1.0)))
        if "FETCH_SIZE" in d:
            out.append("   HBM-side fetch (x2, KiB->B)  %16.0f bytes" % (2 * 1024 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])))
open("gpurun_out/pmcdeep_${tag}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
PY
