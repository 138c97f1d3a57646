#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <tag>   -> PMC passes over `tools/kbench.py 3` (conv + wgrad 32->32 @ 2x128^3)
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 tools/kbench.py 3 > gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 tools/kbench.py 3 > gpurun_out/pmc_${tag}_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python3 tools/kbench.py 3 > gpurun_out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- python3 tools/kbench.py 3 > gpurun_out/pmc_${tag}_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/pmc_${tag}_traffic.json
python3 - <<PY
import csv, glob, collections
for part in "ab":
    f = glob.glob("gpurun_out/pmc_${tag}_%s/**/*counter_collection.csv" % part, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        import re
        k = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|void ", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k)[:48] + " grid " + r.get("Grid_Size", "?")
        if "conv3" in k or "wgrad3" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
