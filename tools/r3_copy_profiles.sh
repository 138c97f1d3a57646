#!/bin/bash
# copy the evidence of tools/r3_final.sh (gpurun_out/r3f_*, scratch) into profiles/ (tracked): run here, from the repo root
set -e
G=gpurun_out; P=profiles
cp $G/r3f_bench_default.json $P/r03_bench_default.json
cp $G/r3f_bench_eager.json $P/r03_bench_launch_eager.json
cp $G/r3f_bench_config4.json $P/r03_bench_config4_fp16_f30_160x160x80.json
cp $G/r3f_bench_config5.json $P/r03_bench_config5_192cubed_f64_p5.json
cp $G/r3f_kernel_stats.txt $P/r03_kernel_stats.txt
cp $G/r3f_kernel_trace_by_grid.txt $P/r03_kernel_trace_by_grid.txt
cp $G/pmcstep_r3f.json $P/r03_pmc_step_traffic.json
cp $G/pmcstep_r3f.txt $P/r03_pmc_step_traffic_by_kernel.txt
cp $G/r3f_pmc_counters.txt $P/r03_pmc_counters.txt
cp $G/r3f_direct_forms.txt $P/r03_direct_forms_old_vs_new.txt
cp $G/r3f_config4_kernel_trace_by_grid.txt $P/r03_config4_kernel_trace_by_grid.txt
cp $G/r3f_deep_level_convs.txt $P/r03_deep_level_convs.txt
cp $G/r3f_launch_floor.txt $P/r03_launch_floor_per_kernel.txt
python3 tools/pmc_traffic.py $G/pmc_r3f_fetch $G/pmc_r3f_write $P/r03_pmc_traffic.json > /dev/null
python3 - <<'PY'
import json
for n in ("default", "launch_eager", "config4_fp16_f30_160x160x80", "config5_192cubed_f64_p5"):
    d = json.loads(open("profiles/r03_bench_%s.json" % n).read().strip().splitlines()[-1])
    print("%-32s %.2f ms  %.1f M voxels/s  frac_mfma %.4f  roofline %s" % (n, d["ms_per_step"], d["value"] / 1e6, d["roofline_step"]["frac_mfma"], (d.get("roofline") or {}).get("frac")))
PY
