#!/bin/bash
# round-3 evidence run (GPU box, repo root): bench lines, kernel trace, PMC passes -> gpurun_out/r3f_*
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3f_bench_default.json 2> gpurun_out/r3f_bench_default.err
python3 bench.py --steps 20 --warmup 5 --launch eager --no-cpu-baseline --no-parity --no-torch-adam > gpurun_out/r3f_bench_eager.json 2>/dev/null
tools/prof.sh r3f --no-parity --no-torch-adam --launch eager > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/prof_r3f 3 70 > gpurun_out/r3f_kernel_stats.txt
python3 tools/ktrace.py gpurun_out/prof_r3f 3 200 > gpurun_out/r3f_kernel_trace_by_grid.txt
tools/pmc_step.sh r3f > /dev/null 2>&1
tools/pmc.sh r3f > gpurun_out/r3f_pmc_counters.txt 2>&1
for d in 0 1; do echo "RU3D_CONV_S2=$d"; RU3D_CONV_S2=$d RU3D_FUSED_SKIP=$d RU3D_DGRAD_PAIR=$d python3 tools/kbench_direct.py; done > gpurun_out/r3f_direct_forms.txt 2>&1
python3 bench.py --features 30 --dtype fp16 --patch 160 160 80 --steps 10 --warmup 3 --no-cpu-baseline --no-torch-adam > gpurun_out/r3f_bench_config4.json 2>/dev/null
python3 bench.py --features 64 --pools 5 --patch 192 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-torch-adam --no-parity > gpurun_out/r3f_bench_config5.json 2>/dev/null
tools/r3_c4.sh > /dev/null 2>&1; cp gpurun_out/c4_trace.txt gpurun_out/r3f_config4_kernel_trace_by_grid.txt
python3 tools/kbench_deep.py > gpurun_out/r3f_deep_level_convs.txt 2>/dev/null
python3 tools/t_launch_floor.py > gpurun_out/r3f_launch_floor.txt 2>/dev/null
echo done; tail -c 400 gpurun_out/r3f_bench_default.json
