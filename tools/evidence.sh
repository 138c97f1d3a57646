#!/bin/bash
# Evidence run of a round (GPU box, repo root): bench lines, kernel trace, PMC passes -> gpurun_out/<tag>_*
#   tools/evidence.sh r04        (then copy what is to be judged into profiles/)
set -o pipefail
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python3 bench.py --steps 20 --warmup 5 --launch eager --no-cpu-baseline --no-parity --no-torch-adam > gpurun_out/${tag}_bench_launch_eager.json 2>/dev/null
tools/prof.sh ${tag} --no-parity --no-torch-adam --launch eager > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/prof_${tag} 3 70 > gpurun_out/${tag}_kernel_stats.txt
python3 tools/ktrace.py gpurun_out/prof_${tag} 3 220 > gpurun_out/${tag}_kernel_trace_by_grid.txt
tools/pmc_step.sh ${tag} > /dev/null 2>&1
tools/pmc.sh ${tag} > gpurun_out/${tag}_pmc_counters.txt 2>&1
python3 tools/kbench_direct.py > gpurun_out/${tag}_direct_forms.txt 2>&1
python3 bench.py --features 30 --dtype fp16 --patch 160 160 80 --steps 10 --warmup 3 --no-cpu-baseline --no-torch-adam > gpurun_out/${tag}_bench_config4_fp16_f30_160x160x80.json 2>/dev/null
python3 bench.py --features 64 --pools 5 --patch 192 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-torch-adam --no-parity > gpurun_out/${tag}_bench_config5_192cubed_f64_p5.json 2>/dev/null
tools/prof_config4.sh > /dev/null 2>&1; cp gpurun_out/c4_trace.txt gpurun_out/${tag}_config4_kernel_trace_by_grid.txt
python3 tools/kbench_deep.py > gpurun_out/${tag}_deep_level_convs.txt 2>/dev/null
echo "--- RU3D_CONV_SK=0 RU3D_CONV_PC4=0 (round-3 kernels)" >> gpurun_out/${tag}_deep_level_convs.txt
RU3D_CONV_SK=0 RU3D_CONV_PC4=0 python3 tools/kbench_deep.py >> gpurun_out/${tag}_deep_level_convs.txt 2>/dev/null
# the captured step as a timeline (graph replays under the kernel trace)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${tag}_tl -o runc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-parity --no-torch-adam > /dev/null 2>&1
python3 tools/ktimeline.py gpurun_out/prof_${tag}_tl 3 30 > gpurun_out/${tag}_timeline_graph_replay.txt; rm -rf gpurun_out/prof_${tag}_tl
# N > 1 host path on the one GPU: two ranks on cuda:0, gradients over gloo (RCCL refuses two ranks on one device)
RU3D_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_two_ranks_one_device.json 2> gpurun_out/${tag}_bench_two_ranks_one_device.err
python3 tools/t_launch_floor.py > gpurun_out/${tag}_launch_floor_per_kernel.txt 2>/dev/null
rm -rf gpurun_out/prof_${tag}/*.db gpurun_out/prof_c4/*.db
echo done; tail -c 400 gpurun_out/${tag}_bench_default.json
