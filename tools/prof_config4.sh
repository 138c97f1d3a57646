#!/bin/bash
# config-4 kernel trace (F = 30 fp16, 2x160x160x80) -> gpurun_out/c4_trace.txt
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o runc -- python3 bench.py --features 30 --dtype fp16 --patch 160 160 80 --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-parity --no-torch-adam --launch eager > gpurun_out/prof_c4.log 2>&1
python3 tools/ktrace.py gpurun_out/prof_c4 3 70 > gpurun_out/c4_trace.txt
head -60 gpurun_out/c4_trace.txt
