#!/bin/bash
# kernel trace of the graph-replayed step -> idle-gap analysis
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gaps -o runc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-parity --no-torch-adam > gpurun_out/prof_gaps.log 2>&1
python3 tools/kgaps.py gpurun_out/prof_gaps 0.3 > gpurun_out/r3_gaps_graph.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gaps_e -o runc -- python3 bench.py --steps 4 --warmup 2 --launch eager --no-cpu-baseline --no-probe --no-parity --no-torch-adam > gpurun_out/prof_gaps_e.log 2>&1
python3 tools/kgaps.py gpurun_out/prof_gaps_e 0.3 > gpurun_out/r3_gaps_eager.txt
head -14 gpurun_out/r3_gaps_graph.txt; head -3 gpurun_out/r3_gaps_eager.txt
