#!/bin/bash
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ws -o runc -- python3 tools/kbench_deep.py 512 > gpurun_out/prof_ws.log 2>&1
python3 tools/ktrace.py gpurun_out/prof_ws 1 12
