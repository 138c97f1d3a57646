#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# kernel-trace + stats of a short bench run -> gpurun_out/prof_<tag>/ and a per-kernel summary on stdout
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe "$@" > gpurun_out/prof_$tag.log 2>&1
python3 tools/kstats.py gpurun_out/prof_$tag 3 60 > gpurun_out/prof_$tag.txt
head -50 gpurun_out/prof_$tag.txt
