#!/bin/bash
# usage (GPU box, repo root): tools/pmc_step.sh <tag> [bench args]   -> whole-step HBM traffic from two PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950), per kernel and in total, with the guide's corrections
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-torch-adam --no-probe --launch eager $@"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcstep_${tag}_fetch -- python3 $B > gpurun_out/pmcstep_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcstep_${tag}_write -- python3 $B > gpurun_out/pmcstep_${tag}_write.log 2>&1
python3 tools/pmc_step.py gpurun_out/pmcstep_${tag}_fetch gpurun_out/pmcstep_${tag}_write 3 gpurun_out/pmcstep_${tag}.json > gpurun_out/pmcstep_${tag}.txt
head -40 gpurun_out/pmcstep_${tag}.txt
