#!/bin/bash
# round-3 baseline on one box: eager, graph replay, graph replay with the weight gradients on a side stream
set -o pipefail
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-torch-adam"
$B > gpurun_out/r3_eager.json 2> gpurun_out/r3_eager.err && tail -c 600 gpurun_out/r3_eager.json | head -c 10 >/dev/null
$B --launch graph > gpurun_out/r3_graph.json 2> gpurun_out/r3_graph.err
RU3D_WGRAD_STREAM=1 $B --launch graph > gpurun_out/r3_graph_side.json 2> gpurun_out/r3_graph_side.err
python3 - <<'PY'
import json
for t in ("eager", "graph", "graph_side"):
    try:
        d = json.loads(open("gpurun_out/r3_%s.json" % t).read().strip().splitlines()[-1])
        print(t, "ms/step %.3f host %.3f eager %s" % (d["ms_per_step"], d["host_enqueue_ms_per_step"], d.get("ms_per_step_eager")))
    except Exception as e:
        print(t, "failed", e, open("gpurun_out/r3_%s.err" % t).read()[-400:])
PY
