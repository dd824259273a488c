#!/bin/bash
# bench.py on the other shapes of BASELINE.json (one JSON line each -> gpurun_out/<tag>_other_shapes.jsonl)
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_other_shapes.jsonl
: > $OUT
run() { timeout 300 python bench.py --no-cpu-baseline --no-single-image "$@" 2>/dev/null | tail -1 >> $OUT; }
run --height 256 --width 256 --batch 64
run --height 2048 --width 2048 --batch 2
run --height 2048 --width 2048 --batch 2 --coder-streams 6 --transform-streams 2
run --height 2048 --width 2048 --batch 12
run --batch 48
run --batch 40
python - <<PY
import json
for line in open("$OUT"):
    d = json.loads(line)
    print(d["config"]["workload"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["per_launch_frac"])
PY
