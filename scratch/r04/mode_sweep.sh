#!/bin/bash
# r04: the product mode's two knobs re-swept on the round's final kernels (transform streams x batches of coder work in flight)
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s30_mode_sweep.log; : > $L
for ts in 2 3 4; do for cs in 4 5 6; do
  out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-side --steps 100 --min-seconds 1.5 --transform-streams $ts --coder-streams $cs < /dev/null 2>/dev/null | tail -1)
  python - "$ts transform streams, $cs in flight" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    print('%-40s %8.1f Mpx/s  %6.3f ms/step' % (name, d['value'], d['ms_per_step']))
except Exception as e:
    print('%-40s failed: %s' % (name, e))
PY
done; done
cat $L
