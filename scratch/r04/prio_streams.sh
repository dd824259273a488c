#!/bin/bash
# r04: do hardware-queue priorities change what the coder costs the transforms? (product mode, no side legs)
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s19_stream_priorities.log; : > $L
run() {
  name="$1"; shift
  for rep in 1 2; do
  out=$(env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-side --steps 60 --min-seconds 1.0 < /dev/null 2>/dev/null | tail -1)
  python - "$name" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    print('%-56s %8.1f Mpx/s  %6.3f ms/step  gemm frac (one-stream leg) %.4f' % (name, d['value'], d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('%-56s failed: %s %s' % (name, e, raw[:200]))
PY
  done
}
run "default priorities" A=1
run "transform streams high (-1), coder normal (0)" EAE_TRANSFORM_STREAM_PRIORITY=-1 EAE_CODER_STREAM_PRIORITY=0
run "transform streams normal, coder high" EAE_TRANSFORM_STREAM_PRIORITY=0 EAE_CODER_STREAM_PRIORITY=-1

cat $L
