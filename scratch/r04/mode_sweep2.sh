#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s31_mode_sweep2.log; : > $L
run() {
  name="$1"; shift
  out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-side "$@" < /dev/null 2>/dev/null | tail -1)
  python - "$name" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    print('%-58s %8.1f Mpx/s  %6.3f ms/step' % (name, d['value'], d['ms_per_step']))
except Exception as e:
    print('%-58s failed: %s' % (name, e))
PY
}
for rep in 1 2; do for ts in 3 4 5; do run "kodak x24, 100-step blocks, $ts transform streams" --steps 100 --min-seconds 1.5 --transform-streams $ts; done; done
for ts in 3 4 5; do run "kodak x24, 20-step blocks (driver), $ts transform streams" --steps 20 --warmup 5 --transform-streams $ts; done
for ts in 3 4; do run "64 x 256x256, $ts transform streams" --height 256 --width 256 --batch 64 --steps 100 --transform-streams $ts; done
for ts in 3 4; do run "2 x 2048x2048, $ts transform streams" --height 2048 --width 2048 --batch 2 --steps 50 --transform-streams $ts; done
for ts in 3 4; do run "kodak x24 at 2 bpp, $ts transform streams" --bin-width 0.05 --steps 60 --transform-streams $ts; done
cat $L
