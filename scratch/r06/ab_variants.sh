#!/bin/bash
# headline leg of the working tree under variants of the environment / flags, interleaved with the round-5 tree, on one box
F="--no-cpu-baseline --no-dropin-surface --no-single-image"
run() { # label, env, bench, flags
  env $2 python $3 $F $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1'.ljust(28), d['value'], d['ms_per_step'])"
}
for r in 1 2; do
  run old "A=1" scratch/r06/old_tree/bench.py ""
  run new "A=1" bench.py ""
  run new_strips2 "EAE_HIP_T3_STRIPS=2" bench.py ""
  run new_strips3 "EAE_HIP_T3_STRIPS=3" bench.py ""
  run new_2streams "A=1" bench.py "--transform-streams 2"
  run new_4streams "A=1" bench.py "--transform-streams 4"
  run old_2streams "A=1" scratch/r06/old_tree/bench.py "--transform-streams 2"
done
