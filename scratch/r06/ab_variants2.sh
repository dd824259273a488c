#!/bin/bash
F="--no-cpu-baseline --no-dropin-surface --no-single-image"
run() { env $2 python $3 $F $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1'.ljust(28), d['value'], d['ms_per_step'])"
}
for r in 1 2; do
  run old "A=1" scratch/r06/old_tree/bench.py ""
  run old_4streams "A=1" scratch/r06/old_tree/bench.py "--transform-streams 4"
  run new_4streams "A=1" bench.py "--transform-streams 4"
  run new_5streams "A=1" bench.py "--transform-streams 5"
  run new_6streams "A=1" bench.py "--transform-streams 6"
done
