#!/bin/bash
F="--no-cpu-baseline --no-dropin-surface"
for t in old new old new; do
  if [ $t = old ]; then B=scratch/r06/old_tree/bench.py; else B=bench.py; fi
  python $B $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
re=d.get('realistic_entropy')
print('$t', d['value'], d['ms_per_step'], json.dumps(re)[:900])"
done
