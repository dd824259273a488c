#!/bin/bash
# A/B on ONE box: the round-5 tree (scratch/r06/old_tree, built from commit 07efd11) against the working tree, interleaved, headline leg only
F="--no-cpu-baseline --no-dropin-surface --no-single-image"
for r in 1 2 3; do
  for t in old new; do
    if [ $t = old ]; then B=scratch/r06/old_tree/bench.py; else B=bench.py; fi
    python $B $F "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
print('$t', d['value'], d['ms_per_step'], 'gemm', d['roofline']['frac'], 'tconv3', pk['tconv3'].get('frac'), 'conv1', pk['conv1_gdn1'].get('frac'), 'one-stream ms', d.get('one_stream_leg',{}).get('ms_per_step'))"
  done
done
