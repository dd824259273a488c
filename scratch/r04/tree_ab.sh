#!/bin/bash
# A/B of two whole trees on one box: round 3's final tree (git archive 4c17182 built under scratch/r04/libs/r03tree) against this one,
# per-kernel averages of the one-stream leg and the headline, alternating.   bash scratch/r04/tree_ab.sh <tag>
OUT=gpurun_out/r04/${1:-tree_ab}; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
pk = d['roofline']['per_kernel']
print('%.1f Mpx/s  ' % d['value'] + ' '.join('%s %.4f' % (k, v['avg_ms']) for (k, v) in pk.items()))
PY
}
for i in 1 2; do
  for t in r03 r04; do
    if [ $t = r03 ]; then dir=scratch/r04/libs/r03tree; else dir=.; fi
    (cd $dir && timeout -k 10 300 python bench.py --steps 100 < /dev/null > /tmp/ab_$t$i.json 2>/dev/null)
    cp /tmp/ab_$t$i.json $OUT/bench_$t$i.json
    echo "$t run $i: $(line /tmp/ab_$t$i.json)"
  done
done | tee $OUT/tree_ab.log
