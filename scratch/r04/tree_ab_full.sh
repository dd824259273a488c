#!/bin/bash
# the side legs of the default bench line, round 3's final tree against this one on one box (see tree_ab.sh)
OUT=gpurun_out/r04/${1:-s43}; mkdir -p $OUT
for i in 1 2; do
for t in r03 r04; do
  if [ $t = r03 ]; then dir=scratch/r04/libs/r03tree; else dir=.; fi
  (cd $dir && timeout -k 10 400 python bench.py --no-cpu-baseline < /dev/null > /tmp/abf_$t$i.json 2>/dev/null)
  cp /tmp/abf_$t$i.json $OUT/bench_$t$i.json
  python - /tmp/abf_$t$i.json $t$i <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
print(sys.argv[2], '%.1f' % d['value'], 'single %.4f ms, one at a time %.4f ms' % (d['single_image']['ms_per_image'], d['single_image']['latency_ms']),
      'other shapes', [round(e['value'], 1) for e in d['other_shapes']], 'pcie %.1f' % d['pcie_inclusive']['value'],
      'entropy', [(round(e['value'], 1), e['step_over_no_coder_step']) for e in d['realistic_entropy']])
PY
done
done | tee $OUT/tree_ab_full.log
