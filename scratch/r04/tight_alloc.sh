#!/bin/bash
# A/B: the coder kernels' reserved register = v63 (allocation 64, shipped) against the last register of each kernel's own allocation
# (24 / 40 / 48; scratch/r04/libs/tight built with -DEAE_TIGHT_ALLOC): per-kernel averages of the one-stream leg + headline, alternating.
OUT=gpurun_out/r04/${1:-tight}; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
pk = d['roofline']['per_kernel']
print('%.1f Mpx/s  ' % d['value'] + ' '.join('%s %.4f' % (k, v['avg_ms']) for (k, v) in pk.items()))
PY
}
for i in 1 2; do
  for t in v63 tight; do
    if [ $t = tight ]; then export EAE_HIP_LIB=$PWD/scratch/r04/libs/tight/libeae_hip.so; else unset EAE_HIP_LIB; fi
    timeout -k 10 300 python bench.py --steps 100 < /dev/null > /tmp/ab_$t$i.json 2>/dev/null
    cp /tmp/ab_$t$i.json $OUT/bench_$t$i.json
    echo "$t run $i: $(line /tmp/ab_$t$i.json)"
  done
done | tee $OUT/tight_alloc.log
