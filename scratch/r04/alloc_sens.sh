#!/bin/bash
# Which coder kernel's register allocation do the transforms feel? The shipped library (every coder kernel at its own allocation) against
# variants with ONE family at 64 registers (scratch/r04/build_alloc_variants.sh): per-kernel averages of the one-stream leg + headline.
OUT=gpurun_out/r04/${1:-alloc}; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
pk = d['roofline']['per_kernel']
print('%.1f Mpx/s  ' % d['value'] + ' '.join('%s %.4f' % (k.split('_')[0] if not k.startswith('coder') else k[6:9], v['avg_ms']) for (k, v) in pk.items()))
PY
}
for t in ${VARIANTS:-shipped binarise emit debinarise cores all64 shipped}; do
  if [ $t = shipped ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$PWD/scratch/r04/libs/alloc_$t/libeae_hip.so; fi
  timeout -k 10 300 python bench.py --steps 100 < /dev/null > /tmp/ab_$t.json 2>/dev/null
  printf "%-11s %s\n" $t "$(line /tmp/ab_$t.json)"
done | tee $OUT/alloc_sens.log
