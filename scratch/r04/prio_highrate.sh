#!/bin/bash
# r04: wave priorities again, where the coder is what the step pays for: 2.04 and 3.22 bpp, 5 / 7 batches of coder work in flight
mkdir -p gpurun_out/r04
L=gpurun_out/r04/${1:-s46}_prio_highrate.log; : > $L
for bw in 0.05 0.0125; do
for cs in 5 7; do
for v in shipped gemm_p3 gemm_p3_coder_p0 gemm_p2_coder_p1; do
  if [ $v = shipped ]; then unset EAE_HIP_LIB; else export EAE_HIP_LIB=$PWD/scratch/r04/libs/$v/libeae_hip.so; fi
  out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-side --steps 40 --min-seconds 0.6 --bin-width $bw --coder-streams $cs < /dev/null 2>/dev/null | tail -1)
  python - "$v bw $bw in flight $cs" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    print('%-44s %8.1f Mpx/s  %6.3f ms/step  %.3f bpp' % (name, d['value'], d['ms_per_step'], d['rate_bpp']))
except Exception as e:
    print('%-44s failed: %s %s' % (name, e, raw[:200]))
PY
done; done; done
cat $L
