#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s26_waves_per_block_by_layer.log; : > $L
run() {
  name="$1"; shift
  for rep in 1 2 3; do
  out=$(env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-side --steps 100 --min-seconds 1.5 < /dev/null 2>/dev/null | tail -1)
  python - "$name" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    pk = d['roofline']['per_kernel']
    print('%-40s %8.1f Mpx/s  %6.3f ms/step  one-stream %6.3f ms  gemm frac %.4f  conv2 %.4f conv3 %.4f tconv1 %.4f tconv2 %.4f' % (name, d['value'], d['ms_per_step'], d['one_stream_leg']['ms_per_step'], d['roofline']['frac'], pk['conv2_gdn2']['avg_ms'], pk['conv3']['avg_ms'], pk['tconv1_igdn5']['avg_ms'], pk['tconv2_igdn6']['avg_ms']))
except Exception as e:
    print('%-40s failed: %s %s' % (name, e, raw[:200]))
PY
  done
}
run "four waves per block everywhere" EAE_HIP_SPLIT_WPB=4
run "conv_2 one wave per block (default)" A=1
run "four waves per block everywhere" EAE_HIP_SPLIT_WPB=4
run "conv_2 one wave per block (default)" A=1
cat $L
