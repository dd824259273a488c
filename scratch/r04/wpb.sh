#!/bin/bash
# r04: waves per block of the split conv GEMM (the unit of dispatch) next to the coder: 4 (shipped), 2, 1
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s25_waves_per_block.log; : > $L
run() {
  name="$1"; lib="$2"; shift 2
  for rep in 1 2; do
  out=$(EAE_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-side --steps 60 --min-seconds 1.0 "$@" < /dev/null 2>/dev/null | tail -1)
  python - "$name" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    pk = d['roofline']['per_kernel']
    print('%-34s %8.1f Mpx/s  %6.3f ms/step  one-stream %6.3f ms  gemm frac %.4f  conv2 %.4f conv3 %.4f tconv1 %.4f tconv2 %.4f' % (name, d['value'], d['ms_per_step'], d['one_stream_leg']['ms_per_step'], d['roofline']['frac'], pk['conv2_gdn2']['avg_ms'], pk['conv3']['avg_ms'], pk['tconv1_igdn5']['avg_ms'], pk['tconv2_igdn6']['avg_ms']))
except Exception as e:
    print('%-34s failed: %s %s' % (name, e, raw[:200]))
PY
  done
}
run "4 waves per block (shipped)" autoencoder_based_image_compression_amd/lib/libeae_hip.so
run "2 waves per block" scratch/r04/libs/wpb2/libeae_hip.so
run "1 wave per block" scratch/r04/libs/wpb1/libeae_hip.so
run "4 per block, 2 bpp" autoencoder_based_image_compression_amd/lib/libeae_hip.so --bin-width 0.05
run "1 per block, 2 bpp" scratch/r04/libs/wpb1/libeae_hip.so --bin-width 0.05
cat $L
