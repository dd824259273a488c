#!/bin/bash
# r04: instruction-issue priorities of the GEMM waves against the coder's waves (s_setprio), product mode + one-stream leg
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s20_wave_priorities.log; : > $L
run() {
  name="$1"; lib="$2"
  for rep in 1 2; do
  out=$(EAE_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-side --steps 60 --min-seconds 1.0 < /dev/null 2>/dev/null | tail -1)
  python - "$name" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    pk = d['roofline']['per_kernel']
    print('%-44s %8.1f Mpx/s  %6.3f ms/step  one-stream %6.3f ms  gemm frac %.4f  conv2 %.4f tconv2 %.4f enc %.3f dec %.3f' % (name, d['value'], d['ms_per_step'], d['one_stream_leg']['ms_per_step'], d['roofline']['frac'], pk['conv2_gdn2']['avg_ms'], pk['tconv2_igdn6']['avg_ms'], pk['coder_encode']['avg_ms'], pk['coder_decode']['avg_ms']))
except Exception as e:
    print('%-44s failed: %s %s' % (name, e, raw[:200]))
PY
  done
}
run "shipped (coder 3, GEMM default 0)" autoencoder_based_image_compression_amd/lib/libeae_hip.so
run "coder 0, GEMM 0" scratch/r04/libs/coder_p0/libeae_hip.so
run "coder 0, GEMM 3" scratch/r04/libs/gemm_p3_coder_p0/libeae_hip.so
run "coder 3, GEMM 3" scratch/r04/libs/gemm_p3/libeae_hip.so
cat $L
