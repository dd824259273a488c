#!/bin/bash
mkdir -p gpurun_out/r05h
for v in 0 1; do
  EAE_CODER_BEHIND_TCONV1=$v python bench.py --no-cpu-baseline --no-dropin-surface > gpurun_out/r05h/bench_behind_$v.json 2> gpurun_out/r05h/bench_behind_$v.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05h/bench_behind_$v.json') if l.startswith('{')][-1])
pk=d['roofline']['per_kernel']
print('behind tconv1 = $v', 'value', d['value'], 'one-stream', d['one_stream_leg']['value'], 'roofline', d['roofline']['frac'], {k:(v['avg_ms'], v.get('frac')) for k,v in pk.items() if 'conv' in k},
      'single', d['single_image']['latency_ms'], d['single_image']['ms_per_image'], 'other', [(o['workload'], o['value'], o['roofline']['frac']) for o in d['other_shapes']],
      'entropy', [(r['rate_bpp'], r['value'], r['step_over_no_coder_step']) for r in d['realistic_entropy']])
PY
done
