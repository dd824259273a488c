#!/bin/bash
mkdir -p gpurun_out/r04
line() { python - "$1" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
pk = d['roofline']['per_kernel']
print('%.1f Mpx/s  ' % d['value'] + ' '.join('%s %.4f' % (k.split('_')[0] if not k.startswith('coder') else k[6:9], v['avg_ms']) for (k, v) in pk.items()))
PY
}
for i in 1 2 3 4; do
  timeout -k 10 300 python bench.py --steps 100 < /dev/null > /tmp/b$i.json 2>/dev/null; echo "run $i: $(line /tmp/b$i.json)"
done | tee gpurun_out/r04/s39_bench4.log
timeout -k 10 600 python -m pytest tests/test_gpu_latent.py tests/test_gpu_kernels.py tests/test_coder_device.py -m gpu -x -q < /dev/null 2>&1 | tail -2
