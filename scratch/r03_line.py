"""One bench line -> one text line of its main figures (stdin: bench.py's JSON line)."""
import json
import sys

label = ' '.join(sys.argv[1:])
for raw in sys.stdin:
    if not raw.startswith('{'):
        continue
    d = json.loads(raw)
    r = d['roofline']
    pk = r.get('per_kernel', {})
    print('{0:28s} {1:8.1f} Mpx/s  {2:7.3f} ms/step  bpp {3:6.3f}  one-stream {4:7.3f} ms  gemm frac {5:6.4f}  cpu {6} ms  '
          'enc {7} dec {8} conv1 {9} tconv3 {10} latent {11}'.format(
              label, d['value'], d['ms_per_step'], d['rate_bpp'], d['one_stream_leg']['ms_per_step'], r['frac'], d['host_cpu_ms_per_step'],
              pk.get('coder_encode', {}).get('avg_ms'), pk.get('coder_decode', {}).get('avg_ms'), pk.get('conv1_gdn1', {}).get('avg_ms'),
              pk.get('tconv3', {}).get('avg_ms'), pk.get('latent', {}).get('avg_ms')))
