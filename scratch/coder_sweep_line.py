import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bw', d['config']['bin_width_multiplier'], 'coder streams', sys.argv[1], 'Mpx/s', d['value'], 'ms/step', d['ms_per_step'],
      'bpp', d['rate_bpp'], 'gemm frac', d['roofline']['frac'], d['roofline']['per_launch_ms'])
