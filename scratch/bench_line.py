"""Runs bench.py with the given flags under the current environment (EAE_HIP_LIB honoured) and prints the main figures."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
d = json.loads(line)
r = d['roofline']
print('%.1f Mpx/s  %.4f ms/step  frac %.4f  %s' % (d['value'], d['ms_per_step'], r['frac'], ' '.join('%s %.4f' % (k.split('_')[0], v) for (k, v) in r['per_launch_ms'].items())))
