"""Per-thread CPU time of a running bench.py (which threads of a rank burn host CPU: 8 ranks share a 16-CPU quota)."""
import os, subprocess, sys, time
p = subprocess.Popen([sys.executable, 'bench.py', '--no-cpu-baseline', '--no-single-image', '--steps', '3000', '--warmup', '10'] + sys.argv[1:],
                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
time.sleep(6.)
def snap():
    out = {}
    for tid in os.listdir('/proc/{}/task'.format(p.pid)):
        try:
            f = open('/proc/{}/task/{}/stat'.format(p.pid, tid)).read()
        except OSError:
            continue
        comm = f[f.index('(') + 1:f.rindex(')')]
        rest = f[f.rindex(')') + 2:].split()
        out[tid] = (comm, int(rest[11]), int(rest[12]))
    return out
a = snap(); time.sleep(3.); b = snap()
hz = os.sysconf('SC_CLK_TCK')
rows = sorted(((b[t][1] - a[t][1] + b[t][2] - a[t][2])/hz/3., b[t][0], t, (b[t][1] - a[t][1])/hz/3., (b[t][2] - a[t][2])/hz/3.) for t in b if t in a)
for r in rows[::-1][:10]:
    print('%5.2f cpu  %-18s tid %s  user %.2f sys %.2f' % r)
print('total', round(sum(r[0] for r in rows), 2))
p.wait()
