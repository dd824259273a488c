"""Is the sys-heavy runtime thread there for ANY steady stream of launches? (minimal torch loop, optional events)"""
import os, subprocess, sys, time
code = r'''
import torch, time, sys
x = torch.randn(1 << 24, device="cuda")
mode = sys.argv[1]
t0 = time.time()
evs = []
while time.time() - t0 < 9:
    for _ in range(20):
        y = x * 2
    if mode == "events":
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    if mode == "sync":
        torch.cuda.current_stream().synchronize()
    if len(evs) > 1000: evs = evs[-10:]
    if mode != "sync":
        while torch.cuda.current_stream().query() is False and False: pass
        time.sleep(0.002)
'''
for mode in ('plain', 'events', 'sync'):
    p = subprocess.Popen([sys.executable, '-c', code, mode], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(4.)
    def snap():
        out = {}
        for tid in os.listdir('/proc/{}/task'.format(p.pid)):
            try:
                f = open('/proc/{}/task/{}/stat'.format(p.pid, tid)).read()
            except OSError:
                continue
            rest = f[f.rindex(')') + 2:].split()
            out[tid] = (int(rest[11]), int(rest[12]))
        return out
    a = snap(); time.sleep(3.); b = snap()
    hz = os.sysconf('SC_CLK_TCK')
    rows = sorted((((b[t][0] - a[t][0])/hz/3., (b[t][1] - a[t][1])/hz/3.) for t in b if t in a), key=lambda r: -(r[0] + r[1]))
    print(mode, [(round(u, 2), round(s, 2)) for (u, s) in rows[:3]])
    p.wait()
