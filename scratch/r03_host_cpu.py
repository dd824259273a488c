"""Host CPU per step by thread, product mode (8 ranks share a 16-CPU quota on an 8-GPU node): per-thread CPU seconds of this
process between the two barriers of one timed block of bench.run_pipeline. Usage: python scratch/r03_host_cpu.py [steps [graphs 0/1 [transform streams]]]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
GRAPHS = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
TSTREAMS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
import bench
import torch


def snap():
    out = {}
    for tid in os.listdir('/proc/self/task'):
        try:
            f = open('/proc/self/task/{}/stat'.format(tid)).read()
        except OSError:
            continue
        rest = f[f.rindex(')') + 2:].split()
        out[int(tid)] = (int(rest[11]), int(rest[12]))
    return out


SAMPLES = {}


def sampler(stop):
    # what the non-Python threads are doing: current system call (number + first arguments) and kernel wait channel
    import time
    mine = {t.native_id for t in threading.enumerate()}
    while not stop.is_set():
        for tid in os.listdir('/proc/self/task'):
            if int(tid) in mine:
                continue
            try:
                sc = open('/proc/self/task/{}/syscall'.format(tid)).read().split()
                wc = open('/proc/self/task/{}/wchan'.format(tid)).read().strip()
            except OSError:
                continue
            key = (int(tid), ' '.join(sc[:3]), wc)
            SAMPLES[key] = SAMPLES.get(key, 0) + 1
        time.sleep(0.002)


class Ctx(bench.Context):
    snaps = []
    stop = threading.Event()

    def barrier(self):
        if not self.snaps:
            threading.Thread(target=sampler, args=(self.stop,), daemon=True, name='sampler').start()
        else:
            self.stop.set()
        names = {t.native_id: t.name for t in threading.enumerate()}
        for tid in os.listdir('/proc/self/task'):
            try:
                names.setdefault(int(tid), 'runtime thread "%s"' % open('/proc/self/task/{}/comm'.format(tid)).read().strip())
            except OSError:
                pass
        self.snaps.append((snap(), names))


args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
ctx = Ctx(args, device, 1, 0, bench.usable_cpus())
variables = bench.synthetic_model(1.)
BATCH = int(os.environ.get('BATCH', '24'))      # BATCH=1 CODER_STREAMS=8 ... 3000 1 6: the pipelined single-image leg of bench.py
(H, W) = (int(os.environ.get('H', '512')), int(os.environ.get('W', '768')))
run = bench.run_pipeline(ctx, BATCH, STEPS, 10, variables, H, W, coder_streams=int(os.environ.get('CODER_STREAMS', '0')) or bench.auto_coder_streams(H, W),
                         transform_streams=TSTREAMS, use_graphs=GRAPHS, one_stream_steps=os.environ.get('ONE', '0') == '1')
(a, _), (b, names) = ctx.snaps[0], ctx.snaps[1]
hz = os.sysconf('SC_CLK_TCK')
sec = run['elapsed']
print('graphs', GRAPHS, 'transform streams', TSTREAMS, {k: v for (k, v) in os.environ.items() if k.startswith(('HSA_', 'ROC_', 'AMD_', 'GPU_', 'EAE_', 'DEBUG_'))})
print('block %.3f s, %.3f ms/step' % (sec, sec/STEPS*1e3))
rows = sorted(((b[t][0] - a.get(t, (0, 0))[0] + b[t][1] - a.get(t, (0, 0))[1])/hz, t) for t in b)
for (cpu, t) in rows[::-1][:6]:
    print('%6.3f s cpu = %5.2f ms/step  user %.2f sys %.2f  tid %d %s' % (cpu, cpu/STEPS*1e3, (b[t][0] - a.get(t, (0, 0))[0])/hz, (b[t][1] - a.get(t, (0, 0))[1])/hz,
                                                                       t, names.get(t, '(not a Python thread: runtime)')))
print('total %.2f ms/step' % (sum(r[0] for r in rows)/STEPS*1e3))

busiest = rows[-1][1]
print('samples of the busiest thread (tid %d): system call number + first two arguments | kernel wait channel -> count' % busiest)
for (key, n) in sorted(SAMPLES.items(), key=lambda kv: -kv[1]):
    if key[0] == busiest:
        print('   ', key[1], '|', key[2], '->', n)
import time
idle0 = snap(); time.sleep(1.0); idle1 = snap()
print('the same thread with the GPU idle for 1 s: %.3f s cpu' % ((idle1[busiest][0] - idle0[busiest][0] + idle1[busiest][1] - idle0[busiest][1])/hz))
print('context switches:', [l.strip() for l in open('/proc/self/task/%d/status' % busiest) if 'ctxt' in l])
