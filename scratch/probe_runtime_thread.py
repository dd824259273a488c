"""How much host CPU the HIP runtime's own thread burns per unit of GPU work, by submission pattern (torch only, no product code):
kernels of ~0.1 ms launched back to back for 2 s on one stream (a) with no event at all, (b) an event recorded every 7 launches,
(c) the same and a second thread polling the events, (d) a hipGraph of 27 kernels replayed, (e) long kernels (~3 ms)."""
import os, sys, threading, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch


def snap():
    out = {}
    for tid in os.listdir('/proc/self/task'):
        try:
            f = open('/proc/self/task/{}/stat'.format(tid)).read()
        except OSError:
            continue
        rest = f[f.rindex(')') + 2:].split()
        out[int(tid)] = int(rest[11]) + int(rest[12])
    return out


hz = os.sysconf('SC_CLK_TCK')
x = torch.zeros(64*1024*1024, device='cuda')          # 256 MB: one add_ is ~0.1 ms
big = torch.zeros(1024*1024*1024, device='cuda')      # 4 GB: ~2 ms
torch.cuda.synchronize()


def measure(label, body, seconds=2.0):
    mine = {t.native_id for t in threading.enumerate()}
    torch.cuda.synchronize()
    a = snap()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        n += body()
        while n > 400 and False:
            pass
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    b = snap()
    runtime = sum(b[t] - a.get(t, 0) for t in b if t not in mine)/hz
    python = sum(b[t] - a.get(t, 0) for t in b if t in mine)/hz
    print('%-58s wall %.2f s  kernels %6d  runtime threads %.2f s cpu (%.0f us per kernel)  python threads %.2f s' % (
        label, wall, n, runtime, runtime/max(n, 1)*1e6, python))


def plain():
    for _ in range(27):
        x.add_(1.)
    torch.cuda.current_stream().synchronize() if False else None
    return 27


events = []


def with_events():
    for i in range(27):
        x.add_(1.)
        if i % 7 == 6:
            e = torch.cuda.Event()
            e.record()
            events.append(e)
    if len(events) > 64:
        events[0].synchronize()
        del events[:32]
    return 27


def throttled():
    # keep the queue shallow: wait for the event of two bodies ago (what a slot does)
    for i in range(27):
        x.add_(1.)
    e = torch.cuda.Event()
    e.record()
    events.append(e)
    if len(events) > 4:
        old = events.pop(0)
        while not old.query():
            time.sleep(0.0002)
    return 27


g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(1.)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(27):
            x.add_(1.)


def graph():
    with torch.cuda.stream(s):
        g.replay()
        e = torch.cuda.Event()
        e.record()
        events.append(e)
    if len(events) > 4:
        old = events.pop(0)
        while not old.query():
            time.sleep(0.0002)
    return 27


def long_kernels():
    for _ in range(4):
        big.add_(1.)
    e = torch.cuda.Event()
    e.record()
    events.append(e)
    if len(events) > 4:
        old = events.pop(0)
        while not old.query():
            time.sleep(0.0002)
    return 4


measure('launches, no events (queue fills up)', plain)
del events[:]
measure('launches, an event every 7, waited for 64 behind', with_events)
del events[:]
measure('27 launches + 1 event, polled 4 bodies behind', throttled)
del events[:]
measure('graph of 27 kernels + 1 event, polled 4 behind', graph)
del events[:]
measure('4 long kernels + 1 event, polled 4 behind', long_kernels)
time.sleep(0.5)
a = snap(); time.sleep(1.0); b = snap()
print('idle second: %.2f s cpu in all threads' % (sum(b[t] - a.get(t, 0) for t in b)/hz))
