"""How many HIP streams really run concurrently on this box (one GPU): k streams, one ~1 ms single-block spin kernel each
(torch.cuda._sleep), wall time of the lot -> concurrency = k / (wall / one). Run for several GPU_MAX_HW_QUEUES settings
(a fresh process each: the runtime reads the flag once) with and without the default stream among the k.
usage: python scratch/probe_streams.py            (driver: spawns the children)
       python scratch/probe_streams.py child <use_default 0|1>"""
import os
import subprocess
import sys
import time


def child(use_default):
    import torch
    torch.cuda.init()
    cycles = int(2.0e6)
    torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    out = []
    pool = [torch.cuda.Stream() for _ in range(12)]
    for k in (1, 2, 3, 4, 5, 6, 8, 10, 12):
        streams = pool[:k]
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for (i, s) in enumerate(streams):
                if use_default and i == 0:
                    torch.cuda._sleep(cycles)
                else:
                    with torch.cuda.stream(s):
                        torch.cuda._sleep(cycles)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out.append('{0}:{1:.1f}'.format(k, k*one/best))
    print('GPU_MAX_HW_QUEUES={0} default_stream_used={1} one={2:.3f} ms  concurrency by k: {3}'.format(
        os.environ.get('GPU_MAX_HW_QUEUES'), use_default, one*1e3, ' '.join(out)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(int(sys.argv[2]))
    else:
        for q in (None, '2', '4', '8', '16', '32'):
            for use_default in (0, 1):
                env = dict(os.environ)
                env.pop('GPU_MAX_HW_QUEUES', None)
                if q:
                    env['GPU_MAX_HW_QUEUES'] = q
                subprocess.run([sys.executable, os.path.abspath(__file__), 'child', str(use_default)], env=env)
