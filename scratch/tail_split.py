"""conv2+GDN2 on 24 Kodak-sized images: one launch vs main part (fused kernel) + a tail of t images launched on a second,
lower-priority stream (small-granularity NT=2 blocks + GDN pass) that should fill the main kernel's last, partly empty round."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline

variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False, 'cuda')
v = enc.v
n = 24
x = torch.randn((n, 128, 192, 128), device='cuda')*0.5
out = torch.empty((n, 64, 96, 128), device='cuda')
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)

def conv(lo_i, hi_i):
    dev.conv5x5s2(x[lo_i:hi_i], enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], out=out[lo_i:hi_i])

def run(t, reps=30):
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        with torch.cuda.stream(hi):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if t:
                ev_in = torch.cuda.Event(); ev_in.record()
                lo.wait_event(ev_in)
                with torch.cuda.stream(lo):
                    conv(n - t, n)
                    ev_t = torch.cuda.Event(); ev_t.record()
            conv(0, n - t)
            if t:
                hi.wait_event(ev_t)
            b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    times.sort()
    return times[len(times)//2]

ref = None
for t in (0, 1, 2, 3, 4, 6):
    ms = run(t)
    print('tail images', t, 'ms', round(ms, 4))
    if t == 0:
        ref = out.clone()
    else:
        assert torch.equal(out, ref)
