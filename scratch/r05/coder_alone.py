"""The coder chain of one batch alone on the GPU (binarise + encode core + emit; decode core + debinarise + compare), HIP events,
at the bin widths of bench.py's `realistic_entropy`: batch 24 and one image."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import bench          # noqa: E402
import torch          # noqa: E402

args = bench.parse_args(['--no-cpu-baseline'])
torch.cuda.set_device(0)
ctx = bench.Context(args, torch.device('cuda', 0), 1, 0, bench.usable_cpus())
for batch in (24, 1):
    for width in (1.0, 0.25, 0.125, 0.05, 0.0125):
        (enc, dec) = bench.coder_alone_ms(ctx, batch, bench.synthetic_model(width), 512, 768)
        print('batch {0:2d} bin width {1:6.4f}: binarise + encode {2:.4f} ms, decode + compare {3:.4f} ms'.format(batch, width, enc, dec))
