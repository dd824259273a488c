"""One Kodak image per step (BASELINE.json configs[1] literally), pipelined over 6 transform streams as bench.py's `single_image`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
variables = bench.synthetic_model(1.)
for (b, h, w) in ((1, 512, 768), (4, 512, 768), (1, 256, 256)):
    run = bench.run_pipeline(ctx, b, 300, 30, variables, h, w, coder_streams=8, transform_streams=6, use_graphs=True, min_seconds=0.5, max_blocks=5)
    print('%d x %dx%d per step: %.4f ms/step  %.1f Mpx/s' % (b, h, w, run['elapsed']/300*1e3, 300*b*h*w/run['elapsed']/1e6))
