"""The PCIe-inclusive leg of bench.py on its own (uint8 batch in from pinned host memory, uint8 reconstruction back, every step)
next to the resident leg. Usage: python scratch/r03_pcie.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
import bench
import torch

args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
variables = bench.synthetic_model(1.)
n = bench.auto_coder_streams(512, 768)
for pcie in (False, True, False, True):
    run = bench.run_pipeline(ctx, 24, STEPS, 10, variables, 512, 768, coder_streams=n, transform_streams=2, use_graphs=True, min_seconds=0.6,
                             max_blocks=5, pcie=pcie)
    print('pcie' if pcie else 'resident', '%.3f ms/step  %.1f Mpx/s' % (run['elapsed']/STEPS*1e3, STEPS*24*512*768/run['elapsed']/1e6),
          'host cpu ms/step', [round(float(v), 2) for v in run.get('cpu_ms', [])] if 'cpu_ms' in run else '')
