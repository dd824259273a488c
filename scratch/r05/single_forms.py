"""r05: the two single-image figures of bench.py (pipelined ms per image; submit -> result latency) on their own, for sweeps over the
small-layer launch forms (the forms are read from the environment when the library loads: one process per setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
import bench
args = bench.parse_args(['--no-cpu-baseline', '--no-dropin-surface'])
args.transform_streams = 3
sys.setswitchinterval(1e-4)
torch.cuda.set_device(0)
ctx = bench.Context(args, torch.device('cuda', 0), 1, 0, bench.usable_cpus())
ctx.inputs = bench.load_inputs(args)
variables = bench.synthetic_model(1.0)
one = bench.run_pipeline(ctx, 1, 300, 30, variables, 512, 768, coder_streams=8, transform_streams=6, use_graphs=True)
alone = bench.run_pipeline(ctx, 1, 100, 10, variables, 512, 768, coder_streams=1, transform_streams=1, use_graphs=True, serial=True)
print('%-60s pipelined %.4f ms per image, one at a time %.4f ms' % (
    ' '.join('%s=%s' % (k, os.environ[k]) for k in ('EAE_HIP_PACK', 'EAE_HIP_FORCE_TILE', 'EAE_HIP_FORCE_NT', 'EAE_HIP_GEMM') if k in os.environ) or 'default',
    one['elapsed']/300*1e3, alone['elapsed']/100*1e3))
