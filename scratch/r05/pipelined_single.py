"""r05: one Kodak image per step, pipelined (6 transform streams, 8 coder batches in flight, graphs): the leg of bench.py on its own,
for a kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
import bench
args = bench.parse_args(['--no-cpu-baseline', '--no-dropin-surface'] + (['--fuse-latent'] if os.environ.get('FUSE') == '1' else []))
args.transform_streams = 3
sys.setswitchinterval(1e-4)
torch.cuda.set_device(0)
ctx = bench.Context(args, torch.device('cuda', 0), 1, 0, bench.usable_cpus())
ctx.inputs = bench.load_inputs(args)
variables = bench.synthetic_model(1.0)
(ts, cs) = (int(os.environ.get('TS', '6')), int(os.environ.get('CS', '8')))
one = bench.run_pipeline(ctx, 1, int(os.environ.get('STEPS', '300')), 30, variables, 512, 768, coder_streams=cs, transform_streams=ts, use_graphs=os.environ.get('GRAPHS', '1') == '1', one_stream_steps=os.environ.get('ONE', '0') == '1')
print('transform streams %d, coder in flight %d: pipelined %.4f ms per image; process CPU per image %s ms' % (ts, cs, one['elapsed']/int(os.environ.get('STEPS', '300'))*1e3, one['host_cpu_ms_per_step']))
# is the launching thread the limit? its own CPU time and the process's, per image, over a second run
import time, threading
(w0, p0, t0) = (time.perf_counter(), time.process_time(), time.thread_time())
n = int(os.environ.get('STEPS', '300'))
again = bench.run_pipeline(ctx, 1, n, 30, variables, 512, 768, coder_streams=cs, transform_streams=ts, use_graphs=os.environ.get('GRAPHS', '1') == '1', one_stream_steps=os.environ.get('ONE', '0') == '1')
(w1, p1, t1) = (time.perf_counter(), time.process_time(), time.thread_time())
print('second run: %.4f ms per image timed; whole call (incl. construction, capture, warm-up) wall %.1f ms, process CPU %.1f ms, this thread CPU %.1f ms' % (
    again['elapsed']/n*1e3, (w1 - w0)*1e3, (p1 - p0)*1e3, (t1 - t0)*1e3))
