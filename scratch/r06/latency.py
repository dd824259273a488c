"""r06: one Kodak image at a time (submit -> result on the host: BASELINE.json configs[1] literally), with the host's share split into
the submit call and the result call. Run once per setting of EAE_RESULT_SPIN_SECONDS / EAE_RESULT_BY_CALLER (read at import)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from autoencoder_based_image_compression_amd import codec
args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
variables = bench.synthetic_model(1.)
n = int(os.environ.get('N_IMAGES', '200'))
spent = {'submit': 0., 'result': 0., 'n_submit': 0, 'n_result': 0}
(real_submit, real_result) = (codec.BatchCodec.submit, codec.Ticket.result)
def timed_submit(self, x):
    t0 = time.perf_counter(); out = real_submit(self, x); spent['submit'] += time.perf_counter() - t0; spent['n_submit'] += 1; return out
def timed_result(self):
    t0 = time.perf_counter(); out = real_result(self); spent['result'] += time.perf_counter() - t0; spent['n_result'] += 1; return out
if os.environ.get('LATENCY_SPLIT', '1') == '1':
    codec.BatchCodec.submit = timed_submit
    codec.Ticket.result = timed_result
for rep in range(3):
    for k in spent: spent[k] = 0
    alone = bench.run_pipeline(ctx, 1, n, 10, variables, 512, 768, coder_streams=1, transform_streams=1, use_graphs=True, serial=True)
    print('spin %s by_caller %s: %.4f ms per image (median block of %d); host cpu %.3f ms per image; submit %.1f us x %d, result %.1f us x %d (all blocks incl. warm-up)' % (
        os.environ.get('EAE_RESULT_SPIN_SECONDS', 'default'), os.environ.get('EAE_RESULT_BY_CALLER', 'default'), alone['elapsed']/n*1e3, len(alone['block_seconds']),
        alone['host_cpu_ms_per_step'][0], spent['submit']/max(spent['n_submit'], 1)*1e6, spent['n_submit'], spent['result']/max(spent['n_result'], 1)*1e6, spent['n_result']))
