"""r04: what one submit() of one Kodak image costs the launch thread when no slot has to be waited for (the first submits of a block), against
the steady state of the pipelined single-image leg (where submit() may wait for a free slot)."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from autoencoder_based_image_compression_amd import codec
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0); torch.cuda.set_device(device)
ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
variables = bench.synthetic_model(1.)
# borrow run_pipeline's set-up by running it once (tables, statistics), then drive a codec of the same settings by hand
run = bench.run_pipeline(ctx, 1, 20, 5, variables, 512, 768, coder_streams=8, transform_streams=6, use_graphs=True, min_seconds=0., max_blocks=1)
images = torch.from_numpy(bench.synthetic_images(1000, 1, 512, 768)).to(device)
bin_widths = variables[var.BIN_WIDTHS_NAME]
with codec.BatchCodec(variables, False, bin_widths, run['map_mean_host'], run['probabilities'], bench.IDX_MAP_EXCEPTION, 1, 512, 768,
                      device=device, nb_in_flight=8, nb_transform_streams=6, use_graphs=True) as c:
    for _ in range(30):
        c.submit(images)
    c.drain()
    first, steady = [], []
    for rep in range(20):
        c.drain(); torch.cuda.synchronize()
        ts = []
        for i in range(40):
            t0 = time.perf_counter(); c.submit(images); ts.append(time.perf_counter() - t0)
        first.extend(ts[:6]); steady.extend(ts[20:])
    c.drain()
    print('submit() of one image: %.1f us while slots are free (median of the first 6 of a burst), %.1f us in the steady state of the burst'
          % (statistics.median(first)*1e6, statistics.median(steady)*1e6))
