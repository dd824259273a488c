"""r04: cProfile of the RESULT WORKER thread over 2,000 pipelined single-image steps (the launch thread was profiled by submit_profile.py)."""
import os, sys, cProfile, pstats, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from autoencoder_based_image_compression_amd import codec
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
prof = cProfile.Profile()
orig_run = codec._Worker.run
def run(self):
    prof.enable()
    try:
        orig_run(self)
    finally:
        prof.disable()
args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0); torch.cuda.set_device(device)
ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
variables = bench.synthetic_model(1.)
(h, w) = (int(os.environ.get('H', '512')), int(os.environ.get('W', '768')))
B = int(os.environ.get("B", "1"))
r0 = bench.run_pipeline(ctx, B, 20, 5, variables, h, w, coder_streams=8, transform_streams=6, use_graphs=True, min_seconds=0., max_blocks=1)
images = torch.from_numpy(bench.synthetic_images(1000, B, h, w)).to(device)
codec._Worker.run = run
with codec.BatchCodec(variables, False, variables[var.BIN_WIDTHS_NAME], r0['map_mean_host'], r0['probabilities'], bench.IDX_MAP_EXCEPTION, B, h, w,
                      device=device, **(codec.product_mode(h, w) if B > 2 else dict(nb_in_flight=8, nb_transform_streams=6, use_graphs=True))) as c:
    for _ in range(30):
        c.submit(images)
    c.drain()
    N = 2000 if B == 1 else 300
    t0 = time.perf_counter()
    tickets = [c.submit(images) for _ in range(N)]
    c.drain()
    print('%dx%d: %.4f ms per image' % (h, w, (time.perf_counter() - t0)/N*1e3))
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats('tottime').print_stats(16)
print(out.getvalue()[:4000])
