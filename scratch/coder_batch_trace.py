"""The device coder alone on bench.py's symbols for (h, w, batch): run under `rocprofv3 --kernel-trace --stats` to see the
duration of every coder kernel without the transforms beside it."""
import os, sys
import numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats

(h, w, batch) = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
map_size = (h//16)*(w//16)
variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False, 'cuda')
images = torch.from_numpy(bench.synthetic_images(1000, batch, h, w)).cuda()
y = enc(images)
mm = dev.map_means(y).cpu().numpy()
probs = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), variables[var.BIN_WIDTHS_NAME], mm, 10)
q = dev.quantize_maps(y, torch.from_numpy(variables[var.BIN_WIDTHS_NAME]).cuda(), torch.from_numpy(mm).cuda(), want_shifted=False,
                      want_symbols=True, want_flags=False)
sym = q['symbols'].reshape(-1, map_size)
p = torch.from_numpy(probs).cuda()
rows = torch.arange(128, dtype=torch.int32).repeat(batch)
rows[67::128] = -1
rows = rows.cuda()
ws = dev.coder_workspace(sym.shape[0], map_size, 10, sym.device)
s = dev.coder_encode_batch(sym, p, rows, 10, workspace=ws)
for _ in range(5):
    torch.cuda.synchronize()
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record()
    dev.coder_encode_batch(sym, p, rows, 10, out=s, workspace=ws)
    b.record()
    dev.coder_decode_batch(s, p, rows, expected=sym, workspace=ws)
    c.record()
    torch.cuda.synchronize()
    print('encode ms', round(a.elapsed_time(b), 3), 'decode+compare ms', round(b.elapsed_time(c), 3), 'errors', int((s.status != 0).sum().item()))
