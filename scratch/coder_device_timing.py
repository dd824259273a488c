"""Times the device coder on the bench's own symbols (24 Kodak-sized images) for several lanes-per-wave settings."""
import os, sys, time
import numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats

variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False, 'cuda')
images = torch.from_numpy(bench.synthetic_images(1000, 24, 512, 768)).cuda()
y = enc(images)
mm = dev.map_means(y).cpu().numpy()
probs = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), variables[var.BIN_WIDTHS_NAME], mm, 10)
q = dev.quantize_maps(y, torch.from_numpy(variables[var.BIN_WIDTHS_NAME]).cuda(), torch.from_numpy(mm).cuda(), want_shifted=False, want_symbols=True, want_flags=False)
sym = q['symbols'].reshape(-1, 1536)
p = torch.from_numpy(probs).cuda()
rows = torch.arange(128, dtype=torch.int32).repeat(24)
rows[67::128] = -1
rows = rows.cuda()
print('abs mean', sym.abs().float().mean().item(), 'max', sym.abs().max().item())
for mode in (1, 3, 4, 5):
    for lanes in ((0, 1, 8) if mode < 4 else (64,)):
        (s, _) = dev.coder_compress_maps(sym, p, rows, 10, mode=min(mode, 2) if mode < 4 else 1, lanes_per_wave=lanes)
        ws = dev.coder_workspace(sym.shape[0], 1536, 10, sym.device)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            if mode == 3:
                dev.coder_verify_maps(s, sym, p, rows, lanes)
            elif mode == 4:
                dev.coder_encode_batch(sym, p, rows, 10, out=s, workspace=ws)
            elif mode == 5:
                dev.coder_decode_batch(s, p, rows, expected=sym, workspace=ws)
            else:
                dev.coder_compress_maps(sym, p, rows, 10, mode=mode, out=s, lanes_per_wave=lanes)
        b.record(); torch.cuda.synchronize()
        print('mode', mode, 'lanes', lanes, 'ms', round(a.elapsed_time(b)/5, 3), 'bits', int(s.nb_bits().sum().item()), 'errors', int((s.status != 0).sum().item()))
nb = s.nb_bits().cpu().numpy().astype(numpy.int64)
a = sym.abs().to(torch.int64)
dec = (torch.clamp(a, max=10) + (a < 10).to(torch.int64)).sum(1).cpu().numpy()
nz = (a != 0).sum(1).cpu().numpy()
order = numpy.argsort(nb)[::-1][:8]
print('bits per map: mean', nb.mean(), 'p50', numpy.percentile(nb, 50), 'p99', numpy.percentile(nb, 99), 'max', nb.max())
print('densest maps (map, bits, decisions, nonzeros):', [(int(m), int(nb[m]), int(dec[m]), int(nz[m])) for m in order])
print('decisions per map: mean', dec.mean(), 'max', dec.max())
def timeit(symbols, mode, lanes=1):
    (s2, _) = dev.coder_compress_maps(symbols, p, rows, 10, mode=min(mode, 2), lanes_per_wave=lanes)
    torch.cuda.synchronize()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record()
    for _ in range(5):
        if mode == 3:
            dev.coder_verify_maps(s2, symbols, p, rows, lanes)
        else:
            dev.coder_compress_maps(symbols, p, rows, 10, mode=mode, out=s2, lanes_per_wave=lanes)
    b_.record(); torch.cuda.synchronize()
    return round(a_.elapsed_time(b_)/5, 3)
zeros = torch.zeros_like(sym)
ones = torch.ones_like(sym)
few = sym[:64].contiguous()
print('all-zero symbols: encode', timeit(zeros, 1), 'verify', timeit(zeros, 3))
print('all-one symbols: encode', timeit(ones, 1), 'verify', timeit(ones, 3))
rows = rows[:64].contiguous()
print('64 maps only: encode', timeit(few, 1), 'verify', timeit(few, 3))
def timeit(symbols, mode, lanes=1):
    (s2, _) = dev.coder_compress_maps(symbols, p, rows, 10, mode=min(mode, 2), lanes_per_wave=lanes)
    torch.cuda.synchronize()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record()
    for _ in range(5):
        if mode == 3:
            dev.coder_verify_maps(s2, symbols, p, rows, lanes)
        else:
            dev.coder_compress_maps(symbols, p, rows, 10, mode=mode, out=s2, lanes_per_wave=lanes)
    b_.record(); torch.cuda.synchronize()
    return round(a_.elapsed_time(b_)/5, 3)
zeros = torch.zeros_like(sym)
ones = torch.ones_like(sym)
few = sym[:64].contiguous()
print('all-zero symbols: encode', timeit(zeros, 1), 'verify', timeit(zeros, 3))
print('all-one symbols: encode', timeit(ones, 1), 'verify', timeit(ones, 3))
rows = rows[:64].contiguous()
print('64 maps only: encode', timeit(few, 1), 'verify', timeit(few, 3))
if os.environ.get('EAE_CODER_DEBUG_CLOCKS'):
    rows = torch.arange(128, dtype=torch.int32).repeat(24).cuda()
    for lanes in ((0, 1, 8) if mode < 4 else (64,)):
        (s3, _) = dev.coder_compress_maps(sym, p, rows, 10, mode=1, lanes_per_wave=lanes)
        (s3, _) = dev.coder_compress_maps(sym, p, rows, 10, mode=1, lanes_per_wave=lanes, out=s3)
        torch.cuda.synchronize()
        cyc = s3.stage.cpu().numpy().astype(numpy.float64); ref = s3.bypass_bits.cpu().numpy().astype(numpy.float64)
        print('standalone lanes', lanes, 'cycles/map mean', cyc.mean(), 'max', cyc.max(), 'refclk ticks', ref.mean(), 'MHz', 100*cyc.mean()/ref.mean())
print('--- denser symbols (as with half the bin width): batch kernels')
rows = torch.arange(128, dtype=torch.int32).repeat(24); rows[67::128] = -1; rows = rows.cuda()
for mul in (1, 2, 4):
    dense = (sym.to(torch.int32)*mul + (torch.randint(0, mul, sym.shape, device='cuda', dtype=torch.int32) if mul > 1 else 0)).to(torch.int16)
    ws = dev.coder_workspace(dense.shape[0], 1536, 10, dense.device)
    s4 = dev.coder_encode_batch(dense, p, rows, 10, workspace=ws)
    torch.cuda.synchronize()
    a_, b_, c_ = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    a_.record()
    for _ in range(5):
        dev.coder_encode_batch(dense, p, rows, 10, out=s4, workspace=ws)
    b_.record()
    for _ in range(5):
        dev.coder_decode_batch(s4, p, rows, expected=dense, workspace=ws)
    c_.record(); torch.cuda.synchronize()
    nb = s4.bac_bits.cpu().numpy()
    print('x%d: bits/map mean %.0f max %d  maps beyond the 2048-bit window %d  encode %.3f ms  decode+compare %.3f ms  errors %d' % (
        mul, nb[nb > 0].mean(), nb.max(), int((nb > 2048).sum()), a_.elapsed_time(b_)/5, b_.elapsed_time(c_)/5, int((s4.status != 0).sum().item())))
