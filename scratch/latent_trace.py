"""Phase breakdown of latent_wave_kernel from its tracing build (SRC=latent EXTRA=-DEAE_LATENT_TRACE scratch/variant.sh)."""
import os, sys, numpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoencoder_based_image_compression_amd import device as dev
rng = numpy.random.RandomState(0)
x = torch.from_numpy((rng.laplace(size=(24, 32, 48, 128))).astype(numpy.float32)).cuda()
gamma = rng.uniform(2e-5, 0.01, size=(128, 128)).astype(numpy.float32); gamma = torch.from_numpy(0.5*(gamma + gamma.T)).cuda()
g3 = dev.pack_gamma(gamma); g4 = dev.pack_gamma(gamma*2)
b = torch.ones(128, device='cuda'); bw = torch.ones(128, device='cuda'); mean = torch.zeros(128, device='cuda')
sym = torch.empty((24, 128, 1536), dtype=torch.int16, device='cuda'); flags = torch.zeros((24, 128), dtype=torch.int32, device='cuda')
checks = torch.zeros(64, dtype=torch.int32, device='cuda')
def run():
    return dev.latent_stage(x, bw, mean, gdn_in=(g3, b), igdn_out=(g4, b), want_symbols=True, want_flags=True, out_symbols=sym, out_flags=flags, out_checks=checks)
for _ in range(3): run()
torch.cuda.synchronize()
checks.zero_()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); e.record(); torch.cuda.synchronize()
acc = checks[16:20].cpu().numpy().astype('float64')*16
waves = 24*32*48//32
print('launch %.1f us; ticks per wave:' % (a.elapsed_time(e)*1e3))
for (n, v) in zip(['gdn_3 (MFMA loop + sqrt, /)', 'quantiser + stores', 'inverse_gdn_4', 't_out stores'], acc):
    print('  %-28s %9.0f' % (n, v/waves))
