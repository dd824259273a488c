"""codec.BatchCodec on the largest image the kernels accept (1 x 8192 x 8176: maps of 261,632 symbols): per-map bits against the host
library, squared error against the reconstruction it returns."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from autoencoder_based_image_compression_amd import codec, device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import compression
(H, W) = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 8176)
v = var.random_variables(1., False, seed=0, bias_std=0.01)
v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
rng = numpy.random.RandomState(5)
x = rng.randint(16, 236, size=(1, H, W), dtype=numpy.uint8)
for _ in range(2):
    x = ((x.astype(numpy.uint16) + numpy.roll(x, 1, 1) + numpy.roll(x, 1, 2) + numpy.roll(x, -1, 2))//4).astype(numpy.uint8)
bw = numpy.ones(128, dtype=numpy.float32)
mean = numpy.zeros(128, dtype=numpy.float32)
probabilities = numpy.clip(rng.rand(128, 10), 0.05, 0.95)
xd = torch.from_numpy(x).cuda()
t0 = time.time()
with codec.BatchCodec(v, False, bw, mean, probabilities, 67, 1, H, W, nb_in_flight=1, keep_reconstruction=True) as c:
    ticket = c.submit(xd)
    values = ticket.result()
    print('codec: %.2f s, bits per pixel %.4f' % (time.time() - t0, float(values['nb_bits'][0])/(H*W)))
    rec = ticket.reconstruction_uint8.cpu().numpy()
    assert int(values['sse'][0]) == int(((x.astype(numpy.int64) - rec.astype(numpy.int64))**2).sum())
    y = pipeline.DeviceEncoder(v, False)(xd)
    q = dev.quantize_maps(y, torch.from_numpy(bw).cuda(), torch.from_numpy(mean).cuda(), want_symbols=True)
    symbols = q['symbols'].cpu().numpy()
    t0 = time.time()
    (rec_sym, nb_bits) = compression.code_planar_symbols(symbols, probabilities, 67)
    print('host coder: %.2f s' % (time.time() - t0))
    assert numpy.array_equal(rec_sym, symbols)
    assert int(values['coder_bits'][0]) == int(nb_bits.astype(numpy.int64).sum()), (int(values['coder_bits'][0]), int(nb_bits.astype(numpy.int64).sum()))
    print('bits, squared error and round trip agree: coder bits', int(values['coder_bits'][0]), 'exception bits', int(values['exception_bits'][0]))
