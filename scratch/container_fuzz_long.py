"""Time-boxed fuzz of the container: random shapes, models, bin widths (low ones make long streams that leave the decoder's
LDS windows), exception map on / off: decode_images(encode_images(x)) must equal the in-memory reconstruction bit for bit
and decode_symbols the encoder's symbols."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from autoencoder_based_image_compression_amd import pipeline, device as dev, container
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var

rng = numpy.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.
gold = numpy.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'coder_golden.npz'))
t0 = time.time()
cases = 0
longest = 0
while time.time() - t0 < budget:
    learned = bool(rng.randint(2))
    (n, h, w) = (int(rng.randint(1, 4)), 16*int(rng.randint(1, 17)), 16*int(rng.randint(1, 17)))
    v = var.random_variables(1., learned, seed=int(rng.randint(1 << 30)), bias_std=0.02)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    for k in ('encoder/weights_3',):
        v[k] = (v[k]*numpy.float32(rng.choice([1., 4., 30.]))).astype(numpy.float32)       # larger latents: denser maps
    x = rng.randint(0, 256, size=(n, h, w)).astype(numpy.uint8)
    bw = rng.uniform(0.05, 2., size=128).astype(numpy.float32)
    mm = rng.normal(scale=0.1, size=128).astype(numpy.float32)
    exc = int(rng.choice([-1, 67, 0, 127]))
    enc = pipeline.DeviceEncoder(v, learned)
    decoder = pipeline.DeviceDecoder(v, learned)
    xd = torch.from_numpy(x).cuda()
    y = enc(xd)
    q = dev.quantize_maps(y, torch.from_numpy(bw).cuda(), torch.from_numpy(mm).cuda(), want_shifted=True, want_symbols=True)
    (_, rec, _) = decoder(q['shifted'])
    probs = gold['real_probabilities_1'] if rng.randint(2) else numpy.clip(rng.beta(0.7, 0.7, size=(128, 10)), 0.01, 0.99)
    try:
        (blob, info) = container.encode_images(x, enc, bw, mm, probs, exc)
    except Exception as e:        # symbols beyond int16 etc. are legitimate refusals; anything else is a bug
        assert type(e).__name__ in ('AssertionError', 'RuntimeError', 'OverflowError'), repr(e)
        continue
    header = container.read_header(blob)
    (_, sym) = container.decode_symbols(blob)
    assert numpy.array_equal(sym.cpu().numpy(), q['symbols'].cpu().numpy()), ('symbols', cases, n, h, w, exc)
    out = container.decode_images(blob, decoder)
    assert numpy.array_equal(out, rec.cpu().numpy()), ('images', cases, n, h, w, exc)
    longest = max(longest, int(header['bits'].max()))
    cases += 1
print('cases', cases, 'longest stream bits', longest)
