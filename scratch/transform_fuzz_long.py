"""Time-boxed fuzz of the whole transform chain against the C oracle: random batch / height / width (multiples of 16, odd
tile remainders included), fixed- and learned-bin-width models, random weights, biases, bin widths and map means; latents,
quantised latents, float and uint8 reconstructions must be equal bit for bit; and through codec.BatchCodec the symbols'
bit counts must equal the host coder's on the oracle's symbols. Uses oracle/ (a checker script, like the tests)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from oracle import transforms as orc
from autoencoder_based_image_compression_amd import pipeline, device as dev, codec
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import compression

rng = numpy.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.
gold = numpy.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'coder_golden.npz'))
probabilities = gold['real_probabilities_1']
t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    learned = bool(rng.randint(2))
    (n, h, w) = (int(rng.randint(1, 5)), 16*int(rng.randint(1, 11)), 16*int(rng.randint(1, 13)))
    v = var.random_variables(1., learned, seed=int(rng.randint(1 << 30)), bias_std=0.02)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    x = rng.randint(0, 256, size=(n, h, w)).astype(numpy.float64)
    for _ in range(int(rng.randint(0, 4))):
        x = (x + numpy.roll(x, 1, 1) + numpy.roll(x, -1, 2))/3.
    x = numpy.round(x).astype(numpy.uint8)
    bw = rng.uniform(0.4, 3., size=128).astype(numpy.float32)
    mm = rng.normal(scale=0.1, size=128).astype(numpy.float32)
    y_ref = orc.encoder(x.astype(numpy.float32)[..., None], v, learned)
    xd = torch.from_numpy(x).cuda()
    y = pipeline.DeviceEncoder(v, learned)(xd)
    assert numpy.array_equal(y.cpu().numpy(), y_ref), ('latents', cases, n, h, w, learned)
    tiled = numpy.tile(bw.reshape(1, 1, 1, 128), y_ref.shape[:3] + (1,))
    cq = tiled*numpy.round((y_ref - mm)/tiled)
    rec_ref = orc.decoder(cq + mm, v, learned)[..., 0]
    u8_ref = numpy.round(rec_ref.clip(min=16., max=235.)).astype(numpy.uint8)
    c = codec.BatchCodec(v, learned, bw, mm, probabilities, 67, n, h, w, keep_reconstruction=True, use_graphs=bool(rng.randint(2)),
                         nb_transform_streams=int(rng.randint(1, 3)), one_stream_steps=bool(rng.randint(2)))
    for _ in range(2):
        t = c.submit(xd)
        r = t.result()
    assert numpy.array_equal(t.reconstruction_uint8.cpu().numpy(), u8_ref), ('reconstruction', cases, n, h, w, learned)
    sse = ((x.astype(numpy.int64) - u8_ref.astype(numpy.int64))**2).reshape(n, -1).sum(axis=1)
    assert numpy.array_equal(r['sse'], sse), ('sse', cases)
    sym = numpy.round(cq/tiled).astype(numpy.int16)
    planar = numpy.ascontiguousarray(sym.transpose(0, 3, 1, 2).reshape(n, 128, -1))
    (_, nb) = compression.code_planar_symbols(planar, probabilities, 67, nb_threads=4)
    assert numpy.array_equal(r['coder_bits'], nb.astype(numpy.int64).sum(axis=1)), ('bits', cases, n, h, w)
    assert numpy.array_equal(r['nb_deads'], (numpy.abs(cq).sum(axis=(1, 2)) == 0).sum(axis=1)), ('deads', cases)
    c.close()
    cases += 1
print('cases', cases)
