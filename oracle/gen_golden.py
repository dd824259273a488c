"""Generates tests/golden/*.npz by running the REAL reference code in this container -- TEST INFRASTRUCTURE ONLY.

  python oracle/gen_golden.py            (needs /root/reference and `make -C oracle ref`)

* the reference's numpy modules (kodak_tensorflow/tools/tools.py, lossless/compression.py, lossless/stats.py,
  svhn/...) are imported from /root/reference with the one-line `numpy.float = numpy.floating` shim (numpy >= 1.24
  removed `numpy.float`, tools.py:91,124);
* the reference's C++ coder is the build in oracle/_ref (oracle/ref_shim.cpp over the untouched sources); it is also
  installed as `lossless.interface_cython` so that the reference's own `rescale_compress_lossless_maps` runs on it.
Only inputs and expected outputs are written (data, never source). The reference cannot travel to the GPU box; these
fixtures can.
"""
import os
import pickle
import sys
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference/kodak_tensorflow'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

from oracle import coder as oracle_coder   # noqa: E402


def import_reference():
    numpy.float = numpy.floating   # the shim
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.path.insert(0, REF)
    ref = oracle_coder.CoderLib('ref')
    fake = types.ModuleType('lossless.interface_cython')

    def compress_lossless_flattened_map(ref_map_int16, probabilities):
        # what interface_cython.pyx:13-59 does, on the real C++ through oracle/ref_shim.cpp
        (rec, nb_bits) = ref.compress_lossless(ref_map_int16, probabilities)
        return (rec, nb_bits)
    fake.compress_lossless_flattened_map = compress_lossless_flattened_map
    import lossless
    sys.modules['lossless.interface_cython'] = fake
    lossless.interface_cython = fake
    import tools.tools as tls
    import lossless.compression
    import lossless.stats
    return (ref, tls, lossless.compression, lossless.stats)


def dump_streams(ref, symbols, probabilities):
    c = ref.coder(len(symbols)*max(32, len(probabilities)), probabilities)
    for v in symbols:
        c.write_signed_ueg0(int(v))
    c.stop_bac_encoding()
    return (c.bytes_bac(), c.written_bac(), c.bytes_bypass(), c.written_bypass())


def coder_golden(ref):
    g = {}
    # ---- the reference's own known-answer cases (c++/source/tests.cpp) -------------------------------------------
    # test_compress_lossless, tests.cpp:354-376
    x = numpy.array([0, -2, 0, 765, -21, 8, -439, 0, 0, 0, 0, -9], dtype=numpy.int16)
    p = numpy.full(8, 0.5)
    (rec, nb) = ref.compress_lossless(x, p)
    (bb, nbb, yb, nyb) = dump_streams(ref, x, p)
    g.update(ka_compress_in=x, ka_compress_p=p, ka_compress_rec=rec, ka_compress_bits=nb, ka_compress_bac=bb,
             ka_compress_bac_bits=nbb, ka_compress_byp=yb, ka_compress_byp_bits=nyb)
    # test_binary_arithmetic_coder, tests.cpp:69-132
    probs20 = numpy.array([0.01, 0.99, 0.9, 0.76, 0.1, 0.01, 0.99, 0.5, 0.51, 0.2, 0.52, 0.01, 0.1, 0.01, 0.2, 0.90, 0.05, 0.5, 0.53, 0.2])
    bits20 = numpy.array([1 if 8 <= i <= 14 else 0 for i in range(20)], dtype=numpy.uint8)
    c = ref.coder(72, probs20)
    for (b, pr) in zip(bits20, probs20):
        c.bac_encoding(int(b), float(pr))
    c.stop_bac_encoding()
    g.update(ka_bac_bits_in=bits20, ka_bac_p=probs20, ka_bac_stream=c.bytes_bac(), ka_bac_nbits=c.written_bac())
    c.start_bac_decoding()
    g['ka_bac_decoded'] = numpy.array([c.bac_decoding(float(pr)) for pr in probs20], dtype=numpy.uint8)
    # test_read_signed_ueg0, tests.cpp:280-352
    x = numpy.array([0, 1, -2, -7, 8, -8, 9, -9, 127, -523], dtype=numpy.int16)
    c = ref.coder(200, numpy.full(8, 0.5))
    for v in x:
        c.write_signed_ueg0(int(v))
    c.stop_bac_encoding()
    g.update(ka_sueg0_in=x, ka_sueg0_bac=c.bytes_bac(), ka_sueg0_bac_bits=c.written_bac(), ka_sueg0_byp=c.bytes_bypass(),
             ka_sueg0_byp_bits=c.written_bypass())
    # test_read_eg0, tests.cpp:164-209 and test_read_truncated_unary, tests.cpp:211-278
    x = numpy.array([0, 1, 2, 11, 128, 504, 65535], dtype=numpy.uint16)
    c = ref.coder(231, numpy.full(8, 0.1))
    for v in x:
        c.write_eg0(int(v))
    g.update(ka_eg0_in=x, ka_eg0_byp=c.bytes_bypass(), ka_eg0_byp_bits=c.written_bypass(),
             ka_eg0_decoded=numpy.array([c.read_eg0() for _ in x], dtype=numpy.uint16))
    c = ref.coder(56, numpy.full(8, 0.5))
    for v in x:
        c.write_truncated_unary(int(v))
    c.stop_bac_encoding()
    g.update(ka_tu_bac=c.bytes_bac(), ka_tu_bac_bits=c.written_bac())
    c.start_bac_decoding()
    g['ka_tu_decoded'] = numpy.array([c.read_truncated_unary() for _ in x], dtype=numpy.uint16)
    # test_lossless.py:96-101
    x = numpy.array([0, 1, -2, 2, 1, 0, 0, 0], dtype=numpy.int16)
    (rec, nb) = ref.compress_lossless(x, numpy.array([0.5, 0.5, 0.5]))
    g.update(ka_flat_in=x, ka_flat_rec=rec, ka_flat_bits=nb)
    # count_nb_bits, tests.cpp:26-35 (0..64) and the whole EG0 domain
    g['nb_bits_0_65536'] = numpy.array([ref.count_nb_bits(i) for i in range(65537)], dtype=numpy.uint8)

    # ---- synthetic Laplace maps x REAL probability rows of model 1_10000 (lossless/results/...) ---------------------
    stats_dir = os.path.join(REF, 'lossless/results/1_10000/training_index_10')
    rng = numpy.random.RandomState(7)
    cases = []
    for (mult, scale) in (('1', 2.5), ('2', 1.2), ('10', 0.4)):
        probs = numpy.load(os.path.join(stats_dir, 'binary_probabilities_{}.npy'.format(mult)))
        g['real_probabilities_{}'.format(mult)] = probs
        for row in (0, 67, 127):
            x = numpy.round(rng.laplace(size=1536)*scale).astype(numpy.int16)
            cases.append((x, probs[row]))
    g['real_map_mean'] = numpy.load(os.path.join(stats_dir, 'map_mean.npy'))
    with open(os.path.join(stats_dir, 'idx_map_exception.pkl'), 'rb') as f:
        g['real_idx_map_exception'] = numpy.int64(pickle.load(f))
    # ---- edge cases --------------------------------------------------------------------------------------------------
    cases.append((numpy.zeros(64, dtype=numpy.int16), numpy.full(10, 0.9)))                       # all-zero map
    cases.append((numpy.array([32767, -32767, 0, 1, -1, 32767], dtype=numpy.int16), numpy.full(10, 0.5)))   # extremes
    cases.append((numpy.array([-32768, 5, -32768], dtype=numpy.int16), numpy.full(4, 0.3)))     # abs() wraps to 32768
    cases.append((numpy.arange(-40, 41, dtype=numpy.int16), numpy.array([0.5])))                # L = 1
    cases.append((numpy.arange(-60, 61, dtype=numpy.int16), numpy.linspace(0.9, 0.1, 40)))      # L = 40, |x| >= L -> EG0
    cases.append((numpy.round(rng.laplace(size=500)*4).astype(numpy.int16), numpy.full(10, 0.01)))   # p = 0.01
    cases.append((numpy.round(rng.laplace(size=500)*4).astype(numpy.int16), numpy.full(10, 0.99)))   # p = 0.99
    cases.append((numpy.array([3], dtype=numpy.int16), numpy.full(255, 0.5)))                   # L = 255
    cases.append((numpy.round(rng.laplace(size=300)*0.2).astype(numpy.int16), numpy.clip(rng.rand(10), 0.02, 0.98)))
    g['nb_cases'] = numpy.int64(len(cases))
    for (i, (x, p)) in enumerate(cases):
        (rec, nb) = ref.compress_lossless(x, p)
        (bb, nbb, yb, nyb) = dump_streams(ref, x, p)
        assert nb == nbb + nyb and numpy.array_equal(rec, x)
        g['case{}_in'.format(i)] = x
        g['case{}_p'.format(i)] = numpy.asarray(p, dtype=numpy.float64)
        g['case{}_bac'.format(i)] = bb
        g['case{}_bac_bits'.format(i)] = numpy.int64(nbb)
        g['case{}_byp'.format(i)] = yb
        g['case{}_byp_bits'.format(i)] = numpy.int64(nyb)
    # ---- error behaviour ------------------------------------------------------------------------------------------------
    errors = []
    pd = os.path.join(REF, 'lossless/pseudo_data')
    for name in ('binary_probabilities_scale_compress_invalid_0.npy', 'binary_probabilities_scale_compress_invalid_1.npy',
                 'binary_probabilities_scale_compress_valid.npy'):
        g['pseudo_' + name[:-4]] = numpy.load(os.path.join(pd, name))
    err_cases = [
        (numpy.array([1, 2, 3], dtype=numpy.int16), numpy.array([0.5, numpy.nan, 0.5])),     # NaN used -> type 4
        (numpy.array([0, 0, 0], dtype=numpy.int16), numpy.array([0.5, numpy.nan, 0.5])),     # NaN never used -> fine
        (numpy.array([2], dtype=numpy.int16), numpy.array([0.5, 0., 0.5])),
        (numpy.array([2], dtype=numpy.int16), numpy.array([0.5, 1., 0.5])),
        (numpy.array([2], dtype=numpy.int16), numpy.array([1.5, 0.5, 0.5])),
        (numpy.full(40, 30, dtype=numpy.int16), numpy.full(10, 0.99)),                        # capacity error (type 1)
        (numpy.zeros(0, dtype=numpy.int16), numpy.full(10, 0.5)),                             # empty -> type 1 at stop
    ]
    g['nb_err_cases'] = numpy.int64(len(err_cases))
    for (i, (x, p)) in enumerate(err_cases):
        try:
            (rec, nb) = ref.compress_lossless(x, p)
            msg = 'ok:{}'.format(nb)
        except Exception as exc:
            msg = '{0}:{1}'.format(type(exc).__name__, exc)
        errors.append(msg)
        g['err{}_in'.format(i)] = x
        g['err{}_p'.format(i)] = p
    g['err_messages'] = numpy.array(errors)
    numpy.savez_compressed(os.path.join(OUT, 'coder_golden.npz'), **g)
    print('coder_golden.npz:', len(cases), 'stream cases,', errors)


def tools_golden(tls, ref_compression, ref_stats, g_coder_dir):
    g = {}
    rng = numpy.random.RandomState(11)
    # quantiser half-way cases (round half to even everywhere)
    halves = numpy.array([-3.5, -2.5, -1.5, -0.5, 0.5, 1.5, 2.5, 3.5, 0.49999997, -0.49999997, 1e-8, 7.25], dtype=numpy.float32)
    data = numpy.zeros((1, 3, 4, 128), dtype=numpy.float32)
    data[0, :, :, 0] = halves.reshape(3, 4)
    data[0, :, :, 1] = halves.reshape(3, 4)*numpy.float32(0.3)
    data[0, :, :, 2:] = rng.laplace(size=(3, 4, 126)).astype(numpy.float32)*2
    bw = rng.uniform(0.3, 2.5, size=128).astype(numpy.float32)
    bw[0] = 1.
    bw[1] = numpy.float32(0.3)
    g.update(q_in=data, q_bw=bw, q_out=tls.quantize_per_map(data, bw))
    g.update(int16_in=halves*numpy.float32(3.), int16_out=tls.cast_float_to_int16(halves*numpy.float32(3.)))
    bt = numpy.concatenate([numpy.array([15.431, -0.001, 0., 235.678, 143.18, 1.111], dtype=numpy.float32),
                            (numpy.arange(14, 238) + 0.5).astype(numpy.float32), rng.uniform(-20, 280, size=200).astype(numpy.float32)])
    g.update(bt601_in=bt, bt601_out=tls.cast_bt601(bt))
    # seeded latents (2, 8, 12, 128): rate_3d, count_nb_deads, discrete_entropy, count_symbols
    y = (rng.laplace(size=(2, 8, 12, 128))*rng.uniform(0.2, 6., size=128)).astype(numpy.float32)
    y[:, :, :, 5] = 0.
    y[0, :, :, 9] = 0.
    bw2 = rng.uniform(0.5, 2., size=128).astype(numpy.float32)
    cq = tls.quantize_per_map(y, bw2)
    g.update(lat_y=y, lat_bw=bw2, lat_cq=cq, lat_nb_deads=tls.count_nb_deads(cq),
             lat_rate=numpy.array([tls.rate_3d(cq[j], bw2, 128, 192) for j in range(2)]),
             lat_entropy=numpy.array([tls.discrete_entropy(cq[0, :, :, c], bw2[c].item()) for c in range(128)]),
             lat_average_entropies=numpy.float64(tls.average_entropies(y, bw2)))
    hist = tls.count_symbols(cq[0, :, :, 3], bw2[3].item())
    g.update(lat_count_symbols_3=hist)
    # reference's own entropy example (test_tools.py:268-320): 5 x 0.1, 0.2, 0.3
    samples = numpy.array([0., 0., 0., 1., 1., 2., 3., 4., 5., 6.], dtype=numpy.float32)   # counts 3,2,1,1,1,1,1
    g.update(ent_in=samples, ent_out=numpy.float64(tls.discrete_entropy(samples, 1.)))
    # psnr_2d (test_tools.py:494-509: 12s vs 15s -> 38.5883785143)
    a = numpy.full((4, 6), 12, dtype=numpy.uint8)
    b = numpy.full((4, 6), 15, dtype=numpy.uint8)
    g.update(psnr_known=numpy.float64(tls.psnr_2d(a, b)))
    ra = rng.randint(0, 256, size=(32, 48)).astype(numpy.uint8)
    rb = rng.randint(0, 256, size=(32, 48)).astype(numpy.uint8)
    g.update(psnr_a=ra, psnr_b=rb, psnr_ab=numpy.float64(tls.psnr_2d(ra, rb)))
    # float_to_str (tools.py:589-593)
    g['float_to_str'] = numpy.array([tls.float_to_str(v) for v in (1., 0.5, 1.25, -2., 10000., -0.75)])
    # stats: count_binary_decisions hand cases (test_lossless.py:257-298) + compute_binary_probabilities
    abs1 = numpy.array([0.75, 0.05, 0.1, 0.2, 0.2, 0.15], dtype=numpy.float32)   # the reference test's own inputs
    (z1, o1) = ref_stats.count_binary_decisions(abs1, 0.05, 7)
    g.update(cbd1_in=abs1, cbd1_zeros=z1, cbd1_ones=o1)
    abs2 = numpy.array([210., 6., 9., 6.], dtype=numpy.float32)
    (z2, o2) = ref_stats.count_binary_decisions(abs2, 3., 7)
    g.update(cbd2_in=abs2, cbd2_zeros=z2, cbd2_ones=o2)
    mean = y.mean(axis=(0, 1, 2)).astype(numpy.float32)
    g.update(lat_mean=mean, lat_binary_probabilities=ref_stats.compute_binary_probabilities(y, bw2, mean, 10))
    # rescale_compress_lossless_maps through the reference's own Python on the real C++ coder
    probs_path = os.path.join(REF, 'lossless/results/1_10000/training_index_10/binary_probabilities_1.npy')
    centered = y - numpy.tile(mean, (2, 8, 12, 1))
    ccq = tls.quantize_per_map(centered, bw2)
    g.update(lossless_cq=ccq, lossless_bits=numpy.array([ref_compression.rescale_compress_lossless_maps(ccq[j], bw2, probs_path, 67) for j in range(2)]),
             lossless_bits_no_exception=numpy.array([ref_compression.rescale_compress_lossless_maps(ccq[j], bw2, probs_path) for j in range(2)]))
    sym = tls.cast_float_to_int16(ccq[0]/numpy.tile(bw2.reshape(1, 1, 128), (8, 12, 1)))
    (rec, nb_each) = ref_compression.compress_lossless_maps(sym, probs_path, 67)
    g.update(lossless_symbols=sym, lossless_rec=rec, lossless_bits_each_map=nb_each)
    # find_index_map_exception (stats.py:197-241) and its pieces, on latents with a near-uniform map (index 41), a map
    # whose values all fall into one unit interval (index 7 -> divergence 1), integer-valued extrema (closed last interval)
    rng2 = numpy.random.RandomState(21)
    ys = (rng2.laplace(size=(3, 6, 8, 128))*rng2.uniform(0.4, 5., size=128)).astype(numpy.float32)
    ys[:, :, :, 41] = rng2.uniform(-6., 6., size=(3, 6, 8)).astype(numpy.float32)
    ys[:, :, :, 7] = rng2.uniform(0.05, 0.95, size=(3, 6, 8)).astype(numpy.float32)
    ys[:, :, :, 12] = rng2.laplace(size=(3, 6, 8)).astype(numpy.float32).clip(-6.5, 8.5)
    ys[0, 0, 0, 12] = numpy.float32(9.)          # max of map 12 is an integer: it belongs to the last interval
    ys[1, 2, 3, 12] = numpy.float32(-7.)         # min of map 12 is an integer
    divergences = numpy.zeros(128)
    for i in range(128):
        probs = ref_stats.compute_probabilities_intervals(ys[:, :, :, i], 1.)[1]
        nz = numpy.extract(probs != 0., probs)
        divergences[i] = tls.jensen_shannon_divergence(nz, (1./nz.size)*numpy.ones(nz.size)) if nz.size > 1 else 1.
    (edges12, probs12) = ref_stats.compute_probabilities_intervals(ys[:, :, :, 12], 1.)
    (edges_h, probs_h) = ref_stats.compute_probabilities_intervals(ys[:, :, :, 3], 0.5)
    p0 = numpy.array([0.1, 0.2, 0.3, 0.4])
    p1 = numpy.array([0.25, 0.25, 0.25, 0.25])
    g.update(stats_y=ys, stats_divergences=divergences, stats_idx=numpy.int64(ref_stats.find_index_map_exception(ys)),
             stats_edges12=edges12, stats_probs12=probs12, stats_edges_half=edges_h, stats_probs_half=probs_h,
             js_p0=p0, js_p1=p1, js_out=numpy.float64(tls.jensen_shannon_divergence(p0, p1)))
    # compute_bjontegaard (tools.py:157-263) on two rate-distortion curves
    bd_r0 = numpy.array([0.12, 0.25, 0.48, 0.91, 1.43]); bd_p0 = numpy.array([27.1, 29.8, 32.6, 35.9, 38.2])
    bd_r1 = numpy.array([0.10, 0.22, 0.41, 0.80, 1.31, 1.9]); bd_p1 = numpy.array([27.4, 30.1, 32.7, 36.0, 38.6, 40.3])
    g.update(bd_r0=bd_r0, bd_p0=bd_p0, bd_r1=bd_r1, bd_p1=bd_p1, bd_out=numpy.float64(tls.compute_bjontegaard(bd_r0, bd_p0, bd_r1, bd_p1)))
    # rgb_to_ycbcr (tools.py:1019-1083) on a random picture plus the corners of the RGB cube
    rgb = numpy.random.RandomState(33).randint(0, 256, size=(24, 32, 3)).astype(numpy.uint8)
    rgb[0, :8] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255]]
    g.update(rgb_in=rgb, rgb_out=tls.rgb_to_ycbcr(rgb))
    numpy.savez_compressed(os.path.join(OUT, 'tools_golden.npz'), **g)
    print('tools_golden.npz: rate', g['lat_rate'], 'bits', g['lossless_bits'], 'psnr', g['psnr_known'], 'idx exception', g['stats_idx'])


def svhn_golden():
    """BASELINE.json configs[0]: the reference's numpy SVHN entropy autoencoder (svhn/), seeded random initialisation
    (its trained .pkl files are absent from the mount). The 15 MB of parameters are NOT stored: both sides draw them with
    numpy.random.normal in the same order under the same seed; a checksum pins that."""
    import importlib
    for name in [m for m in sys.modules if m.split('.')[0] in ('tools', 'eae', 'lossless', 'svhn')]:
        del sys.modules[name]
    sys.path.remove(REF)
    svhn_root = '/root/reference/svhn'
    sys.path.insert(0, svhn_root)
    ref_eae = importlib.import_module('eae.EntropyAutoencoder')
    ref_utils = importlib.import_module('eae.utils')
    ref_tls = importlib.import_module('tools.tools')
    ref_svhn = importlib.import_module('svhn.svhn')
    g = {}
    numpy.random.seed(20)
    ae = ref_eae.EntropyAutoencoder(3072, 300, 200, 1., 15., False)
    params = ae._EntropyAutoencoder__parameters_eae
    g['seed'] = numpy.int64(20)
    g['param_checksum'] = numpy.array([params['weights_encoder']['l1'].sum(), params['weights_encoder']['latent'].sum(),
                                       params['weights_decoder']['l1'].sum(), params['weights_decoder']['mean'].sum()])
    rng = numpy.random.RandomState(21)
    images = rng.randint(0, 256, size=(3, 3072)).astype(numpy.uint8)
    mean_training = rng.uniform(90., 140., size=(1, 3072))
    std_training = numpy.float64(57.3)
    x = ref_svhn.preprocess_svhn(images, mean_training, std_training)
    (hidden, y) = ae.encoder(x)
    g.update(images=images, mean_training=mean_training, std_training=std_training, preprocessed=x, hidden_encoder=hidden, y=y)
    for (i, bw) in enumerate((1., 0.25)):
        q = ref_tls.quantization(y, bw)
        ent = ref_tls.discrete_entropy(q, bw)
        (hd, rec) = ae.decoder(q)
        rec_u8 = ref_tls.cast_float_to_uint8(rec*std_training + numpy.tile(mean_training, (3, 1)))
        psnr = ref_tls.mean_psnr(images, rec_u8)
        g['bw{}'.format(i)] = numpy.float64(bw)
        g['q{}'.format(i)] = q
        g['entropy{}'.format(i)] = numpy.float64(ent)
        g['rate{}'.format(i)] = numpy.float64(200*ent/3072)
        g['reconstruction{}'.format(i)] = rec
        g['rec_u8_{}'.format(i)] = rec_u8
        g['psnr{}'.format(i)] = numpy.float64(psnr)
        g['count_symbols{}'.format(i)] = ref_tls.count_symbols(q, bw)
    # batch = 1 (the config's own batch size) through the reference's compute_rate_psnr
    os.makedirs('/tmp/eae_golden', exist_ok=True)
    (rate1, psnr1) = ref_utils.compute_rate_psnr(images[:1], mean_training, std_training, ae, 1., 1, '/tmp/eae_golden/rec.png')
    g.update(rate_batch1=numpy.float64(rate1), psnr_batch1=numpy.float64(psnr1))
    halves = numpy.array([-0.5, 0.5, 1.5, 2.5, 254.5, 255.5, 300., -3., 17.49999], dtype=numpy.float64)
    g.update(u8_in=halves, u8_out=ref_tls.cast_float_to_uint8(halves), lrelu_in=halves - 2., lrelu_out=ref_tls.leaky_relu(halves - 2.))
    numpy.savez_compressed(os.path.join(OUT, 'svhn_golden.npz'), **g)
    print('svhn_golden.npz: rate', g['rate0'], g['rate1'], 'psnr', g['psnr0'], g['psnr1'], 'batch1', rate1, psnr1)


if __name__ == '__main__':
    if not os.path.isdir(REF):
        raise SystemExit('/root/reference is not mounted: the fixtures can only be generated in the build container.')
    os.makedirs(OUT, exist_ok=True)
    (ref, tls, ref_compression, ref_stats) = import_reference()
    coder_golden(ref)
    tools_golden(tls, ref_compression, ref_stats, OUT)
    svhn_golden()
