"""GPU parity of the SVHN path (BASELINE.json configs[0], batch = 1 and 3): the mirrored modules against the outputs of
the reference's own numpy code (tests/golden/svhn_golden.npz) and, exactly, against the C oracle."""
import os

import numpy
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'svhn_golden.npz')


@pytest.fixture(scope='module')
def gold():
    with numpy.load(GOLD) as data:
        return {k: data[k] for k in data.files}


@pytest.fixture(scope='module')
def ae(gold):
    from autoencoder_based_image_compression_amd.svhn.eae.EntropyAutoencoder import EntropyAutoencoder
    numpy.random.seed(int(gold['seed']))
    return EntropyAutoencoder(3072, 300, 200, 1., 15., False)


def _close(a, b, rel=1e-12):
    return numpy.abs(a - b).max() <= rel*max(1., numpy.abs(b).max())


def test_preprocess_encoder_decoder(gold, ae):
    from autoencoder_based_image_compression_amd.svhn.svhn import svhn
    from oracle import svhn as orc
    x = svhn.preprocess_svhn(gold['images'], gold['mean_training'], gold['std_training'])
    assert x.dtype == numpy.float64 and numpy.array_equal(x, gold['preprocessed'])       # element-wise: exact
    (hidden, y) = ae.encoder(x)
    (hidden_o, y_o) = orc.encoder(x, ae.get_parameters())
    assert numpy.array_equal(hidden, hidden_o) and numpy.array_equal(y, y_o)              # same FMA chain: exact
    assert _close(hidden, gold['hidden_encoder']) and _close(y, gold['y'])                # vs numpy.dot: 1e-12
    for i in (0, 1):
        (hd, rec) = ae.decoder(gold['q{}'.format(i)])
        (hd_o, rec_o) = orc.decoder(gold['q{}'.format(i)], ae.get_parameters())
        assert numpy.array_equal(rec, rec_o) and _close(rec, gold['reconstruction{}'.format(i)])
    with pytest.raises(TypeError):
        svhn.preprocess_svhn(gold['images'].astype(numpy.float64), gold['mean_training'], gold['std_training'])
    with pytest.raises(ValueError):
        svhn.preprocess_svhn(gold['images'][0], gold['mean_training'], gold['std_training'])


def test_tools(gold):
    from autoencoder_based_image_compression_amd.svhn.tools import tools as tls
    for i in (0, 1):
        bw = float(gold['bw{}'.format(i)])
        q = tls.quantization(gold['y'], bw)
        assert numpy.array_equal(q, gold['q{}'.format(i)])
        assert numpy.array_equal(tls.count_symbols(q, bw), gold['count_symbols{}'.format(i)])
        assert tls.discrete_entropy(q, bw) == gold['entropy{}'.format(i)]
        assert tls.mean_psnr(gold['images'], gold['rec_u8_{}'.format(i)]) == gold['psnr{}'.format(i)]
    assert numpy.array_equal(tls.cast_float_to_uint8(gold['u8_in']), gold['u8_out'])
    with pytest.raises(AssertionError):
        tls.discrete_entropy(gold['y'], 1.)                      # "The quantization was omitted."
    with pytest.raises(ValueError):
        tls.quantization(gold['y'], 0.)
    with pytest.raises(TypeError):
        tls.quantization(gold['images'], 1.)
    with pytest.raises(ValueError):
        tls.mean_psnr(gold['images'], gold['images'])            # MSE == 0


def test_compute_rate_psnr_config0(gold, ae):
    """svhn/eae/utils.py:8-80 on batch = 1 (the configuration BASELINE.json names) and batch = 3."""
    from autoencoder_based_image_compression_amd.svhn.eae import utils
    (rate, psnr) = utils.compute_rate_psnr(gold['images'][:1], gold['mean_training'], gold['std_training'], ae, 1., 1, None)
    assert rate == gold['rate_batch1'] and psnr == gold['psnr_batch1']
    for i in (0, 1):
        (rate, psnr, rec_u8) = utils.compute_rate_psnr(gold['images'], gold['mean_training'], gold['std_training'], ae,
                                                       float(gold['bw{}'.format(i)]), 1, None, return_reconstruction=True)
        assert numpy.array_equal(rec_u8, gold['rec_u8_{}'.format(i)])
        assert rate == gold['rate{}'.format(i)] and psnr == gold['psnr{}'.format(i)]
