"""Pins the SVHN oracle (oracle/svhn_oracle.c) against outputs of the reference's own numpy code
(tests/golden/svhn_golden.npz, made by oracle/gen_golden.py from /root/reference/svhn). Tolerance 1e-12 relative to the
largest magnitude: float64 rounding of K <= 3072-term dot products in a different summation order (numpy.dot is BLAS)."""
import os

import numpy
import pytest

from autoencoder_based_image_compression_amd.svhn.eae.EntropyAutoencoder import EntropyAutoencoder
from oracle import svhn as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'svhn_golden.npz')


@pytest.fixture(scope='module')
def gold():
    with numpy.load(GOLD) as data:
        return {k: data[k] for k in data.files}


@pytest.fixture(scope='module')
def params(gold):
    numpy.random.seed(int(gold['seed']))
    ae = EntropyAutoencoder(3072, 300, 200, 1., 15., False)
    return ae.get_parameters()


def _close(a, b, rel=1e-12):
    return numpy.abs(a - b).max() <= rel*max(1., numpy.abs(b).max())


def test_seeded_parameters_equal_the_reference_initialisation(gold, params):
    """EntropyAutoencoder.py:155-179: same numpy.random.normal calls in the same order."""
    sums = numpy.array([params['weights_encoder']['l1'].sum(), params['weights_encoder']['latent'].sum(),
                        params['weights_decoder']['l1'].sum(), params['weights_decoder']['mean'].sum()])
    assert numpy.array_equal(sums, gold['param_checksum'])
    assert params['weights_encoder']['l1'].shape == (3072, 300) and params['biases_decoder']['mean'].shape == (1, 3072)


def test_oracle_encoder_decoder_match_the_reference(gold, params):
    (hidden, y) = orc.encoder(gold['preprocessed'], params)
    assert _close(hidden, gold['hidden_encoder']) and _close(y, gold['y'])
    for i in (0, 1):
        (_, rec) = orc.decoder(gold['q{}'.format(i)], params)
        assert _close(rec, gold['reconstruction{}'.format(i)])


def test_leaky_relu_and_cast_known_answers(gold):
    from autoencoder_based_image_compression_amd.svhn.tools import tools as tls
    assert numpy.array_equal(tls.leaky_relu(gold['lrelu_in']), gold['lrelu_out'])
    assert list(gold['u8_out']) == [0, 0, 2, 2, 254, 255, 255, 0, 17]   # round half to even, clip to [0, 255]
