"""Test-time surface of the reference's numpy `EntropyAutoencoder` (svhn/eae/EntropyAutoencoder.py): constructor and
random initialisation (:54-180), `encoder` (:218-247), `decoder` (:249-278). The hand-written backpropagation, the
piecewise linear density model and the gradient checks of the reference class are out of scope (training)."""
import numpy

from ... import device as dev
from ...kodak import _backend as bk


class EntropyAutoencoder(object):
    """Fully connected float64 entropy autoencoder nb_visible -> nb_hidden -> nb_y -> nb_hidden -> nb_visible."""

    def __init__(self, nb_visible, nb_hidden, nb_y, bin_width_init, gamma, is_bin_width_learned, **unused_training_options):
        """Same leading parameters as the reference (:54-56); the training hyper-parameters are accepted and ignored.
        Parameters are drawn with `numpy.random.normal` in the reference's order (:155-179), so the same
        `numpy.random.seed` gives the same model as the reference."""
        self.nb_visible = nb_visible
        self.nb_hidden = nb_hidden
        self.nb_y = nb_y
        self.bin_width = bin_width_init
        self.gamma = gamma
        self.is_bin_width_learned = is_bin_width_learned
        self.set_parameters(self._initialize_parameters_eae())

    def _initialize_parameters_eae(self):
        p = dict()
        p['weights_encoder'] = {
            'l1': numpy.random.normal(loc=0., scale=0.01, size=(self.nb_visible, self.nb_hidden)),
            'latent': numpy.random.normal(loc=0., scale=0.05, size=(self.nb_hidden, self.nb_y))
        }
        p['biases_encoder'] = {'l1': numpy.zeros((1, self.nb_hidden)), 'latent': numpy.zeros((1, self.nb_y))}
        p['weights_decoder'] = {
            'l1': numpy.random.normal(loc=0., scale=0.05, size=(self.nb_y, self.nb_hidden)),
            'mean': numpy.random.normal(loc=0., scale=0.01, size=(self.nb_hidden, self.nb_visible))
        }
        p['biases_decoder'] = {'l1': numpy.zeros((1, self.nb_hidden)), 'mean': numpy.zeros((1, self.nb_visible))}
        return p

    def set_parameters(self, parameters_eae):
        """Installs parameters given in the reference's nested-dict layout (`_EntropyAutoencoder__parameters_eae`)."""
        self._parameters_eae = parameters_eae
        self._device = None

    def get_parameters(self):
        return self._parameters_eae

    def _dev(self):
        if self._device is None:
            p = self._parameters_eae
            self._device = {(g, k): bk.to_device(p[g][k], numpy.float64) for g in p for k in p[g]}
        return self._device

    def encoder(self, visible_units):
        """(hidden_encoder, y) = (LReLU(x W1 + b1), hidden W2 + b2) (:218-247)."""
        d = self._dev()
        x = bk.to_device(visible_units, numpy.float64)
        hidden_encoder = dev.svhn_dense(x, d[('weights_encoder', 'l1')], d[('biases_encoder', 'l1')], True)
        y = dev.svhn_dense(hidden_encoder, d[('weights_encoder', 'latent')], d[('biases_encoder', 'latent')], False)
        return (bk.to_host(hidden_encoder), bk.to_host(y))

    def decoder(self, y_tilde):
        """(hidden_decoder, reconstruction) = (LReLU(y W3 + b3), hidden W4 + b4) (:249-278)."""
        d = self._dev()
        y = bk.to_device(y_tilde, numpy.float64)
        hidden_decoder = dev.svhn_dense(y, d[('weights_decoder', 'l1')], d[('biases_decoder', 'l1')], True)
        reconstruction = dev.svhn_dense(hidden_decoder, d[('weights_decoder', 'mean')], d[('biases_decoder', 'mean')], False)
        return (bk.to_host(hidden_decoder), bk.to_host(reconstruction))
