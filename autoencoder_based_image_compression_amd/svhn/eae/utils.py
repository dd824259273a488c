"""`compute_rate_psnr` of the reference (svhn/eae/utils.py:8-80): test-time rate and PSNR of a batch of SVHN digits."""
import numpy

from ..svhn import svhn as svhn_module
from ..tools import tools as tls


def compute_rate_psnr(reference_uint8, mean_training, std_training, entropy_ae, bin_width, nb_vertically=None,
                      path_to_reconstruction=None, return_reconstruction=False):
    """Same arguments as the reference. The PNG mosaic it writes (`tls.visualize_rows`, :75-79) is out of scope:
    `nb_vertically` / `path_to_reconstruction` are accepted and ignored. Returns (rate, psnr)
    [+ reconstruction_uint8 when `return_reconstruction`]."""
    reference_float64 = svhn_module.preprocess_svhn(reference_uint8, mean_training, std_training)
    (nb_images, nb_pixels) = reference_uint8.shape
    y = entropy_ae.encoder(reference_float64)[1]
    quantized_y = tls.quantization(y, bin_width)
    disc_entropy = tls.discrete_entropy(quantized_y, bin_width)
    rate = entropy_ae.nb_y*disc_entropy/nb_pixels
    reconstruction_float64 = entropy_ae.decoder(quantized_y)[1]
    rec_rescaled_float64 = reconstruction_float64*std_training + numpy.tile(mean_training, (nb_images, 1))
    reconstruction_uint8 = tls.cast_float_to_uint8(rec_rescaled_float64)
    psnr = tls.mean_psnr(reference_uint8, reconstruction_uint8)
    if return_reconstruction:
        return (rate, psnr, reconstruction_uint8)
    return (rate, psnr)
