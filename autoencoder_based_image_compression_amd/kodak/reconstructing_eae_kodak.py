"""The rate-distortion harness of the reference (kodak_tensorflow/reconstructing_eae_kodak.py) on the mirrored surface.

`fix_gamma` :31-243 and `vary_gamma_fix_bin_widths` :401-556 with the reference's arguments, written against exactly
the calls the reference makes (`EntropyAutoencoder`, `IsolatedDecoder`, `tf.Session`, `eae.batching.*`, `tls.*`,
`lossless.compression.rescale_compress_lossless_maps`), so it doubles as the proof that the reference's own script
drops onto this package. With `path_to_checking_r` the reconstructions and their crops are written as PNG
like the reference does (:227-232, :550-555). `write_reference` :558-591 and `evaluate_cached` (the `.npy` result cache and
the Bjontegaard dictionary of the reference's `__main__`, :593-760, :809-860) complete the entropy-autoencoder side.
Out of scope here: matplotlib plots and running the JPEG2000 / HEVC baselines (external codecs).
"""
import os
import pickle

import numpy

from . import tf_shim as tf
from .eae import batching
from .eae.graph.EntropyAutoencoder import EntropyAutoencoder
from .eae.graph.IsolatedDecoder import IsolatedDecoder
from .lossless import compression
from .tools import tools as tls


def fix_gamma(reference_uint8, bin_width_init, multipliers, idx_training, gamma_scaling, batch_size,
              are_bin_widths_learned, is_lossless, path_to_checking_r=None, list_rotation=None, positions_top_left=None,
              root='.', return_nb_deads=False):
    """Rate and PSNR of one trained entropy autoencoder at several multiples of its bin widths (:31-243).

    reference_uint8 : uint8 (nb_images, h_in, w_in). multipliers : float32 (nb_points,).
    Files are looked up like the reference does, relative to `root`:
      eae/results/<suffix>/nb_itvs_per_side_<idx>.pkl, eae/results/<suffix>/model_<idx>.ckpt (a TF checkpoint, or a sibling .npz),
      lossless/results/<suffix>/training_index_<idx>/{map_mean.npy, idx_map_exception.pkl, binary_probabilities_<m>.npy}.
    Returns (rate, psnr) float64 (nb_points, nb_images) [+ array_nb_deads int32 when `return_nb_deads`].
    """
    nb_points = multipliers.size
    (nb_images, h_in, w_in) = reference_uint8.shape
    rate = numpy.zeros((nb_points, nb_images))
    psnr = numpy.zeros((nb_points, nb_images))
    if are_bin_widths_learned:
        suffix = 'learning_bw_{0}_{1}'.format(tls.float_to_str(bin_width_init), tls.float_to_str(gamma_scaling))
    else:
        suffix = '{0}_{1}'.format(tls.float_to_str(bin_width_init), tls.float_to_str(gamma_scaling))
    path_to_nb_itvs_per_side_load = os.path.join(root, 'eae/results/{0}/nb_itvs_per_side_{1}.pkl'.format(suffix, idx_training))
    path_to_restore = os.path.join(root, 'eae/results/{0}/model_{1}.ckpt'.format(suffix, idx_training))
    path_to_stats = os.path.join(root, 'lossless/results/{0}/training_index_{1}/'.format(suffix, idx_training))
    path_to_map_mean = os.path.join(path_to_stats, 'map_mean.npy')

    entropy_ae = EntropyAutoencoder(batch_size, h_in, w_in, bin_width_init, gamma_scaling, path_to_nb_itvs_per_side_load,
                                    are_bin_widths_learned)
    with tf.Session() as sess:
        entropy_ae.initialization(sess, path_to_restore)
        y_float32 = batching.encode_mini_batches(numpy.expand_dims(reference_uint8, axis=3), sess, entropy_ae, batch_size)
        bin_widths = entropy_ae.get_bin_widths()
    tf.reset_default_graph()

    isolated_decoder = IsolatedDecoder(batch_size, h_in, w_in, are_bin_widths_learned)
    array_nb_deads = numpy.zeros((nb_points, nb_images), dtype=numpy.int32)
    map_mean = numpy.load(path_to_map_mean)
    tiled_map_mean = numpy.tile(map_mean, (nb_images, y_float32.shape[1], y_float32.shape[2], 1))
    if is_lossless:
        with open(os.path.join(path_to_stats, 'idx_map_exception.pkl'), 'rb') as file:
            idx_map_exception = pickle.load(file)
    centered_y_float32 = y_float32 - tiled_map_mean
    with tf.Session() as sess:
        isolated_decoder.initialization(sess, path_to_restore)
        for i in range(nb_points):
            multiplier = multipliers[i].item()
            str_multiplier = tls.float_to_str(multiplier)
            bin_widths_test = multiplier*bin_widths
            centered_quantized_y_float32 = tls.quantize_per_map(centered_y_float32, bin_widths_test)
            array_nb_deads[i, :] = tls.count_nb_deads(centered_quantized_y_float32)
            off_centered_quantized_y_float32 = centered_quantized_y_float32 + tiled_map_mean
            expanded_reconstruction_uint8 = batching.decode_mini_batches(off_centered_quantized_y_float32, sess,
                                                                         isolated_decoder, batch_size)
            reconstruction_uint8 = numpy.squeeze(expanded_reconstruction_uint8, axis=3)
            if is_lossless:
                path_to_binary_probabilities = os.path.join(path_to_stats, 'binary_probabilities_{}.npy'.format(str_multiplier))
            for j in range(nb_images):
                if is_lossless:
                    nb_bits = compression.rescale_compress_lossless_maps(centered_quantized_y_float32[j, :, :, :],
                                                                         bin_widths_test,
                                                                         path_to_binary_probabilities,
                                                                         idx_map_exception=idx_map_exception)
                    rate[i, j] = float(nb_bits)/(h_in*w_in)
                else:
                    rate[i, j] = tls.rate_3d(centered_quantized_y_float32[j, :, :, :], bin_widths_test, h_in, w_in)
                psnr[i, j] = tls.psnr_2d(reference_uint8[j, :, :], reconstruction_uint8[j, :, :])
                if path_to_checking_r is not None:
                    # the reference's PNG dumps (:227-232): the reconstruction, rotated for the portrait images, and its crops
                    path_to_storage = os.path.join(path_to_checking_r, 'reconstruction_fix_gamma', suffix,
                                                   'lossless' if is_lossless else 'approx', 'multiplier_{}'.format(str_multiplier))
                    os.makedirs(path_to_storage, exist_ok=True)
                    paths = [os.path.join(path_to_storage, 'reconstruction_{}.png'.format(j))]
                    paths += [os.path.join(path_to_storage, 'reconstruction_{0}_crop_{1}.png'.format(j, index_crop))
                              for index_crop in range(positions_top_left.shape[1])]
                    tls.visualize_rotated_luminance(reconstruction_uint8[j, :, :], j in list_rotation, positions_top_left, paths)
    tf.reset_default_graph()
    if return_nb_deads:
        return (rate, psnr, array_nb_deads)
    return (rate, psnr)


def fix_gamma_batched(reference_uint8, bin_width_init, multipliers, idx_training, gamma_scaling, batch_size,
                      are_bin_widths_learned, root='.', return_nb_deads=False):
    """`fix_gamma(..., is_lossless=True)` through `codec.BatchCodec`: same files, same returned arrays (equal element for
    element, tests/test_gpu_surface.py), but every rate point is a handful of asynchronous launches per mini-batch instead
    of one `sess.run` + numpy pass + 127 coder calls per image: the latents, symbols and streams never leave HBM."""
    import torch

    from .. import codec
    from .eae.graph import variables as var
    nb_points = multipliers.size
    (nb_images, h_in, w_in) = reference_uint8.shape
    if reference_uint8.dtype != numpy.uint8:
        raise TypeError('`luminances_uint8.dtype` is not equal to `numpy.uint8`.')     # eae/batching.py:86-87
    tls.subdivide_set(nb_images, batch_size)                                            # ValueError like encode_mini_batches
    if are_bin_widths_learned:
        suffix = 'learning_bw_{0}_{1}'.format(tls.float_to_str(bin_width_init), tls.float_to_str(gamma_scaling))
    else:
        suffix = '{0}_{1}'.format(tls.float_to_str(bin_width_init), tls.float_to_str(gamma_scaling))
    path_to_restore = os.path.join(root, 'eae/results/{0}/model_{1}.ckpt'.format(suffix, idx_training))
    path_to_stats = os.path.join(root, 'lossless/results/{0}/training_index_{1}/'.format(suffix, idx_training))
    variables = var.restore_variables(path_to_restore, are_bin_widths_learned)       # IOError if absent
    bin_widths = variables[var.BIN_WIDTHS_NAME]
    map_mean = numpy.load(os.path.join(path_to_stats, 'map_mean.npy'))
    with open(os.path.join(path_to_stats, 'idx_map_exception.pkl'), 'rb') as file:
        idx_map_exception = pickle.load(file)
    rate = numpy.zeros((nb_points, nb_images))
    psnr = numpy.zeros((nb_points, nb_images))
    array_nb_deads = numpy.zeros((nb_points, nb_images), dtype=numpy.int32)
    images = torch.from_numpy(numpy.ascontiguousarray(reference_uint8)).cuda()
    for i in range(nb_points):
        multiplier = multipliers[i].item()
        binary_probabilities = compression.load_binary_probabilities(
            os.path.join(path_to_stats, 'binary_probabilities_{}.npy'.format(tls.float_to_str(multiplier))))
        with codec.BatchCodec(variables, are_bin_widths_learned, multiplier*bin_widths, map_mean, binary_probabilities,
                              idx_map_exception, batch_size, h_in, w_in) as batch_codec:
            tickets = [batch_codec.submit(images[lo:lo + batch_size]) for lo in range(0, nb_images, batch_size)]
            for (k, ticket) in enumerate(tickets):
                values = ticket.result()      # raises what the image-by-image path raises; the codec is closed either way
                sl = slice(k*batch_size, (k + 1)*batch_size)
                rate[i, sl] = values['nb_bits'].astype(numpy.float64)/(h_in*w_in)
                psnr[i, sl] = [tls.psnr_from_sse(int(v), h_in*w_in) for v in values['sse']]
                array_nb_deads[i, sl] = values['nb_deads']
    if return_nb_deads:
        return (rate, psnr, array_nb_deads)
    return (rate, psnr)


def vary_gamma_fix_bin_widths(reference_uint8, bin_width_init, idxs_training, gammas_scaling, batch_size,
                              path_to_checking_r=None, list_rotation=None, positions_top_left=None, root='.'):
    """Rate and PSNR of several entropy autoencoders, each trained with a different scaling coefficient (:401-556)."""
    nb_points = gammas_scaling.size
    if idxs_training.size != nb_points:
        raise ValueError('`gammas_scaling.size` is not equal to `idxs_training.size`.')
    (nb_images, h_in, w_in) = reference_uint8.shape
    rate = numpy.zeros((nb_points, nb_images))
    psnr = numpy.zeros((nb_points, nb_images))
    for i in range(nb_points):
        gamma_scaling = gammas_scaling[i].item()
        idx_training = idxs_training[i].item()
        suffix = '{0}_{1}'.format(tls.float_to_str(bin_width_init), tls.float_to_str(gamma_scaling))
        path_to_nb_itvs_per_side_load = os.path.join(root, 'eae/results/{0}/nb_itvs_per_side_{1}.pkl'.format(suffix, idx_training))
        path_to_restore = os.path.join(root, 'eae/results/{0}/model_{1}.ckpt'.format(suffix, idx_training))
        entropy_ae = EntropyAutoencoder(batch_size, h_in, w_in, bin_width_init, gamma_scaling, path_to_nb_itvs_per_side_load, False)
        with tf.Session() as sess:
            entropy_ae.initialization(sess, path_to_restore)
            y_float32 = batching.encode_mini_batches(numpy.expand_dims(reference_uint8, axis=3), sess, entropy_ae, batch_size)
            bin_widths = entropy_ae.get_bin_widths()
        tf.reset_default_graph()
        isolated_decoder = IsolatedDecoder(batch_size, h_in, w_in, False)
        quantized_y_float32 = tls.quantize_per_map(y_float32, bin_widths)
        with tf.Session() as sess:
            isolated_decoder.initialization(sess, path_to_restore)
            expanded_reconstruction_uint8 = batching.decode_mini_batches(quantized_y_float32, sess, isolated_decoder, batch_size)
        reconstruction_uint8 = numpy.squeeze(expanded_reconstruction_uint8, axis=3)
        tf.reset_default_graph()
        for j in range(nb_images):
            rate[i, j] = tls.rate_3d(quantized_y_float32[j, :, :, :], bin_widths, h_in, w_in)
            psnr[i, j] = tls.psnr_2d(reference_uint8[j, :, :], reconstruction_uint8[j, :, :])
            if path_to_checking_r is not None:                                           # the PNG dumps of :491-495, :550-555
                path_to_storage = os.path.join(path_to_checking_r, 'reconstruction_vary_gamma_fix_bin_widths', suffix)
                os.makedirs(path_to_storage, exist_ok=True)
                paths = [os.path.join(path_to_storage, 'reconstruction_{}.png'.format(j))]
                paths += [os.path.join(path_to_storage, 'reconstruction_{0}_crop_{1}.png'.format(j, index_crop))
                          for index_crop in range(positions_top_left.shape[1])]
                tls.visualize_rotated_luminance(reconstruction_uint8[j, :, :], j in list_rotation, positions_top_left, paths)
    return (rate, psnr)


def write_reference(reference_uint8, path_to_checking_r, list_rotation, positions_top_left):
    """Writes the luminance images and their crops under `<path_to_checking_r>/reference/` (:558-591)."""
    os.makedirs(os.path.join(path_to_checking_r, 'reference'), exist_ok=True)
    for i in range(reference_uint8.shape[0]):
        paths = [os.path.join(path_to_checking_r, 'reference/reference_{}.png'.format(i))]
        paths += [os.path.join(path_to_checking_r, 'reference/reference_{0}_crop_{1}.png'.format(i, index_crop))
                  for index_crop in range(positions_top_left.shape[1])]
        tls.visualize_rotated_luminance(reference_uint8[i, :, :], i in list_rotation, positions_top_left, paths)


# The three entropy-autoencoder experiments of the reference's `__main__` (:609-626).
DICT_VARY_GAMMA_FIX_BIN_WIDTHS = {
    'bin_width_init': 1.,
    'idxs_training': numpy.array([10, 10, 10, 10, 10, 7, 6], dtype=numpy.int32),
    'gammas_scaling': numpy.array([10000., 12000., 16000., 24000., 40000., 72000., 96000.])
}
DICT_FIX_GAMMA_LEARN_BIN_WIDTHS = {
    'bin_width_init': 0.5,
    'multipliers': numpy.array([1., 1.25, 1.5, 2., 3., 4., 6., 8., 10.], dtype=numpy.float32),
    'idx_training': 10,
    'gamma_scaling': 10000.
}
DICT_FIX_GAMMA_FIX_BIN_WIDTHS = {
    'bin_width_init': 1.,
    'multipliers': numpy.array([1., 1.25, 1.5, 2., 3., 4., 6., 8., 10.], dtype=numpy.float32),
    'idx_training': 10,
    'gamma_scaling': 10000.
}


def _cached(path_to_rate, path_to_psnr, label, compute, verbose):
    """The reference's result cache (:674-760): two `.npy` files per curve; present -> loaded, absent -> computed and saved."""
    if os.path.isfile(path_to_rate) and os.path.isfile(path_to_psnr):
        if verbose:
            print('For {0}, the rates at "{1}" and the PSNRs at "{2}" are loaded.'.format(label, path_to_rate, path_to_psnr))
            print('Delete them manually to re-compute them.')
        return (numpy.load(path_to_rate), numpy.load(path_to_psnr))
    if verbose:
        print('For {}, the rates and the PSNRs are computed.'.format(label))
    (rate, psnr) = compute()
    numpy.save(path_to_rate, rate)
    numpy.save(path_to_psnr, psnr)
    return (rate, psnr)


def evaluate_cached(reference_uint8, path_to_checking_r, list_rotation, positions_top_left, code_lossless, batch_size=4,
                    root='.', write_ref=False, dump_images=True, batched=False, verbose=True,
                    dict_vary_gamma_fix_bin_widths=None, dict_fix_gamma_learn_bin_widths=None,
                    dict_fix_gamma_fix_bin_widths=None):
    """The entropy-autoencoder part of the reference's `__main__` (:593-760, :809-860) with its on-disk result layout.

    Under `path_to_checking_r` (the reference uses 'eae/visualization/test/checking_reconstructing/<kodak|bsds>'):
      rate_vary_gamma_fix_bin_widths.npy, psnr_vary_gamma_fix_bin_widths.npy               (orange curve)
      rate_fix_gamma_learn_bin_widths_<code>.npy, psnr_fix_gamma_learn_bin_widths_<code>.npy (green curve)
      rate_fix_gamma_fix_bin_widths_<code>.npy, psnr_fix_gamma_fix_bin_widths_<code>.npy     (red curve)
    with <code> = 'lossless' if `code_lossless` else 'approx'. A curve whose two files exist is loaded, not recomputed.
    When `rate_jpeg2000.npy` / `psnr_jpeg2000.npy` and `rate_hevc.npy` / `psnr_hevc.npy` are there too (the reference
    makes them with external codecs, out of scope here), `dictionary_bjontegaard_<code>.pkl` is written like :839-860.
    `batched` routes the lossless `fix_gamma` curves through `fix_gamma_batched` (same arrays, everything in HBM; no PNG
    dumps). Returns a dict with the six arrays, the mean curves and the Bjontegaard dictionary (or None).
    """
    vary = dict_vary_gamma_fix_bin_widths or DICT_VARY_GAMMA_FIX_BIN_WIDTHS
    learn = dict_fix_gamma_learn_bin_widths or DICT_FIX_GAMMA_LEARN_BIN_WIDTHS
    fix = dict_fix_gamma_fix_bin_widths or DICT_FIX_GAMMA_FIX_BIN_WIDTHS
    os.makedirs(path_to_checking_r, exist_ok=True)
    if write_ref:
        write_reference(reference_uint8, path_to_checking_r, list_rotation, positions_top_left)
    dumps = (path_to_checking_r, list_rotation, positions_top_left) if dump_images else (None, None, None)
    str_code = 'lossless' if code_lossless else 'approx'

    def path(name):
        return os.path.join(path_to_checking_r, name)

    def run_fix_gamma(config, are_bin_widths_learned):
        if batched and code_lossless:
            return fix_gamma_batched(reference_uint8, config['bin_width_init'], config['multipliers'], config['idx_training'],
                                     config['gamma_scaling'], batch_size, are_bin_widths_learned, root=root)
        return fix_gamma(reference_uint8, config['bin_width_init'], config['multipliers'], config['idx_training'],
                         config['gamma_scaling'], batch_size, are_bin_widths_learned, code_lossless, *dumps, root=root)
    out = {}
    (out['rate_vary_gamma_fix_bin_widths'], out['psnr_vary_gamma_fix_bin_widths']) = _cached(
        path('rate_vary_gamma_fix_bin_widths.npy'), path('psnr_vary_gamma_fix_bin_widths.npy'), 'the orange curve',
        lambda: vary_gamma_fix_bin_widths(reference_uint8, vary['bin_width_init'], vary['idxs_training'], vary['gammas_scaling'],
                                          batch_size, *dumps, root=root), verbose)
    (out['rate_fix_gamma_learn_bin_widths'], out['psnr_fix_gamma_learn_bin_widths']) = _cached(
        path('rate_fix_gamma_learn_bin_widths_{}.npy'.format(str_code)), path('psnr_fix_gamma_learn_bin_widths_{}.npy'.format(str_code)),
        'the green curve', lambda: run_fix_gamma(learn, True), verbose)
    (out['rate_fix_gamma_fix_bin_widths'], out['psnr_fix_gamma_fix_bin_widths']) = _cached(
        path('rate_fix_gamma_fix_bin_widths_{}.npy'.format(str_code)), path('psnr_fix_gamma_fix_bin_widths_{}.npy'.format(str_code)),
        'the red curve', lambda: run_fix_gamma(fix, False), verbose)
    for key in list(out):
        out['mean_' + key] = numpy.mean(out[key], axis=1)                                                       # :809-815
    out['dict_bjontegaard'] = None
    external = [path('{0}_{1}.npy'.format(kind, codec_name)) for codec_name in ('jpeg2000', 'hevc') for kind in ('rate', 'psnr')]
    if all(os.path.isfile(p) for p in external):
        means = {name: numpy.mean(numpy.load(path('{}.npy'.format(name))), axis=1)
                 for name in ('rate_jpeg2000', 'psnr_jpeg2000', 'rate_hevc', 'psnr_hevc')}
        out['dict_bjontegaard'] = {
            '{0}_{1}'.format(curve, codec_name): tls.compute_bjontegaard(out['mean_rate_' + curve], out['mean_psnr_' + curve],
                                                                         means['rate_' + codec_name], means['psnr_' + codec_name])
            for curve in ('fix_gamma_learn_bin_widths', 'fix_gamma_fix_bin_widths') for codec_name in ('jpeg2000', 'hevc')}
        with open(path('dictionary_bjontegaard_{}.pkl'.format(str_code)), 'wb') as file:
            pickle.dump(out['dict_bjontegaard'], file, protocol=2)
    return out
