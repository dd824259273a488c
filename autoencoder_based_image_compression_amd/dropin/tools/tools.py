from autoencoder_based_image_compression_amd.kodak.tools.tools import (average_entropies, cast_bt601, cast_float_to_int16, count_nb_deads, count_symbols,  # noqa: F401
                            discrete_entropy, float_to_str, psnr_2d, quantize_per_map, rate_3d, subdivide_set)

_OUT_OF_SCOPE = ('compute_bjontegaard', 'visualize_rotated_luminance', 'plot_graphs', 'histogram', 'read_image_mode',
                 'save_image', 'rgb_to_ycbcr', 'crop_option_2d', 'crop_repeat_2d', 'untar_archive', 'tile_cauchy',
                 'jensen_shannon_divergence', 'kl_divergence', 'normed_histogram', 'visualize_crops',
                 'visualize_luminances', 'visualize_representation', 'visualize_weights', 'clean_sort_list_strings',
                 'convert_approx_entropy', 'expand_all', 'expand_parameters', 'gradient_density_approximation',
                 'loss_density_approximation', 'loss_entropy_reconstruction', 'approximate_entropy', 'approximate_probability',
                 'area_under_piecewise_linear_function', 'differential_entropy', 'index_linear_piece', 'opposite_vlogv',
                 'reshape_4d_to_2d', 'reshape_2d_to_4d')


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError('tools.tools.{} is outside the compression inference path this build replaces '
                                  '(SURVEY.md section 8: plotting / image I/O / training helpers).'.format(name))
    raise AttributeError(name)
