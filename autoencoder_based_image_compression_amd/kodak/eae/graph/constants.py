"""Architecture constants of the entropy autoencoder under the reference's names (kodak_tensorflow/eae/graph/constants.py:
42-59). Only what the inference path reads is here; the training hyper-parameters (:5-41) are out of scope."""

# Lower bound of the random initialisation of the GDN / IGDN weights (constants.py:22; tfutils.initialize_weights_gdn,
# tfutils.py:445-478).
MIN_GAMMA_BETA = 2.e-5

# (feature maps, kernel width, stride) of the three analysis layers; the synthesis transform mirrors them.
_ANALYSIS_LAYERS = ((128, 9, 4), (128, 5, 2), (128, 5, 2))
((NB_MAPS_1, WIDTH_KERNEL_1, STRIDE_1),
 (NB_MAPS_2, WIDTH_KERNEL_2, STRIDE_2),
 (NB_MAPS_3, WIDTH_KERNEL_3, STRIDE_3)) = _ANALYSIS_LAYERS

# A latent feature map is this many times smaller than the image, along each axis.
STRIDE_PROD = 1
for (_, _, _stride) in _ANALYSIS_LAYERS:
    STRIDE_PROD *= _stride
del _stride
