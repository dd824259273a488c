"""Architecture constants of the entropy autoencoder (reference: kodak_tensorflow/eae/graph/constants.py:42-59).

Only the constants the inference path reads are kept; the training hyper-parameters (:5-41) are out of scope.
"""

# `MIN_GAMMA_BETA` bounds the random initialisation of the GDN/IGDN weights (constants.py:22, used by
# tfutils.initialize_weights_gdn, tfutils.py:445-478).
MIN_GAMMA_BETA = 2.e-5

NB_MAPS_1 = 128
NB_MAPS_2 = 128
NB_MAPS_3 = 128
WIDTH_KERNEL_1 = 9
WIDTH_KERNEL_2 = 5
WIDTH_KERNEL_3 = 5
STRIDE_1 = 4
STRIDE_2 = 2
STRIDE_3 = 2

# The height (width) of the latent variable feature maps is `STRIDE_PROD` times smaller than the height (width) of
# the images fed into the entropy autoencoder.
STRIDE_PROD = STRIDE_1*STRIDE_2*STRIDE_3
