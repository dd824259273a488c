# tfutils.gdn / tfutils.inverse_gdn as device ops on numpy arrays (tfutils.py:363-397, 480-509)
from autoencoder_based_image_compression_amd.kodak.tfutils.tfutils import gdn, initialize_weights_gdn, inverse_gdn  # noqa: F401
