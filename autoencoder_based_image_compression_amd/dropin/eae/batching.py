from autoencoder_based_image_compression_amd.kodak.eae.batching import *  # noqa: F401,F403
from autoencoder_based_image_compression_amd.kodak.eae.batching import decode_mini_batches, encode_mini_batches  # noqa: F401
