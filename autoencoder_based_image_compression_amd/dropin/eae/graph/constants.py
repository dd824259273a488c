from autoencoder_based_image_compression_amd.kodak.eae.graph.constants import *  # noqa: F401,F403
