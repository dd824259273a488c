from autoencoder_based_image_compression_amd.kodak.tf_shim import *  # noqa: F401,F403
from autoencoder_based_image_compression_amd.kodak.tf_shim import Session, reset_default_graph, Placeholder, Node, __version__  # noqa: F401
