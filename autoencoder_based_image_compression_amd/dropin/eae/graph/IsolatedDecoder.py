from autoencoder_based_image_compression_amd.kodak.eae.graph.IsolatedDecoder import IsolatedDecoder  # noqa: F401
