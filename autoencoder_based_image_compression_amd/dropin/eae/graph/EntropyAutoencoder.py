from autoencoder_based_image_compression_amd.kodak.eae.graph.EntropyAutoencoder import EntropyAutoencoder  # noqa: F401
