from autoencoder_based_image_compression_amd.kodak.lossless.stats import compute_binary_probabilities, count_binary_decisions  # noqa: F401
