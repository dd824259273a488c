from autoencoder_based_image_compression_amd.kodak.lossless.compression import compress_lossless_maps, rescale_compress_lossless_maps  # noqa: F401
