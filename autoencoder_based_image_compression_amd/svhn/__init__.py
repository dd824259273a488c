"""Host-side mirror of the hot-path part of the reference's `svhn/` tree (BASELINE.json configs[0]): the float64 fully
connected entropy autoencoder at test time -- `eae.EntropyAutoencoder.encoder/decoder`, `eae.utils.compute_rate_psnr`,
the `tools.tools` helpers it calls and `svhn.svhn.preprocess_svhn`. Training (hand-written backprop), the VAE demo, the
JPEG baselines and plotting of the reference tree are out of scope."""
