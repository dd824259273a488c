"""MI355X-native compression inference path of thierrydumas/autoencoder_based_image_compression.

Scope: analysis transform -> GDN -> uniform quantiser -> (histogram/entropy | UEG0 + binary
arithmetic coder) -> IGDN -> synthesis transform -> BT.601 cast -> PSNR, behind the reference's own Python call
surface. The arithmetic lives in two C-ABI libraries built in-tree:

* ``lib/libeae_hip.so``   hand-written gfx950 kernels (include/eae_hip.h)
* ``lib/libeae_coder.so`` host lossless coder (include/eae_coder.h)

``kodak`` mirrors ``kodak_tensorflow/`` of the reference module by module; ``svhn`` mirrors ``svhn/``.
"""
__version__ = '0.1.0'
