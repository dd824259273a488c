"""MI355X-native compression inference path of thierrydumas/autoencoder_based_image_compression.

Scope: analysis transform -> GDN -> uniform quantiser -> (histogram/entropy | UEG0 + binary
arithmetic coder) -> IGDN -> synthesis transform -> BT.601 cast -> PSNR, behind the reference's own Python call
surface. The arithmetic lives in two C-ABI libraries built in-tree:

* ``lib/libeae_hip.so``   hand-written gfx950 kernels (include/eae_hip.h)
* ``lib/libeae_coder.so`` host lossless coder (include/eae_coder.h)

``kodak`` mirrors ``kodak_tensorflow/`` of the reference module by module; ``svhn`` mirrors ``svhn/``.
"""
__version__ = '0.1.0'

import os as _os
import sys as _sys

# ---- hardware queues -----------------------------------------------------------------------------------------------------------------
# The HIP runtime multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 when the variable is not set), and
# streams that share a queue serialise, busy ones with busy ones. The product mode of `codec.BatchCodec` keeps three transform
# streams and three to eight coder streams busy (plus a feed and a fetch stream): on four queues that was measured to cost up to
# 30 %. The runtime reads the variable once, when it initialises, so the package asks for what its product mode needs HERE, at
# import, as long as that is still possible: the caller has not chosen a value and no HIP call has been made yet. What was found is
# kept in `HW_QUEUES` for `codec.stream_budget()`, which caps a codec's streams to the queues the process really has and says so.
HW_QUEUES_WANTED = 16


def _configure_hw_queues(environ=None, runtime_is_up=None):
    """-> (effective number of hardware queues, who decided: 'caller' | 'package' | 'runtime default')."""
    environ = _os.environ if environ is None else environ
    if runtime_is_up is None:
        torch_module = _sys.modules.get('torch')
        runtime_is_up = bool(torch_module is not None and torch_module.cuda.is_initialized())
    text = environ.get('GPU_MAX_HW_QUEUES')
    if text is not None:
        try:
            return (max(1, int(text)), 'caller')
        except ValueError:
            return (4, 'caller')            # the runtime ignores what it cannot read
    if not runtime_is_up:
        environ['GPU_MAX_HW_QUEUES'] = str(HW_QUEUES_WANTED)
        return (HW_QUEUES_WANTED, 'package')
    return (4, 'runtime default')           # too late to ask: the runtime came up before this package was imported


HW_QUEUES = _configure_hw_queues()
