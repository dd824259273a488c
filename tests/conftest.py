import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def test_library(monkeypatch):
    """Makes lib/libeae_hip_test.so -- the product's sources compiled with -DEAE_TEST_HOOKS -DEAE_EXPERIMENTAL_CODER (csrc/Makefile) --
    the library every call of the package goes to while the test runs: for the tests that need the eae_hip_debug_* hooks or the
    experimental coder round trips, which the product library does not export (tests/test_abi.py). Every other test runs on
    lib/libeae_hip.so; `EAE_HIP_LIB=test python -m pytest tests -m gpu` runs ALL of them on the test build."""
    from autoencoder_based_image_compression_amd import _native
    lib = _native.hip_test()
    monkeypatch.setattr(_native, '_hip', lib)
    return lib


class _LaunchOptions(object):
    """Kernel-form overrides for the parity tests of every form: the library reads EAE_HIP_* once at load (csrc/hip/misc.hip), so a
    test that changes them inside the process has the library read them again (eae_hip_debug_reload_launch_options); the
    hand-off fault injection has no environment variable at all (eae_hip_debug_set_split_mute). Both hooks exist in the test build
    only (fixture `test_library`)."""
    NAMES = ('EAE_HIP_GEMM', 'EAE_HIP_SPLIT_WAVES', 'EAE_HIP_SPLIT_WPB', 'EAE_HIP_FORCE_TILE', 'EAE_HIP_FORCE_NT', 'EAE_HIP_LATENT', 'EAE_HIP_LATENT_LDS',
             'EAE_HIP_ASSUME_PARTITIONED', 'EAE_HIP_PACK')

    def __init__(self, monkeypatch, lib):
        self._mp = monkeypatch
        self._lib = lib

    def setenv(self, name, value):
        assert name in self.NAMES, name
        self._mp.setenv(name, value)
        assert self._lib.eae_hip_debug_reload_launch_options() == 0

    def delenv(self, name, raising=False):
        assert name in self.NAMES, name
        self._mp.delenv(name, raising=raising)
        assert self._lib.eae_hip_debug_reload_launch_options() == 0

    def clear(self):
        for name in self.NAMES:
            self._mp.delenv(name, raising=False)
        assert self._lib.eae_hip_debug_reload_launch_options() == 0

    def split_mute(self, on):
        assert self._lib.eae_hip_debug_set_split_mute(1 if on else 0) == 0


@pytest.fixture
def launch_options(monkeypatch, test_library):
    options = _LaunchOptions(monkeypatch, test_library)
    options.clear()
    yield options
    options.split_mute(False)
    options.clear()
