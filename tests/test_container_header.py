"""The container header is untrusted input: every field that sizes a device buffer or a copy is checked on the host before
anything reaches the GPU (an inflated per-map bit count with a payload sized to match made the unpacking
kernel write past the map's stream region). No GPU needed: `read_header` is pure host code."""
import numpy
import pytest

from autoencoder_based_image_compression_amd import container


def build_blob(nb_images=1, height=32, width=48, nb_maps=128, length=10, idx_map_exception=67, bits=None, version=container.VERSION,
               payload_extra=0):
    head = container._HEADER.pack(container.MAGIC, version, 0, nb_images, height, width, nb_maps, length, 0, idx_map_exception)
    nb_rows = nb_images if idx_map_exception >= 0 else 0
    if bits is None:
        bits = numpy.full((nb_images*nb_maps, 2), 9, dtype=numpy.uint32)
    payload = int(((bits.astype(numpy.int64) + 7)//8).sum()) + payload_extra
    return b''.join([head, numpy.ones(nb_maps, numpy.float32).tobytes(), numpy.zeros(nb_maps, numpy.float32).tobytes(),
                     numpy.full((nb_maps, length), 0.5).tobytes(), numpy.full((nb_rows, length), 0.5).tobytes(),
                     bits.astype(numpy.uint32).tobytes(), bytes(payload)])


def test_well_formed_header_is_accepted():
    header = container.read_header(build_blob())
    assert (header['nb_images'], header['height'], header['width'], header['nb_maps']) == (1, 32, 48, 128)
    assert header['bits'].shape == (128, 2) and header['exception_probabilities'].shape == (1, 10)
    assert container.read_header(build_blob(idx_map_exception=-1))['exception_probabilities'].shape == (0, 10)


def test_capacity_matches_the_coder_library():
    """The pure-Python capacity equals the host library's (the reference's compression.cpp:24 rule)."""
    from autoencoder_based_image_compression_amd import _native
    lib = _native.coder()
    for (map_size, length) in ((6, 10), (1536, 10), (16384, 10), (7, 40), (1, 1), (1536, 255)):
        assert container.stream_capacity_bits(map_size, length) == 8*int(lib.eae_coder_stream_capacity_bytes(map_size, length))


@pytest.mark.parametrize('kwargs', [
    dict(nb_maps=64), dict(nb_maps=129), dict(length=0), dict(height=0), dict(width=0), dict(nb_images=0), dict(height=40),
    dict(width=50), dict(idx_map_exception=128), dict(idx_map_exception=-2), dict(version=container.VERSION + 1), dict(payload_extra=1),
])
def test_bad_sizes_are_rejected(kwargs):
    with pytest.raises(ValueError):
        container.read_header(build_blob(**kwargs))


@pytest.mark.parametrize('which', [(0, 0), (127, 1), (60, 0)])
def test_inflated_bit_count_with_matching_payload_is_rejected(which):
    """One count above the capacity of a stream (6 symbols * 32 bits = 192 bits for a 32x48 image), payload sized to match."""
    bits = numpy.full((128, 2), 9, dtype=numpy.uint32)
    bits[which] = container.stream_capacity_bits(6, 10)
    container.read_header(build_blob(bits=bits))              # exactly the capacity: legal
    bits[which] += 1
    with pytest.raises(ValueError):
        container.read_header(build_blob(bits=bits))
    bits[which] = 0xFFFFFFF0
    with pytest.raises(ValueError):
        container.read_header(build_blob(bits=bits)[:4096])   # and no 500 MB payload is needed to find out
