"""The container and the standalone decoder (autoencoder_based_image_compression_amd/container.py).
There is no reference counterpart to compare bytes with (the reference never serialises its streams), so the bar is
self-consistency at the reference's own quantities: the decoded images equal the in-memory path's reconstruction bit for
bit, the decoded symbols equal the encoder's, and the bit counts in the header are those the reference-shaped
`lossless.compression` functions report for the same maps."""
import os

import numpy
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coder_golden.npz')


@pytest.fixture(scope='module')
def model():
    from autoencoder_based_image_compression_amd import pipeline
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    v = var.random_variables(1., False, seed=4, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
    with numpy.load(GOLD) as g:
        probabilities = g['real_probabilities_1']
    return {'variables': v, 'encoder': pipeline.DeviceEncoder(v, False), 'decoder': pipeline.DeviceDecoder(v, False),
            'probabilities': probabilities}


def in_memory_path(model, images, bin_widths, map_mean):
    from autoencoder_based_image_compression_amd import device as dev
    y = model['encoder'](torch.from_numpy(images).cuda())
    q = dev.quantize_maps(y, torch.from_numpy(bin_widths).cuda(), torch.from_numpy(map_mean).cuda(), want_shifted=True, want_symbols=True)
    (_, rec, _) = model['decoder'](q['shifted'])
    return q['symbols'].cpu().numpy(), rec.cpu().numpy()


@pytest.mark.parametrize('idx_map_exception', [67, -1])
@pytest.mark.parametrize('shape,scale', [((2, 64, 96), 0.5), ((3, 48, 80), 0.05)])
def test_blob_roundtrip_equals_the_in_memory_path(model, shape, scale, idx_map_exception):
    from autoencoder_based_image_compression_amd import container
    from autoencoder_based_image_compression_amd.kodak.lossless import compression
    rng = numpy.random.RandomState(shape[1])
    images = rng.randint(16, 236, size=shape).astype(numpy.uint8)
    bin_widths = numpy.full(128, scale, dtype=numpy.float32)        # the small bin width produces Exp-Golomb escapes
    map_mean = rng.normal(scale=0.1, size=128).astype(numpy.float32)
    (blob, info) = container.encode_images(images, model['encoder'], bin_widths, map_mean, model['probabilities'], idx_map_exception)
    (symbols, reconstruction) = in_memory_path(model, images, bin_widths, map_mean)
    (header, decoded_symbols) = container.decode_symbols(blob)
    assert numpy.array_equal(decoded_symbols.cpu().numpy(), symbols)
    assert numpy.array_equal(container.decode_images(blob, model['decoder']), reconstruction)
    assert (header['nb_images'], header['height'], header['width']) == shape and header['idx_map_exception'] == idx_map_exception
    assert len(blob) == info['header_bytes'] + info['payload_bytes']
    # bit counts of the shared-table maps are those of the reference-shaped coder call on the same symbols
    (_, nb_bits) = compression.code_planar_symbols(symbols, model['probabilities'], idx_map_exception)
    keep = numpy.ones(128, dtype=bool)
    if idx_map_exception >= 0:
        keep[idx_map_exception] = False
        assert (info['nb_bits'][:, idx_map_exception] > 0).all()
    assert numpy.array_equal(info['nb_bits'][:, keep], nb_bits[:, keep])
    # the payload is the streams and nothing else: bits rounded up to bytes per stream
    bits = header['bits'].astype(numpy.int64)
    assert info['payload_bytes'] == int(((bits + 7)//8).sum())


def test_malformed_blobs_and_arguments(model):
    from autoencoder_based_image_compression_amd import container, pipeline
    images = numpy.random.RandomState(0).randint(16, 236, size=(1, 32, 48)).astype(numpy.uint8)
    ones = numpy.ones(128, dtype=numpy.float32)
    (blob, _) = container.encode_images(images[..., None], model['encoder'], ones, 0*ones, model['probabilities'], 67)
    with pytest.raises(ValueError):
        container.read_header(b'XXXX' + blob[4:])
    with pytest.raises(ValueError):
        container.read_header(blob[:-3])
    with pytest.raises(ValueError):
        container.read_header(blob[:20])
    with pytest.raises(TypeError):
        container.encode_images(images.astype(numpy.float32), model['encoder'], ones, 0*ones, model['probabilities'])
    with pytest.raises(ValueError):
        container.encode_images(images, model['encoder'], ones[:5], 0*ones, model['probabilities'])
    learned = pipeline.DeviceDecoder(model['variables'], True)
    with pytest.raises(ValueError):
        container.decode_images(blob, learned)
    # a corrupted payload still decodes to SOMETHING or raises the coder's error; it never crashes the process
    corrupted = bytearray(blob)
    corrupted[-10] ^= 0xFF
    try:
        container.decode_images(bytes(corrupted), model['decoder'])
    except RuntimeError:
        pass


def test_mutated_header_bit_counts(model):
    """Per-map bit counts are untrusted: above a stream's capacity -> ValueError on the host (nothing launched); within the
    capacity but wrong (payload resized to match) -> the unpacking stays inside every map's region, decoding gives
    something or raises the coder's error."""
    from autoencoder_based_image_compression_amd import container
    images = numpy.random.RandomState(3).randint(16, 236, size=(2, 32, 48)).astype(numpy.uint8)
    ones = numpy.ones(128, dtype=numpy.float32)
    (blob, _) = container.encode_images(images, model['encoder'], ones, 0*ones, model['probabilities'], 67)
    header = container.read_header(blob)
    start = header['payload_offset'] - header['bits'].nbytes
    capacity = container.stream_capacity_bits(6, header['truncated_unary_length'])

    def with_bits(bits):
        payload = int(((bits.astype(numpy.int64) + 7)//8).sum())
        body = blob[header['payload_offset']:]
        body = body[:payload] + bytes(max(0, payload - len(body)))
        return blob[:start] + bits.astype(numpy.uint32).tobytes() + body

    for (row, col) in ((0, 0), (255, 1), (100, 0)):
        bits = header['bits'].copy()
        bits[row, col] = capacity + 1
        with pytest.raises(ValueError):
            container.decode_symbols(with_bits(bits))
        bits[row, col] = capacity                     # legal size, wrong content
        try:
            container.decode_images(with_bits(bits), model['decoder'])
        except RuntimeError:
            pass
    # the GPU is still in working order afterwards
    assert numpy.array_equal(container.decode_images(blob, model['decoder']), in_memory_path(model, images, ones, 0*ones)[1])


def test_full_kodak_image(model):
    """BASELINE.json configs[1] size: one 512x768 image through the file and back."""
    from autoencoder_based_image_compression_amd import container
    import bench
    images = bench.synthetic_images(7, 1, 512, 768)
    ones = numpy.ones(128, dtype=numpy.float32)
    (blob, info) = container.encode_images(images, model['encoder'], ones, 0*ones, model['probabilities'], 67)
    (_, reconstruction) = in_memory_path(model, images, ones, 0*ones)
    assert numpy.array_equal(container.decode_images(blob, model['decoder']), reconstruction)
    assert info['payload_bytes']*8 < 1.1*int(info['nb_bits'].sum()) + 8*2*128
