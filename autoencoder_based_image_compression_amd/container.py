"""A self-describing container for the coded latent variables and a standalone decoder.

The reference never serialises: `compress_lossless` (kodak_tensorflow/lossless/c++/source/compression.cpp:27-64) builds
the two bitstreams of a map, counts their bits, decodes them again and throws them away, and the decoder side of
`fix_gamma` (reconstructing_eae_kodak.py:192-207) restarts from the float array the encoder side still holds. Here the
streams produced by the device coder are packed into one blob that carries everything but the network weights, and
`decode_images` rebuilds the images from the blob alone: unpack -> arithmetic decode -> symbol * bin width + map mean
(the inverse of lossless/compression.py:142 and reconstructing_eae_kodak.py:178) -> synthesis transform -> BT.601 cast.
The reconstruction is bit-identical to the in-memory path's (tests/test_container.py).

Layout (little endian):
    magic 'EAE1' | version u16 | flags u16 (bit 0: learned bin widths) | nb_images u32 | height u32 | width u32 |
    nb_maps u16 | L u8 | reserved u8 | idx_map_exception i32
    bin_widths f32[nb_maps] | map_mean f32[nb_maps] | binary_probabilities f64[nb_maps][L]
    exception_probabilities f64[nb_images][L]              (only when idx_map_exception >= 0)
    bit counts u32[nb_images * nb_maps][2]                 (arithmetic-coded stream, bypass stream)
    payload: for every image, every map: arithmetic-coded bytes, then bypass bytes (each rounded up to a byte)

The exception map. The reference does not code it: it charges ceil(h*w*entropy) bits for it (compression.py:68-75), the
cost of an ideal adaptive coder. A decodable file has to carry it, so it goes through the same UEG0 + arithmetic coder
with a probability row measured on that very map (stats.py:181-195 applied to its histogram), stored per image.
"""
import struct

import numpy
import torch

from . import device as dev
from .kodak.eae.graph import constants as csts
from .kodak.lossless import interface_cython
from .kodak.lossless import stats as lossless_stats
from .kodak.tools import tools as tls

MAGIC = b'EAE1'
VERSION = 1
_HEADER = struct.Struct('<4sHHIIIHBBi')


def _exception_rows(symbols_planar, idx_map_exception, truncated_unary_length):
    """One probability row per image for its exception map, from the map's own histogram (stats.py:181-195, :56-66)."""
    nb_images = symbols_planar.shape[0]
    (hist, radius) = tls._symbol_histograms(symbols_planar[:, idx_map_exception:idx_map_exception + 1].contiguous())
    rows = numpy.zeros((nb_images, truncated_unary_length), dtype=numpy.float64)
    for i in range(nb_images):
        hist_abs = hist[i, radius:].copy()
        hist_abs[1:] += hist[i, :radius][::-1]
        (zeros, ones) = lossless_stats._decisions_from_hist(hist_abs, truncated_unary_length)
        total = (zeros + ones).astype(numpy.float64)
        with numpy.errstate(invalid='ignore', divide='ignore'):
            p = zeros.astype(numpy.float64)/total
        p[numpy.isnan(p)] = 0.5
        p[p == 0.] = 0.01
        p[p == 1.] = 0.99
        rows[i] = p
    return rows


def stream_capacity_bits(map_size, truncated_unary_length):
    """Bits a stream of the coder can hold at most: map_size * max(32, L) rounded up to a byte, the reference's own capacity
    (compression.cpp:24, Bitstream.cpp:3-11); the coder writes into a region of at least that many bytes + 16 per stream
    (eae_hip_coder_stream_stride_bytes / 2). Pure arithmetic -- the header check does not need the GPU library."""
    return (map_size*max(32, truncated_unary_length) + 7)//8*8


def _raise_for_statuses(results):
    bad = numpy.flatnonzero(results[2])
    if bad.size:
        interface_cython.raise_for_status(int(results[2, bad[0]]), int(results[3, bad[0]]))


def encode_images(luminances_uint8, encoder, bin_widths_test, map_mean, binary_probabilities, idx_map_exception=-1):
    """uint8 (N, H, W) or (N, H, W, 1) luminance images -> (blob bytes, info dict).

    encoder: pipeline.DeviceEncoder of the model; bin_widths_test float32 (128,) (the trained bin widths times the
    multiplier of the rate point); map_mean float32 (128,); binary_probabilities float64 (128, L) (stats.py:13-68).
    info: 'nb_bits' uint32 (N, 128) (arithmetic-coded + bypass bits of every map), 'payload_bytes', 'header_bytes'.
    """
    images = numpy.ascontiguousarray(luminances_uint8)
    if images.dtype != numpy.uint8:
        raise TypeError('`luminances_uint8.dtype` is not equal to `numpy.uint8`.')
    if images.ndim == 4:
        images = images[:, :, :, 0]
    (nb_images, height, width) = images.shape
    probabilities = numpy.ascontiguousarray(binary_probabilities, dtype=numpy.float64)
    (nb_maps, truncated_unary_length) = probabilities.shape
    if truncated_unary_length < 1 or truncated_unary_length > 255:
        raise ValueError('The truncated unary length does not belong to [1, 255].')
    bin_widths = numpy.ascontiguousarray(bin_widths_test, dtype=numpy.float32)
    mean = numpy.ascontiguousarray(map_mean, dtype=numpy.float32)
    if bin_widths.shape != (nb_maps,) or mean.shape != (nb_maps,):
        raise ValueError('`bin_widths_test` and `map_mean` must have one element per map.')
    device = encoder.device
    y = encoder(torch.from_numpy(images).to(device))
    map_size = y.shape[1]*y.shape[2]
    q = dev.quantize_maps(y, torch.from_numpy(bin_widths).to(device), torch.from_numpy(mean).to(device), want_symbols=True)
    if int(q['checks'][0].item()) != 0:
        raise AssertionError('The rounded array elements cannot be represented as 16-bit signed integers.')
    symbols = q['symbols']                                              # [N, 128, map_size] int16, stays in HBM
    prob_row = numpy.tile(numpy.arange(nb_maps, dtype=numpy.int32), nb_images)
    table = probabilities
    exception_rows = numpy.zeros((0, truncated_unary_length), dtype=numpy.float64)
    if 0 <= idx_map_exception < nb_maps:
        exception_rows = _exception_rows(symbols, idx_map_exception, truncated_unary_length)
        table = numpy.concatenate([probabilities, exception_rows])
        prob_row[idx_map_exception::nb_maps] = nb_maps + numpy.arange(nb_images, dtype=numpy.int32)
    else:
        idx_map_exception = -1
    n_maps = nb_images*nb_maps
    streams = dev.coder_encode_batch(symbols.view(n_maps, map_size), torch.from_numpy(table).to(device),
                                     torch.from_numpy(prob_row).to(device), truncated_unary_length)
    results = streams.results.cpu().numpy()
    _raise_for_statuses(results)
    bits = numpy.stack([results[0], results[1]], axis=1).astype(numpy.uint32)          # [n_maps, 2]
    nbytes = (bits.astype(numpy.int64) + 7)//8
    offsets = numpy.concatenate([[0], numpy.cumsum(nbytes.reshape(-1))[:-1]]).astype(numpy.int64).reshape(n_maps, 2)
    payload_bytes = int(nbytes.sum())
    payload = dev.coder_pack_streams(streams, torch.from_numpy(offsets).to(device), payload_bytes)
    head = _HEADER.pack(MAGIC, VERSION, 1 if encoder.are_bin_widths_learned else 0, nb_images, height, width, nb_maps,
                        truncated_unary_length, 0, idx_map_exception)
    pieces = [head, bin_widths.tobytes(), mean.tobytes(), probabilities.tobytes(), exception_rows.tobytes(), bits.tobytes()]
    header_bytes = sum(len(piece) for piece in pieces)
    blob = b''.join(pieces) + payload[:payload_bytes].cpu().numpy().tobytes()
    info = {'nb_bits': (bits[:, 0] + bits[:, 1]).reshape(nb_images, nb_maps), 'payload_bytes': payload_bytes,
            'header_bytes': header_bytes}
    return (blob, info)


def read_header(blob):
    """Parses everything in front of the payload. Raises ValueError on a malformed or truncated blob."""
    if len(blob) < _HEADER.size:
        raise ValueError('The container is truncated.')
    (magic, version, flags, nb_images, height, width, nb_maps, truncated_unary_length, _, idx_map_exception) = _HEADER.unpack_from(blob, 0)
    if magic != MAGIC:
        raise ValueError('The container does not start with the magic bytes.')
    if version != VERSION:
        raise ValueError('The container version {} is not supported.'.format(version))
    # sizes first: they dimension device buffers (a crafted header must not get that far)
    if nb_maps != csts.NB_MAPS_3:
        raise ValueError('The container does not hold {} maps per image.'.format(csts.NB_MAPS_3))
    if truncated_unary_length < 1:
        raise ValueError('The truncated unary length does not belong to [1, 255].')
    if nb_images < 1 or height < 1 or width < 1 or height % csts.STRIDE_PROD != 0 or width % csts.STRIDE_PROD != 0:
        raise ValueError('The image sizes in the container are not positive multiples of {}.'.format(csts.STRIDE_PROD))
    if not -1 <= idx_map_exception < nb_maps:
        raise ValueError('The index of the exception map in the container is out of range.')
    pos = _HEADER.size

    def take(count, dtype):
        nonlocal pos
        nbytes = count*numpy.dtype(dtype).itemsize
        if pos + nbytes > len(blob):
            raise ValueError('The container is truncated.')
        out = numpy.frombuffer(blob, dtype=dtype, count=count, offset=pos).copy()
        pos += nbytes
        return out

    header = {'nb_images': nb_images, 'height': height, 'width': width, 'nb_maps': nb_maps,
              'truncated_unary_length': truncated_unary_length, 'idx_map_exception': idx_map_exception,
              'are_bin_widths_learned': bool(flags & 1)}
    header['bin_widths'] = take(nb_maps, numpy.float32)
    header['map_mean'] = take(nb_maps, numpy.float32)
    header['binary_probabilities'] = take(nb_maps*truncated_unary_length, numpy.float64).reshape(nb_maps, truncated_unary_length)
    nb_rows = nb_images if idx_map_exception >= 0 else 0
    header['exception_probabilities'] = take(nb_rows*truncated_unary_length, numpy.float64).reshape(nb_rows, truncated_unary_length)
    header['bits'] = take(nb_images*nb_maps*2, numpy.uint32).reshape(nb_images*nb_maps, 2)
    header['payload_offset'] = pos
    # a stream can never be longer than the region the coder gives a map (compression.cpp:24): an inflated count would make
    # the unpacking write past it
    map_size = (height//csts.STRIDE_PROD)*(width//csts.STRIDE_PROD)
    if int(header['bits'].max(initial=0)) > stream_capacity_bits(map_size, truncated_unary_length):
        raise ValueError('A bit count of the header exceeds the capacity of a stream.')
    payload_bytes = int(((header['bits'].astype(numpy.int64) + 7)//8).sum())
    if pos + payload_bytes != len(blob):
        raise ValueError('The payload size does not match the bit counts of the header.')
    return header


def decode_symbols(blob, device='cuda'):
    """blob -> (header, int16 symbols [N, 128, map_size] on the device), arithmetic decoding only."""
    header = read_header(blob)
    (nb_images, nb_maps) = (header['nb_images'], header['nb_maps'])
    if header['height'] % 16 != 0 or header['width'] % 16 != 0:
        raise ValueError('The image size in the container is not divisible by 16.')
    map_size = (header['height']//16)*(header['width']//16)
    n_maps = nb_images*nb_maps
    device = torch.device(device)
    bits = header['bits']
    nbytes = (bits.astype(numpy.int64) + 7)//8
    offsets = numpy.concatenate([[0], numpy.cumsum(nbytes.reshape(-1))[:-1]]).astype(numpy.int64).reshape(n_maps, 2)
    payload = numpy.frombuffer(blob, dtype=numpy.uint8, offset=header['payload_offset'])
    payload_device = torch.from_numpy(numpy.concatenate([payload, numpy.zeros(8, dtype=numpy.uint8)])).to(device)
    streams = dev.coder_unpack_streams(payload_device, torch.from_numpy(offsets).to(device),
                                       torch.from_numpy(bits[:, 0].astype(numpy.int32)).to(device),
                                       torch.from_numpy(bits[:, 1].astype(numpy.int32)).to(device), map_size,
                                       header['truncated_unary_length'])
    table = numpy.concatenate([header['binary_probabilities'], header['exception_probabilities']])
    prob_row = numpy.tile(numpy.arange(nb_maps, dtype=numpy.int32), nb_images)
    if header['idx_map_exception'] >= 0:
        prob_row[header['idx_map_exception']::nb_maps] = nb_maps + numpy.arange(nb_images, dtype=numpy.int32)
    symbols = dev.coder_decode_batch(streams, torch.from_numpy(table).to(device), torch.from_numpy(prob_row).to(device))
    _raise_for_statuses(streams.results.cpu().numpy())
    return (header, symbols.view(nb_images, nb_maps, map_size))


def decode_images(blob, decoder):
    """blob + pipeline.DeviceDecoder of the model -> uint8 (N, H, W) reconstructions (BT.601 range, tools.py:61-93)."""
    (header, symbols) = decode_symbols(blob, decoder.device)
    if header['are_bin_widths_learned'] != decoder.are_bin_widths_learned:
        raise ValueError('The container was written by the other kind of model (learned / fixed bin widths).')
    device = decoder.device
    shifted = dev.dequantize_maps(symbols, torch.from_numpy(header['bin_widths']).to(device),
                                  torch.from_numpy(header['map_mean']).to(device))['shifted']
    (h_map, w_map) = (header['height']//16, header['width']//16)
    (_, reconstruction_uint8, _) = decoder(shifted.view(header['nb_images'], h_map, w_map, header['nb_maps']))
    return reconstruction_uint8.cpu().numpy()
