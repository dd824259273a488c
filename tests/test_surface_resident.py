"""Host logic behind the reference-shaped call surface's resident copies (kodak/_backend.py) and batched host tails: runs
without a GPU (a CPU tensor stands in for the device copy; nothing here launches a kernel)."""
import gc

import numpy
import pytest
import torch

from autoencoder_based_image_compression_amd.kodak import _backend as bk
from autoencoder_based_image_compression_amd.kodak.eae import batching
from autoencoder_based_image_compression_amd.kodak.tools import tools as tls


def _publish_cpu(tensor):
    (array, buffer) = bk._host_array(tuple(tensor.shape), bk._numpy_dtype(tensor.dtype))
    torch.from_numpy(array).copy_(tensor)
    return (array, bk.register(array, tensor, buffer))


def test_views_of_a_published_array_find_their_slice():
    tensor = torch.arange(24*2*3*4, dtype=torch.float32).reshape(24, 2, 3, 4)
    (array, record) = _publish_cpu(tensor)
    assert not array.flags.writeable and numpy.array_equal(array, tensor.numpy())
    assert bk.resident(array) == (record, 0)
    assert bk.resident(array[5, :, :, :]) == (record, 5*24)                       # `cq[j, :, :, :]` of the harness
    assert bk.resident(numpy.squeeze(array[..., None], axis=4)[3]) == (record, 3*24)
    assert bk.resident(array[2:4]) == (record, 48)
    assert bk.resident(array[:, :, :, 1]) is None                                 # not contiguous
    assert bk.resident(array.copy()) is None and bk.resident(array.astype(numpy.float64)) is None
    assert bk.resident(array.view(numpy.int32)[1]) is None                        # another dtype
    with pytest.raises(ValueError):
        array[0, 0, 0, 0] = 1.
    with pytest.raises(ValueError):
        array[3][0, 0, 0] = 1.                                                     # nor through a view


def test_an_array_made_writable_again_is_forgotten():
    (array, record) = _publish_cpu(torch.zeros(4, 6))
    view = array[1]
    before = bk.statistics['forgotten_writable']
    array.flags.writeable = True
    array[1, 2] = 5.
    assert bk.resident(view) is None and bk.resident(array) is None
    assert bk.statistics['forgotten_writable'] == before + 1
    array.flags.writeable = False
    assert bk.resident(array) is None                                             # for good


def test_the_device_copy_goes_with_the_array_and_its_bytes_are_reused():
    size = 12345
    (array, record) = _publish_cpu(torch.zeros(size))
    key = id(array)
    view = array[10:20]
    del array
    gc.collect()
    assert key in bk._REGISTRY and bk.resident(view) is not None                  # a view keeps the published array alive
    reused = bk.statistics['buffers_reused']
    del view
    gc.collect()
    assert key not in bk._REGISTRY
    (again, _) = _publish_cpu(torch.ones(size))
    assert bk.statistics['buffers_reused'] == reused + 1 and float(again.sum()) == size


def test_launches_are_whole_mini_batches():
    assert [(s.start, s.stop) for s in batching._launches(24, 4, 512*768)] == [(0, 24)]
    assert [(s.start, s.stop) for s in batching._launches(8, 2, 2048*2048)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert [(s.start, s.stop) for s in batching._launches(28, 4, 512*768)] == [(0, 24), (24, 28)]
    assert [(s.start, s.stop) for s in batching._launches(6, 3, 16*16)] == [(0, 6)]
    with pytest.raises(ValueError):
        list(batching._launches(7, 2, 256))


def test_entropies_of_all_rows_at_once_equal_the_reference_expression_row_by_row():
    """`_entropies_from_hist_rows` == tools.py:523-537 applied to each row, bit for bit, incl. the rows on which the reference
    raises (a uniform histogram of three or more bins exceeds its own `log2` bound by an ulp)."""
    rng = numpy.random.RandomState(1)
    raised = 0
    for trial in range(60):
        hist = numpy.zeros((128, 511), dtype=numpy.int64)
        for c in range(128):
            k = rng.randint(1, 1 + rng.choice([1, 2, 3, 7, 8, 9, 16, 40, 130, 300]))
            idx = rng.choice(511, size=k, replace=False)
            hist[c, idx] = rng.randint(1, rng.choice([50, 1000, 100000]), size=k)
            if k > 1:
                hist[c, idx[0]] += 7
            if trial % 10 == 9 and c == 77:
                hist[c] = 0
                hist[c, 100:111] = 5                    # 11 equal bins: the reference raises
        (expected, expected_error) = ([], None)
        for c in range(128):
            occupied = numpy.flatnonzero(hist[c])
            try:
                expected.append(tls._entropy_from_hist(hist[c, occupied[0]:occupied[-1] + 1]))
            except ValueError as exc:
                expected_error = str(exc)
                break
        if expected_error is not None:
            raised += 1
            with pytest.raises(ValueError) as info:
                tls._entropies_from_hist_rows(hist)
            assert str(info.value) == expected_error
        else:
            assert tls._entropies_from_hist_rows(hist).tobytes() == numpy.array(expected).tobytes()
    assert raised == 6


def test_row_sums_are_numpy_sum_of_every_row_bit_for_bit(monkeypatch):
    """The host library's pairwise row sums (include/eae_coder.h) against `numpy.sum` itself: every run length from 0 to 300 and a
    few long ones, terms of mixed magnitude; the self-check of the first call says `numpy order`; with the check forced to fail the
    function still returns `numpy.sum` of every row."""
    rng = numpy.random.RandomState(3)
    lengths = list(range(0, 301)) + [1000, 1024, 1031, 4097]
    bounds = numpy.concatenate(([0], numpy.cumsum(lengths))).astype(numpy.int64)
    for trial in range(4):
        values = rng.standard_normal(int(bounds[-1]))*numpy.exp2(rng.randint(-40, 10, size=int(bounds[-1])))
        if trial % 2:
            values = -numpy.abs(values)
        expected = numpy.array([numpy.sum(values[bounds[i]:bounds[i + 1]]) for i in range(len(lengths))])
        assert tls._row_sums(values, bounds).tobytes() == expected.tobytes()
    assert tls._ROW_SUMS_IN_NUMPY_ORDER == [True]
    monkeypatch.setattr(tls, '_ROW_SUMS_IN_NUMPY_ORDER', [False])
    assert tls._row_sums(values, bounds).tobytes() == expected.tobytes()


def test_row_sums_refuse_bad_arguments():
    import ctypes
    from autoencoder_based_image_compression_amd import _native
    lib = _native.coder()
    out = numpy.empty(2)
    values = numpy.ones(4)
    descending = numpy.array([0, 3, 1], dtype=numpy.int64)
    as_p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    assert lib.eae_coder_pairwise_row_sums(as_p(values, ctypes.c_double), as_p(descending, ctypes.c_int64), 2, as_p(out, ctypes.c_double)) == 5      # EAE_OUT_OF_RANGE
    assert lib.eae_coder_pairwise_row_sums(as_p(values, ctypes.c_double), None, 2, as_p(out, ctypes.c_double)) == -1      # EAE_NULL_POINTER


def test_exception_bits_of_a_batch_equal_the_per_map_function():
    """`exception_maps_nb_bits` (the result worker's one call per batch) == `exception_map_nb_bits` row by row, incl. histograms
    whose occupied bins have gaps, single-bin maps and the widths a recount over all of int16 gives."""
    from autoencoder_based_image_compression_amd.kodak.lossless import compression
    rng = numpy.random.RandomState(8)
    for (bins, map_size) in ((4095, 1536), (65535, 1536), (511, 256), (4095, 16384)):
        hist = numpy.zeros((24, bins), dtype=numpy.int64)
        for i in range(24):
            k = int(rng.choice([1, 2, 3, 9, 40, 200]))
            centre = bins//2 + int(rng.randint(-20, 21))
            idx = numpy.unique(numpy.clip(centre + rng.randint(-150, 151, size=k), 0, bins - 1))
            counts = rng.multinomial(map_size - idx.size, numpy.ones(idx.size)/idx.size) + 1
            hist[i, idx] = counts
        expected = numpy.array([int(compression.exception_map_nb_bits(row, map_size)) for row in hist], dtype=numpy.int64)
        got = compression.exception_maps_nb_bits(hist, map_size)
        assert got.dtype == numpy.int64 and numpy.array_equal(got, expected)
