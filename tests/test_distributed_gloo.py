"""The N > 1 path on CPU: two processes over `gloo` shard a batch of images, each codes its own shard with the host
coder (no data-path collective), and the single exchange step (one all-reduce of the rate accumulators, or the
all-gather of per-image values for exact parity) reproduces the single-process result bit for bit."""
import os
import socket

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from autoencoder_based_image_compression_amd import sharding
from autoencoder_based_image_compression_amd.kodak.lossless import compression

NB_IMAGES = 5   # ragged on purpose: 3 + 2


def _batch():
    rng = numpy.random.RandomState(42)
    symbols = numpy.round(rng.laplace(size=(NB_IMAGES, 128, 48))*rng.uniform(0.1, 3., size=(1, 128, 1))).astype(numpy.int16)
    probabilities = numpy.clip(rng.rand(128, 10), 0.05, 0.95)
    return (symbols, probabilities)


def _per_image_bits(symbols, probabilities):
    (rec, nb_bits) = compression.code_planar_symbols(symbols, probabilities, idx_map_exception=67, nb_threads=2)
    assert numpy.array_equal(rec, symbols)
    return nb_bits.sum(axis=1).astype(numpy.float64)


def _worker(rank, world_size, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group(backend='gloo', rank=rank, world_size=world_size)
    try:
        (symbols, probabilities) = _batch()
        (start, stop) = sharding.shard_bounds(NB_IMAGES, rank, world_size)
        bits = _per_image_bits(symbols[start:stop], probabilities)
        rate = bits/(128*192)
        local = numpy.array([bits.sum(), rate.sum(), float(stop - start)])
        total = sharding.reduce_statistics(local)
        gathered = sharding.gather_per_image(numpy.stack([bits, rate], axis=1), NB_IMAGES)
        numpy.savez(os.path.join(out_dir, 'rank{}.npz'.format(rank)), total=total, gathered=gathered, bounds=numpy.array([start, stop]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_ranks_reproduce_the_single_process_statistics(tmp_path):
    (symbols, probabilities) = _batch()
    bits = _per_image_bits(symbols, probabilities)
    rate = bits/(128*192)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    results = [numpy.load(str(tmp_path/'rank{}.npz'.format(r))) for r in range(2)]
    assert [tuple(r['bounds']) for r in results] == [(0, 3), (3, 5)]
    for r in results:
        assert r['total'][0] == bits.sum()                       # integer-valued: exact whatever the order
        assert r['total'][2] == NB_IMAGES
        assert abs(r['total'][1] - rate.sum()) < 1e-12
        assert numpy.array_equal(r['gathered'][:, 0], bits)      # global image order restored
        assert numpy.array_equal(r['gathered'][:, 1], rate)
        assert numpy.mean(r['gathered'][:, 1]) == numpy.mean(rate)   # bit-identical mean (exact-parity mode)
    assert numpy.array_equal(results[0]['total'], results[1]['total'])
