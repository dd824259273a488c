"""BASELINE.json's configurations that round 1 only ran from a script pytest did not collect:

  configs[1]  one 512x768 luminance image, bin width 1.0, "bitstream diffed vs the reference lossless/ coder": the device
              coder's arithmetic-coded and bypass BYTES of all 127 coded maps of the path's OWN symbols against the CPU
              coder (the reference's C++ build `oracle/_ref` for the bits, the plain-C oracle -- pinned to the reference's
              streams by tests/golden/coder_golden.npz -- for the bytes);
  configs[2]  the 24-image Kodak-shaped set at bin-width multipliers {0.5, 1.0, 2.0}: per-image bits and squared errors of
              the fused MI355X path equal the CPU evaluation (reference loop: reconstructing_eae_kodak.py:181-225);
  configs[3]  64 images of 256x256: ONE rank's share of the 512-image batch sharded over 8 GPUs, through the same path.
              (The sharding itself -- 8 x 64 with one all-reduce -- is tests/test_distributed_gloo.py and
              tests/test_bench_launcher.py; a rank's arithmetic is this test.)

The CPU side is checker code (oracle/): transforms by oracle/transforms_oracle.c, coder by the reference's own C++ where
its build is present."""
import numpy
import pytest
import torch

pytestmark = pytest.mark.gpu

(L, IDX) = (10, 67)


def cpu_evaluation(images, variables, bwt, map_mean, probabilities, y_cpu):
    """Per-image (bits, squared error, symbols) the way `fix_gamma` forms them, on the host (reconstructing_eae_kodak.py:170-225)."""
    from autoencoder_based_image_compression_amd.kodak.lossless import compression
    from oracle import coder as oracle_coder, transforms as T
    lib = oracle_coder.CoderLib('ref' if oracle_coder.available('ref') else 'oracle')
    (n, h, w) = images.shape
    tiled = numpy.tile(bwt.reshape(1, 1, 1, 128), y_cpu.shape[:3] + (1,))
    cq = tiled*numpy.round((y_cpu - map_mean)/tiled)                       # tools.py:927-929 after the centring of :178
    symbols = numpy.round(cq/tiled).astype(numpy.int16)                    # compression.py:142
    reconstruction = T.decoder(cq + map_mean, variables, False)[..., 0]
    rec_u8 = numpy.round(reconstruction.clip(min=16., max=235.)).astype(numpy.uint8)     # tools.py:93
    bits = numpy.zeros(n, dtype=numpy.int64)
    for j in range(n):
        for ch in range(128):
            flat = numpy.ascontiguousarray(symbols[j, :, :, ch]).reshape(-1)
            if ch == IDX:
                counts = numpy.bincount(flat.astype(numpy.int64) + 32768)
                bits[j] += int(compression.exception_map_nb_bits(counts, flat.size))     # compression.py:68-75
            else:
                (rec, nb) = lib.compress_lossless(flat, probabilities[ch])
                assert numpy.array_equal(rec, flat)
                bits[j] += nb
    sse = ((images.astype(numpy.int64) - rec_u8.astype(numpy.int64))**2).reshape(n, -1).sum(axis=1)
    dead = (numpy.abs(cq).reshape(n, -1, 128).sum(axis=1) == 0).sum(axis=1)
    return (bits, sse, dead, symbols)


def fused_against_cpu(n, h, w, multipliers, seed, statistics=None):
    import bench
    from autoencoder_based_image_compression_amd import codec, device as dev, pipeline
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats
    from oracle import transforms as T
    variables = bench.synthetic_model(1.)
    images = bench.synthetic_images(seed, n, h, w)
    bin_widths = variables[var.BIN_WIDTHS_NAME]
    images_device = torch.from_numpy(images).cuda()
    y_device = pipeline.DeviceEncoder(variables, False)(images_device)
    map_mean = dev.map_means(y_device).cpu().numpy()
    y_gpu = y_device.cpu().numpy()
    y_cpu = T.encoder(images.astype(numpy.float32)[..., None], variables, False)
    assert numpy.array_equal(y_cpu, y_gpu)
    rates = []
    for multiplier in multipliers:
        bwt = (numpy.float32(multiplier)*bin_widths).astype(numpy.float32)
        probabilities = lossless_stats.compute_binary_probabilities(y_gpu, bwt, map_mean, L)
        if statistics is not None:                # tables that were not made from these latents (any valid table codes any symbols)
            (map_mean, probabilities) = statistics
        with codec.BatchCodec(variables, False, bwt, map_mean, probabilities, IDX, n, h, w) as fused:
            got = fused.submit(images_device).result()
        (bits, sse, dead, _) = cpu_evaluation(images, variables, bwt, map_mean, probabilities, y_cpu)
        assert numpy.array_equal(got['nb_bits'], bits), multiplier
        assert numpy.array_equal(got['sse'], sse), multiplier
        assert numpy.array_equal(got['nb_deads'], dead), multiplier
        rates.append(float(bits.mean())/(h*w))
    return rates


def test_the_authors_statistics_on_synthetic_latents():
    """The coder on the AUTHORS' statistics (lossless/results/1_10000/training_index_10/: `map_mean.npy`, `idx_map_exception.pkl` = 67,
    `binary_probabilities_1.npy`, held as data in tests/golden/coder_golden.npz) over the path's own symbols: bits per image of the
    fused MI355X path == the CPU coder's (the reference's C++ where its build is present). What `bench.py: authors_statistics` times."""
    import os
    with numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coder_golden.npz')) as g:
        assert int(g['real_idx_map_exception']) == IDX
        statistics = (g['real_map_mean'].astype(numpy.float32), g['real_probabilities_1'].copy())
    rates = fused_against_cpu(3, 128, 192, (1.0,), 31, statistics=statistics)
    assert rates[0] > 0.


def test_config3_kodak_set_at_three_bin_widths():
    """configs[2]: 24 x 512x768, multipliers {0.5, 1.0, 2.0}: bits, squared error and dead maps of every image, exact."""
    rates = fused_against_cpu(24, 512, 768, (0.5, 1.0, 2.0), seed=1000)
    assert rates[0] > rates[1] > rates[2] > 0.          # a finer quantiser costs more bits


def test_config4_one_rank_share_of_the_256_batch():
    """configs[3]: 64 x 256x256 = what one of 8 ranks codes per step of the 512-image batch."""
    fused_against_cpu(64, 256, 256, (1.0,), seed=1003)          # seed = 1000 + rank, rank 3 of 8 (bench.py)


@pytest.mark.parametrize('multiplier', [1.0, 0.25])
def test_config2_bitstream_bytes_of_all_maps(multiplier):
    """configs[1]: ONE 512x768 image; the two streams of every coded map, byte for byte (0.25: longer streams with escapes)."""
    import bench
    from autoencoder_based_image_compression_amd import device as dev, pipeline
    from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
    from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats
    from oracle import coder as oracle_coder
    variables = bench.synthetic_model(1.)
    images = bench.synthetic_images(7, 1, 512, 768)
    bwt = (numpy.float32(multiplier)*variables[var.BIN_WIDTHS_NAME]).astype(numpy.float32)
    y = pipeline.DeviceEncoder(variables, False)(torch.from_numpy(images).cuda())
    map_mean = dev.map_means(y)
    probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bwt, map_mean.cpu().numpy(), L)
    q = dev.quantize_maps(y, torch.from_numpy(bwt).cuda(), map_mean, want_symbols=True)
    symbols = q['symbols'].reshape(128, -1)
    prob_row = torch.arange(128, dtype=torch.int32)
    prob_row[IDX] = -1
    streams = dev.coder_encode_batch(symbols, torch.from_numpy(probabilities).cuda(), prob_row.cuda(), L)
    decoded = dev.coder_decode_batch(streams, torch.from_numpy(probabilities).cuda(), prob_row.cuda()).cpu().numpy()
    results = streams.results.cpu().numpy()
    data = streams.streams.cpu().numpy()
    half = streams.stride//2
    host_symbols = symbols.cpu().numpy()
    oracle_lib = oracle_coder.CoderLib('oracle')
    ref_lib = oracle_coder.CoderLib('ref') if oracle_coder.available('ref') else None
    assert not results[2].any()
    total = 0
    for m in range(128):
        if m == IDX:
            assert results[0, m] == 0 and results[1, m] == 0
            continue
        (rec, nb_bits, ref) = oracle_lib.compress_lossless(host_symbols[m], probabilities[m], want_streams=True)
        assert (int(results[0, m]), int(results[1, m])) == (ref['bac_bits'], ref['bypass_bits']), m
        assert numpy.array_equal(data[m, :ref['bac_bytes'].size], ref['bac_bytes']), m
        assert numpy.array_equal(data[m, half:half + ref['bypass_bytes'].size], ref['bypass_bytes']), m
        assert numpy.array_equal(decoded[m], host_symbols[m]) and numpy.array_equal(rec, host_symbols[m]), m
        if ref_lib is not None:          # the reference's own C++ build: same size for the same symbols
            assert ref_lib.compress_lossless(host_symbols[m], probabilities[m])[1] == nb_bits, m
        total += nb_bits
    assert total > 0
