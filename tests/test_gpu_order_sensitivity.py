"""north_star's floating-point clause, end to end (SURVEY.md section 7 "Hard parts (iii)"): "results match the
reference bit-exactly for the integer quantized symbols ... and within 1e-4 PSNR for the float reconstruction". The reference's
floats come out of TensorFlow's kernels (eae/batching.py:94-99, 49-53; graph eae/graph/components.py:86-142, 11-84), whose
summation order cannot be run here; so the HIP path is held against the same graph in float64 (order-free) and in float32 in
ANOTHER order (torch-CPU / oneDNN), on BASELINE.json configs[1] and configs[2]: every symbol that differs lay within 1e-4 of a
rounding boundary of the reference's quantiser (tools.py:883-929) and moved by one step, and every image's PSNR (tools.py:831-881)
moved by at most 1e-4 dB. `bench.py` prints the same figures as `order_sensitivity`."""
import numpy
import pytest
import torch

pytestmark = pytest.mark.gpu

PSNR_TOLERANCE_DB = 1e-4          # north_star
BOUNDARY_TOLERANCE = 1e-4         # | |frac((y - mean) / bw)| - 0.5 | of a symbol that differs


def check(report, label):
    rows = report[label]['vs']
    assert set(rows) == {'float64', 'f32_onednn'}
    for (path, per_width) in rows.items():
        for (name, r) in per_width.items():
            where = (label, path, name)
            assert r['symbols'] > 0 and r['psnr_db_mean'] > 0.
            # symbols: a handful in millions (float32 rounding, ~1e-6 of a latent of size 1-4, against the bin width: the share of
            # latents that close to a boundary is ~2e-6 / bin width), each a one-step move across a boundary the latent sat on
            assert r['symbols_differing'] <= max(2, r['symbols']//20000), (where, r)
            assert r['largest_symbol_step'] <= 1 and r['largest_distance_from_a_rounding_boundary'] < BOUNDARY_TOLERANCE, (where, r)
            # reconstruction: single grey levels (a few more around a symbol that moved: the synthesis transform spreads one bin over
            # a 16 x 16 neighbourhood), PSNR within north_star's tolerance
            assert r['largest_pixel_step'] <= (1 if r['symbols_differing'] == 0 else 16), (where, r)
            assert r['delta_psnr_db_per_image_max'] <= PSNR_TOLERANCE_DB, (where, r)
            # bits: a differing symbol moves the coded size of an image by a few bits at most
            assert r['delta_bits_per_image_max'] <= 16*max(1, r['symbols_differing_per_image_max']), (where, r)


def test_single_kodak_image_bin_width_1():
    """configs[1]: one 512 x 768 luminance image, bin width 1.0."""
    import bench
    report = bench.order_sensitivity(torch.device('cuda', 0), bench.usable_cpus(), configs=(('configs[1]', 1, (1.0,)),))
    check(report, 'configs[1]')
    assert report['summary']['within_tolerance'] is True


def test_kodak_set_at_three_bin_widths():
    """configs[2]: the 24 Kodak-sized images at bin widths 0.5 / 1.0 / 2.0 (the rate-PSNR curve's points)."""
    import bench
    report = bench.order_sensitivity(torch.device('cuda', 0), bench.usable_cpus(), configs=(('configs[2]', 24, (0.5, 1.0, 2.0)),))
    check(report, 'configs[2]')
    s = report['summary']
    assert s['within_tolerance'] is True and s['symbols_compared'] == 2*3*24*32*48*128
    print('order sensitivity, configs[2]:', s)


def test_a_finer_quantiser_still_holds():
    """Beyond the configs: bin width 0.125 (1.4 bpp with these weights) puts eight times as many boundaries under the latents."""
    import bench
    report = bench.order_sensitivity(torch.device('cuda', 0), bench.usable_cpus(), configs=(('fine', 4, (0.125,)),))
    check(report, 'fine')
