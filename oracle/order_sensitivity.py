"""End-to-end consequence of the transforms' summation order -- TEST INFRASTRUCTURE (tests/test_gpu_order_sensitivity.py and the
checker leg of bench.py), never imported by the product.

The reference's latents and reconstructions come out of TensorFlow's Eigen / oneDNN kernels behind `sess.run`
(kodak_tensorflow/eae/batching.py:94-99, 49-53; the graph: eae/graph/components.py:86-142, 11-84), whose float32 summation order is
unspecified; TensorFlow is absent here. This build fixes an order of its own and proves GPU == that order bit
for bit. What north_star asks of the FLOAT side -- symbols bit-exact, reconstruction within 1e-4 dB PSNR -- can therefore only be
evidenced as a sensitivity: run the same graph in two other arithmetics,

    float64      the order-free value of the graph (torch-CPU, double precision; rounding order no longer matters at 1e-16)
    f32_onednn   float32 in ANOTHER order (torch-CPU's oneDNN convolutions: blocked, vectorised -- the closest stand-in for TF's kernels)

and state, per image and bin width, against the product's own results: the symbols that differ (count; how far each was from a
rounding boundary: the reference's quantiser is bw * round((y - mean) / bw), tools.py:883-929), the change in coded bits, and the
change in PSNR of the uint8 reconstruction (tools.py:61-93, 831-881). SURVEY.md section 7 "Hard parts (iii)" specified this contract."""
import numpy

from . import transforms_torch


def _psnr(reference_uint8, reconstruction_uint8):
    """tls.psnr_2d per image (tools.py:831-881): float64 MSE of the uint8 pair, 10 log10(255^2 / MSE)."""
    d = reference_uint8.astype(numpy.float64) - reconstruction_uint8.astype(numpy.float64)
    mse = (d*d).reshape(d.shape[0], -1).mean(axis=1)
    return 10.*numpy.log10(255.**2/mse)


class ReferencePath(object):
    """The graph in another arithmetic: encoder -> centred quantiser -> decoder -> cast_bt601, numpy in and out."""

    def __init__(self, name, variables, are_bin_widths_learned, threads=None):
        self.name = name
        self.dtype = numpy.float64 if name == 'float64' else numpy.float32
        self.t = transforms_torch.CpuTransforms(variables, are_bin_widths_learned, threads=threads, dtype=self.dtype)

    def latents(self, images_uint8):
        return self.t.encoder(images_uint8.astype(self.dtype)[..., None])

    def quantise(self, y, bin_widths, map_mean):
        """(symbols int64 [N,h,w,C], quantised latents, (y - mean) / bw in float64 for the boundary test)."""
        bw = bin_widths.astype(self.dtype).reshape(1, 1, 1, -1)
        mean = map_mean.astype(self.dtype).reshape(1, 1, 1, -1)
        centred = y - mean
        q = bw*numpy.round(centred/bw)                    # tools.py:929, in this path's own precision
        symbols = numpy.round(q/bw).astype(numpy.int64)   # cast_float_to_int16 of the rescaled maps (tools.py:95-133)
        return (symbols, q + mean, centred.astype(numpy.float64)/bw.astype(numpy.float64))

    def reconstruct(self, quantised):
        rec = self.t.decoder(quantised.astype(self.dtype))[..., 0]
        return numpy.round(rec.clip(min=16., max=235.)).astype(numpy.uint8)


def compare(images_uint8, product, variables, are_bin_widths_learned, bin_widths_by_name, map_mean, count_bits, threads=None,
            paths=('float64', 'f32_onednn')):
    """product: {name of the bin-width set: dict(symbols=int16 [N,h,w,C] (NHWC order), reconstruction=uint8 [N,H,W])} from the HIP
    path on `images_uint8`; bin_widths_by_name: {name: float32 [C]}; count_bits(symbols NHWC int) -> coded bits per image (the same
    probability tables for every arithmetic, so that only the flipped symbols move it). Returns a JSON-able dict."""
    out = {'images': int(images_uint8.shape[0]), 'height': int(images_uint8.shape[1]), 'width': int(images_uint8.shape[2]), 'paths': {}}
    for path_name in paths:
        path = ReferencePath(path_name, variables, are_bin_widths_learned, threads)
        y = path.latents(images_uint8)
        rows = {}
        for (name, bin_widths) in bin_widths_by_name.items():
            mine = product[name]
            (symbols, quantised, t) = path.quantise(y, bin_widths, map_mean)
            differ = symbols != mine['symbols'].astype(numpy.int64)
            # distance of the (float64-scaled) latent from the nearest rounding boundary k + 0.5, at the symbols that differ
            frac = numpy.abs(numpy.abs(t - numpy.floor(t)) - 0.5)
            margins = frac[differ]
            step = numpy.abs(symbols - mine['symbols'].astype(numpy.int64))[differ]
            rec = path.reconstruct(quantised)
            psnr_path = _psnr(images_uint8, rec)
            psnr_mine = _psnr(images_uint8, mine['reconstruction'])
            bits_path = numpy.asarray(count_bits(symbols), dtype=numpy.int64)
            bits_mine = numpy.asarray(count_bits(mine['symbols'].astype(numpy.int64)), dtype=numpy.int64)
            rows[name] = {
                'symbols': int(symbols.size), 'symbols_differing': int(differ.sum()),
                'symbols_differing_per_image_max': int(differ.reshape(differ.shape[0], -1).sum(axis=1).max()),
                'largest_distance_from_a_rounding_boundary': float(margins.max()) if margins.size else 0.,
                'largest_symbol_step': int(step.max()) if step.size else 0,
                'pixels_differing': int((rec != mine['reconstruction']).sum()),
                'largest_pixel_step': int(numpy.abs(rec.astype(numpy.int64) - mine['reconstruction'].astype(numpy.int64)).max()),
                'delta_bits_per_image_max': int(numpy.abs(bits_path - bits_mine).max()), 'bits_per_image_mean': float(bits_mine.mean()),
                'delta_psnr_db_per_image_max': float(numpy.abs(psnr_path - psnr_mine).max()),
                'psnr_db_mean': float(psnr_mine.mean()),
            }
        out['paths'][path_name] = rows
    return out
