"""Writes tests/golden/transforms_golden.npz: outputs of the CPU restatement of the transforms (oracle/transforms_oracle.c)
for seeded weights, SURVEY.md 8(c) item (5). TEST INFRASTRUCTURE ONLY.

These are NOT outputs of the reference's TensorFlow graph (TensorFlow is absent: conv values are "parity unpinned" by the
reference); they freeze the restatement itself -- its summation order, compiler flags and the float32
arithmetic of the build host -- so that the oracle cannot drift unnoticed, and give the GPU tests a fixture that does not
depend on running the oracle.

Contents, for the fixed-bin-width and the learned-bin-width model (variables.random_variables, seed 0) on a 64x96 and a
256x256 synthetic image: per-layer float64 sums and CRC32 of the float32 bytes (gdn_1, gdn_2, conv_3, latents, igdn_2,
igdn_3, reconstruction), the full latents, the quantised latents at bin width 1 and the uint8 reconstruction.
Needs nothing from /root/reference."""
import os
import sys
import zlib

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var   # noqa: E402  (host numpy only)
from oracle import transforms as T                                                     # noqa: E402


def image(seed, h, w):
    rng = numpy.random.RandomState(seed)
    x = rng.randint(16, 236, size=(1, h, w)).astype(numpy.float32)
    for _ in range(2):
        x = (x + numpy.roll(x, 1, 1) + numpy.roll(x, -1, 2))/numpy.float32(3.)
    return numpy.round(x).astype(numpy.uint8)


def summary(a):
    a = numpy.ascontiguousarray(a, dtype=numpy.float32)
    return numpy.array([numpy.sum(a, dtype=numpy.float64), float(zlib.crc32(a.tobytes()))])


def main():
    g = {}
    for learned in (False, True):
        v = var.random_variables(1., learned, seed=0, bias_std=0.01)
        v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)
        for (h, w) in ((64, 96), (256, 256)):
            tag = '{0}_{1}x{2}'.format('learned' if learned else 'fixed', h, w)
            x = image(h + w, h, w)
            (y, enc) = T.encoder(x.astype(numpy.float32)[..., None], v, learned, return_intermediates=True)
            q = numpy.round(y)                                  # bin width 1, no centring
            (rec, dec) = T.decoder(q, v, learned, return_intermediates=True)
            rec_u8 = numpy.round(rec[..., 0].clip(min=16., max=235.)).astype(numpy.uint8)
            g[tag + '_x'] = x
            g[tag + '_y'] = y
            g[tag + '_q'] = q.astype(numpy.float32)
            g[tag + '_rec_u8'] = rec_u8
            for (name, a) in (('gdn_1', enc['gdn_1']), ('gdn_2', enc['gdn_2']), ('conv_3', enc['conv_3']), ('y', y),
                              ('igdn_2', dec['igdn_2']), ('igdn_3', dec['igdn_3']), ('rec', rec)):
                g[tag + '_sum_crc_' + name] = summary(a)
    out = os.path.join(ROOT, 'tests', 'golden', 'transforms_golden.npz')
    numpy.savez_compressed(out, **g)
    print('transforms_golden.npz', os.path.getsize(out), 'bytes;', {k: g[k].tolist() for k in g if k.endswith('sum_crc_y')})


if __name__ == '__main__':
    main()
