"""One case of tests/test_gpu_kernels.py::test_tconv9x9s4_luma, with the places that differ. usage: t3_case.py N H W [strips]"""
import os, sys
import numpy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 4:
    os.environ['EAE_HIP_T3_STRIPS'] = sys.argv[4]
from autoencoder_based_image_compression_amd import device as dev
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables
from oracle import transforms as orc
shape = tuple(int(a) for a in sys.argv[1:4])
v = variables.random_variables(1., False, seed=7, bias_std=0.01)
rng = numpy.random.RandomState(8)
w6 = (numpy.absolute(v['decoder/weights_6'])*numpy.float32(8.)).astype(numpy.float32)
x = (rng.standard_normal(size=shape + (128,)) + 1.5).astype(numpy.float32)
ref = orc.conv2d_transpose_same(x, w6, 4, None, col2im=True)[..., 0]
ref_u8 = numpy.round(ref.clip(min=16., max=235.)).astype(numpy.uint8)
target = numpy.random.RandomState(9).randint(16, 236, size=ref_u8.shape).astype(numpy.uint8)
wph = dev.pack_tconv9x9s4_weights(torch.from_numpy(w6).cuda())
for rep in range(5):
    f32, u8, sse = dev.tconv9x9s4_luma(torch.from_numpy(x).cuda(), wph, want_f32=True, want_u8=True, ref_u8=torch.from_numpy(target).cuda())
    f32 = f32.cpu().numpy()
    bad = numpy.argwhere(f32 != ref)
    expected = ((target.astype(numpy.int64) - ref_u8.astype(numpy.int64))**2).reshape(shape[0], -1).sum(axis=1)
    print('rep', rep, 'f32 mismatches', len(bad), 'u8', int((u8.cpu().numpy() != ref_u8).sum()), 'sse ok', numpy.array_equal(sse.cpu().numpy(), expected))
    if len(bad):
        rows = sorted(set((int(b[0]), int(b[1])) for b in bad))
        print('  rows (img, pixel row):', rows[:20], '... cols', sorted(set(int(b[2]) for b in bad))[:24])
        for b in bad[:6]:
            print('   ', tuple(int(t) for t in b), f32[tuple(b)], ref[tuple(b)])
