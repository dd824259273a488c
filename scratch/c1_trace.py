"""Phase breakdown of conv1_kernel from a -DEAE_TRACE build (SRC=conv1 EXTRA=-DEAE_TRACE SCRIPT=c1_trace.py scratch/variant.sh)."""
import os, sys, ctypes
import numpy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline, _native
batch, H, W = 24, 512, 768
variables = bench.synthetic_model(1.)
enc = pipeline.DeviceEncoder(variables, False)
v = enc.v
images = torch.from_numpy(bench.synthetic_images(5, batch, H, W)).cuda()
gdn = len(sys.argv) < 2 or sys.argv[1] != 'nogdn'
def run():
    if gdn:
        dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1'])
    else:
        dev.conv9x9s4_u8(images, enc.w1, v['encoder/biases_1'], None, None)
for _ in range(3):
    run()
torch.cuda.synchronize()
(a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
a.record(); run(); b.record(); torch.cuda.synchronize()
waves = batch*(H//4//8)*(W//4//16)*4
buf = numpy.zeros(waves*8, dtype=numpy.int64)
lib = _native.hip()
lib.eae_hip_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.eae_hip_trace_read(buf.ctypes.data, buf.size) == 0
t = buf.reshape(waves, 8)
marks = [0, 1, 2, 3, 4] if gdn else [0, 1, 2, 4]
names = {1: 'patch staging + barrier', 2: 'conv MFMA loop', 3: 'bias + gdn MFMA loop', 4: ('sqrt, /, stores' if gdn else 'bias + stores')}
print('launch %.1f us, %d waves; cycles per wave (mean / median):' % (a.elapsed_time(b)*1e3, waves))
for (p, q) in zip(marks[:-1], marks[1:]):
    d = t[:, q] - t[:, p]
    print('  %-26s %9.0f %9.0f' % (names[q], d.mean(), numpy.median(d)))
life = t[:, 4] - t[:, 0]
span = t[:, 4].max() - t[:, 0].min()
print('  wave lifetime %.0f; launch span %.0f ticks -> %.0f MHz; waves x lifetime / span = %.1f resident waves' % (life.mean(), span, span/(a.elapsed_time(b)*1e3), life.sum()/span))
