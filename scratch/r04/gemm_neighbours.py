"""r04: what kind of neighbour costs the conv GEMM its 8 points in the pipeline (0.87 of peak alone, 0.79 next to the coder)? conv_2 +
GDN_2 of a Kodak batch timed alone and next to synthetic one-wave-block kernels on another stream: dependent integer / FP64 chains
(the coder's serial cores: 48 or 96 waves), streaming passes (its data-parallel passes: 3,048 waves over tens of MB), both."""
import ctypes, os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy, torch
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline
lk = ctypes.CDLL(os.path.join(ROOT, 'scratch', 'r04', 'libs', 'liblk.so'))
lk.lk_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
variables = bench.synthetic_model(1.0)
images = torch.from_numpy(bench.synthetic_images(1000, 24, 512, 768)).cuda()
enc = pipeline.DeviceEncoder(variables, False)
gdn_1 = dev.conv9x9s4_u8(images, enc.w1, enc.v['encoder/biases_1'], enc.g[1], enc.v['encoder/beta_1'])
out = torch.empty((24, 64, 96, 128), device='cuda')
ws = dev.conv_workspace('cuda')
big = torch.zeros(1 << 27, dtype=torch.float32, device='cuda')       # 512 MB
sides = [torch.cuda.Stream() for _ in range(3)]


def conv2():
    dev.conv5x5s2(gdn_1, enc.w2, enc.v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], enc.v['encoder/beta_2'], out=out, workspace=ws)


def neighbours(spec):
    for (i, (kind, blocks, iters, times)) in enumerate(spec):
        for _ in range(times):
            lk.lk_launch(kind, blocks, iters, ctypes.c_void_p(big.data_ptr()), big.numel()*4, ctypes.c_void_p(sides[i % 3].cuda_stream))


def timed(spec):
    times = []
    for _ in range(10):
        torch.cuda.synchronize()
        neighbours(spec)
        conv2()                                   # the neighbours settle
        (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        a.record()
        for _ in range(3):
            conv2()
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b)/3.)
    return float(numpy.median(times))


for _ in range(5):
    conv2()
base = timed([])
print('conv_2 + GDN_2, 24 Kodak images, alone: %.4f ms' % base)
CASES = [
    ('48 chain waves (one serial core kernel)', [(6, 48, 60000, 1)]),
    ('96 chain waves', [(6, 96, 60000, 1)]),
    ('256 chain waves', [(6, 256, 60000, 1)]),
    ('1024 chain waves', [(6, 1024, 60000, 1)]),
    ('48 chain waves in 8 consecutive launches of 1/8 the length (the same work, hopping between SIMDs)', [(6, 48, 7500, 8)]),
    ('48 chain waves in 32 consecutive launches of 1/32 the length', [(6, 48, 1875, 32)]),
    ('96 chain waves in 16 launches of 1/16 (two cores)', [(6, 48, 3750, 16), (6, 48, 3750, 16)]),
    ('96 chain waves, integers only', [(8, 96, 60000, 1)]),
    ('96 chain waves, product by v_mad_u64_u32', [(9, 96, 60000, 1)]),
    ('96 chain waves, FP64 only (cvt, mul, cvt)', [(10, 96, 120000, 1)]),
    ('3048 streaming waves, 16 KB each, x12 launches', [(7, 3048, 512, 12)]),
    ('3048 streaming waves, 64 KB each, x6 launches', [(7, 3048, 2048, 6)]),
    ('96 chain waves + 3048 streaming waves x12', [(6, 96, 60000, 1), (7, 3048, 512, 12)]),
]
for (name, spec) in CASES:
    t = timed(spec)
    print('%-56s %.4f ms (%+.1f %%)' % (name, t, (t/base - 1.)*100.))
