"""r05: how the cost of chain waves to a conv_2 launch grows with their number (r04 measured 48 ... 1024: all +33 %), under the GEMM
forms (EAE_HIP_SPLIT_WPB=1: one-wave blocks; EAE_HIP_GEMM=u: whole tiles). Run once per form (the form is read at library load)."""
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
big = torch.zeros(1 << 27, dtype=torch.float32, device='cuda')
side = torch.cuda.Stream()


def conv2():
    dev.conv5x5s2(gdn_1, enc.w2, enc.v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], enc.v['encoder/beta_2'], out=out, workspace=ws)


def timed(blocks, launches=3):
    times = []
    for _ in range(10):
        torch.cuda.synchronize()
        if blocks:
            lk.lk_launch(6, blocks, 60000*launches//3, ctypes.c_void_p(big.data_ptr()), big.numel()*4, ctypes.c_void_p(side.cuda_stream))
        conv2()
        (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        a.record()
        for _ in range(launches):
            conv2()
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b)/launches)
    return float(numpy.median(times))


for _ in range(5):
    conv2()
base = timed(0)
print('form %s wpb %s: conv_2 + GDN_2 alone %.4f ms' % (os.environ.get('EAE_HIP_GEMM', '-'), os.environ.get('EAE_HIP_SPLIT_WPB', '-'), base))
for blocks in (1, 2, 4, 8, 16, 24, 48, 128, 512, 1024, 2048):
    t = timed(blocks)
    print('  %5d chain waves: %.4f ms (%+.1f %%)' % (blocks, t, (t/base - 1.)*100.))
