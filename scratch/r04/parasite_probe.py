"""r04: conv_2 + GDN_2 of a Kodak batch (the longest launch of the step) timed alone and next to N sleeping one-wave blocks that hold
24 / 32 / 48 / 64 VGPRs each (scratch/r04/parasite.hip): what a resident coder wave costs the transforms by its registers alone."""
import ctypes, os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy, torch
import bench
from autoencoder_based_image_compression_amd import device as dev, pipeline
lib = ctypes.CDLL(os.path.join(ROOT, 'scratch', 'r04', 'libs', 'libparasite.so'))
lib.parasite_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
variables = bench.synthetic_model(1.0)
images = torch.from_numpy(bench.synthetic_images(1000, 24, 512, 768)).cuda()
enc = pipeline.DeviceEncoder(variables, False)
gdn_1 = dev.conv9x9s4_u8(images, enc.w1, enc.v['encoder/biases_1'], enc.g[1], enc.v['encoder/beta_1'])
out = torch.empty((24, 64, 96, 128), device='cuda')
ws = dev.conv_workspace('cuda')
sink = torch.zeros(16, device='cuda')
side = torch.cuda.Stream()


def conv2():
    dev.conv5x5s2(gdn_1, enc.w2, enc.v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], enc.v['encoder/beta_2'], out=out, workspace=ws)


def timed(vgprs, blocks):
    times = []
    for _ in range(12):
        torch.cuda.synchronize()
        if blocks:
            lib.parasite_launch(vgprs, blocks, 400000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))     # 4 ms at 100 MHz
            conv2()          # the sleepers settle while this one runs
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
base = timed(0, 0)
print('conv_2 + GDN_2, 24 Kodak images, alone: %.4f ms' % base)
for blocks in (48, 96, 256):
    for vgprs in (24, 32, 48, 64):
        t = timed(vgprs, blocks)
        print('next to %3d sleeping waves holding %2d VGPRs each: %.4f ms (%+.1f %%)' % (blocks, vgprs, t, (t/base - 1.)*100.))
