"""r06, VERDICT item 5: the coder's streams on a few CUs of their own (hipExtStreamCreateWithCUMask), the transform streams on the rest,
against the product mode (no masks), at three rates. One process per variant (CODER_CUS = 0 / 8 / 16 / 32 / 64; TRANSFORM = 'rest' or 'all').
The streams are made through the runtime directly (ctypes) and wrapped as torch.cuda.ExternalStream; everything else is the product's."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from autoencoder_based_image_compression_amd import codec
coder_cus = int(os.environ.get('CODER_CUS', '0'))
transform = os.environ.get('TRANSFORM', 'rest')
hip = ctypes.CDLL('libamdhip64.so')
kept = []
def masked_stream(device, bits):
    mask = (ctypes.c_uint32*8)(*bits)
    handle = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 8, mask)
    assert rc == 0, rc
    kept.append(handle)
    return torch.cuda.ExternalStream(handle.value, device=device)
real_side_streams = codec._side_streams
def side_streams(count, device, kind='coder'):
    if coder_cus == 0 or (kind == 'transform' and transform == 'all'):
        return real_side_streams(count, device, kind)
    streams = codec._SIDE_STREAMS.setdefault((device.index, kind), [])
    low = [0]*8
    for i in range(coder_cus):
        low[i//32] |= 1 << (i % 32)
    bits = low if kind == 'coder' else [(~w) & 0xFFFFFFFF for w in low]
    while len(streams) < count:
        with torch.cuda.device(device):
            streams.append(masked_stream(device, bits))
    return streams[:count]
codec._side_streams = side_streams
args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
mode = codec.product_mode(512, 768)
for width in (1.0, 0.125, 0.05):
    v = bench.synthetic_model(width)
    leg = bench.run_pipeline(ctx, 24, 20, 5, v, 512, 768, coder_streams=mode['nb_in_flight'], transform_streams=mode['nb_transform_streams'],
                             use_graphs=True, min_seconds=1.0, max_blocks=25, given_statistics=True)
    (bpp, _) = bench.rate_and_psnr(leg['stats'], 512, 768)
    print('coder CUs %3d transform %-4s bin width %-6s %.3f bpp: %8.1f Mpx/s  %.4f ms/step  (%d blocks, min %.4f)' % (
        coder_cus, transform if coder_cus else '-', width, bpp, 24*512*768*20/leg['elapsed']/1e6, leg['elapsed']/20*1e3, len(leg['block_seconds']), min(leg['block_seconds'])/20*1e3))
    sys.stdout.flush()
