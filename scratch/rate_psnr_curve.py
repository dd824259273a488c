"""BASELINE.json configs[2]: the 24-image Kodak-shaped set at bin-width multipliers {0.5, 1.0, 2.0}: rate (bits per pixel of
the lossless code, exception map charged its entropy like compression.py:68-75) and PSNR per image from the MI355X path
(codec.BatchCodec) next to the CPU evaluation (oracle/transforms_oracle.c for the transforms, the reference's own C++
coder from oracle/_ref, numpy for the rest). Writes profiles/rate_psnr_curve.json. Checker use of oracle/: this is a
test script, not product code."""
import json, os, sys, time
import numpy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from autoencoder_based_image_compression_amd import codec, device as dev, pipeline
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var
from autoencoder_based_image_compression_amd.kodak.lossless import compression, stats as lossless_stats
from oracle import coder as oracle_coder, transforms as T

(H, W, N, L, IDX) = (512, 768, 24, 10, 67)
variables = bench.synthetic_model(1.)
images = bench.synthetic_images(1000, N, H, W)
bw = variables[var.BIN_WIDTHS_NAME]
enc = pipeline.DeviceEncoder(variables, False)
y_dev = enc(torch.from_numpy(images).cuda())
map_mean = dev.map_means(y_dev).cpu().numpy()
y_gpu = y_dev.cpu().numpy()
t0 = time.time()
y_cpu = T.encoder(images.astype(numpy.float32)[..., None], variables, False)
t_enc = time.time() - t0
assert numpy.array_equal(y_cpu, y_gpu)
lib = oracle_coder.CoderLib('ref' if oracle_coder.available('ref') else 'oracle')
out = {'images': N, 'height': H, 'width': W, 'cpu_encoder_s': round(t_enc, 2), 'points': []}
for m in (0.5, 1.0, 2.0):
    bwt = (numpy.float32(m)*bw).astype(numpy.float32)
    probs = lossless_stats.compute_binary_probabilities(y_gpu, bwt, map_mean, L)
    c = codec.BatchCodec(variables, False, bwt, map_mean, probs, IDX, N, H, W)
    t0 = time.time()
    g = c.submit(torch.from_numpy(images).cuda()).result()
    t_gpu = time.time() - t0
    c.close()
    # CPU
    t0 = time.time()
    tiled = numpy.tile(bwt.reshape(1, 1, 1, 128), y_cpu.shape[:3] + (1,))
    cq = tiled*numpy.round((y_cpu - map_mean)/tiled)
    sym = numpy.round(cq/tiled).astype(numpy.int16)
    rec = T.decoder(cq + map_mean, variables, False)[..., 0]
    rec_u8 = numpy.round(rec.clip(min=16., max=235.)).astype(numpy.uint8)
    bits = numpy.zeros(N, dtype=numpy.int64)
    for j in range(N):
        for ch in range(128):
            flat = numpy.ascontiguousarray(sym[j, :, :, ch]).reshape(-1)
            if ch == IDX:
                counts = numpy.bincount(flat.astype(numpy.int64) + 32768)
                bits[j] += int(compression.exception_map_nb_bits(counts, flat.size))
            else:
                bits[j] += lib.compress_lossless(flat, probs[ch])[1]
    sse = ((images.astype(numpy.int64) - rec_u8.astype(numpy.int64))**2).reshape(N, -1).sum(axis=1)
    t_cpu = time.time() - t0
    same = bool(numpy.array_equal(bits, g['nb_bits']) and numpy.array_equal(sse, g['sse']))
    psnr = 10.*numpy.log10(255.**2/(sse/float(H*W)))
    out['points'].append({'multiplier': m, 'rate_bpp_mean': float(bits.mean()/(H*W)), 'psnr_db_mean': float(psnr.mean()),
                          'gpu_equals_cpu_bits_and_sse_for_all_images': same, 'gpu_wall_s_cold': round(t_gpu, 3), 'cpu_wall_s': round(t_cpu, 1),
                          'rate_bpp_per_image': [round(float(b)/(H*W), 5) for b in bits], 'psnr_db_per_image': [round(float(p), 4) for p in psnr]})
    print(out['points'][-1]['multiplier'], out['points'][-1]['rate_bpp_mean'], out['points'][-1]['psnr_db_mean'], same, t_gpu, t_cpu)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'rate_psnr_curve.json'), 'w'), indent=1)
