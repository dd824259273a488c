"""The coder round trip of ONE Kodak image's maps on an idle GPU: encode_batch + decode_batch(expected) against the chunked round trip
(coder_roundtrip_trailing), launched directly (streams + events) and replayed as a hipGraph. Wall time per call, synchronised."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy          # noqa: E402
import torch          # noqa: E402
import bench          # noqa: E402
from autoencoder_based_image_compression_amd import device as dev, pipeline          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var          # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats          # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    t.sort()
    return t[len(t)//2]*1e3


def main():
    torch.cuda.set_device(0)
    for width in (1.0, 0.05):
        v = bench.synthetic_model(width)
        images = torch.from_numpy(bench.synthetic_images(1000, 1, 512, 768)).cuda()
        y = pipeline.DeviceEncoder(v, False, 'cuda')(images)
        mean = dev.map_means(y)
        bw = v[var.BIN_WIDTHS_NAME]
        prob = torch.from_numpy(lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bw, mean.cpu().numpy(), 10)).cuda()
        q = dev.quantize_maps(y, torch.from_numpy(bw).cuda(), mean, want_symbols=True)
        symbols = q['symbols'].reshape(128, -1)
        rows = torch.arange(128, dtype=torch.int32)
        rows[67] = -1
        rows = rows.cuda()
        streams = dev.CoderStreams(128, symbols.shape[1], 10, 'cuda')
        ws = dev.coder_trailing_workspace(128, symbols.shape[1], 10, 'cuda')

        def two():
            dev.coder_encode_batch(symbols, prob, rows, 10, out=streams, workspace=ws)
            dev.coder_decode_batch(streams, prob, rows, expected=symbols, workspace=ws)
        print('bin width {0}: two calls                 {1:.4f} ms'.format(width, timed(two)))
        def fused():
            dev.coder_roundtrip_fused(symbols, prob, rows, 10, out=streams, workspace=ws)
        print('bin width {0}: fused (one workgroup per 64 maps) {1:.4f} ms'.format(width, timed(fused)))
        assert not streams.status.cpu().numpy().any()
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
            fused()
        print('bin width {0}: fused as a hipGraph          {1:.4f} ms'.format(width, timed(graph.replay)))
        for chunks in (4, 8):
            def trailing():
                dev.coder_roundtrip_trailing(symbols, prob, rows, 10, chunks=chunks, out=streams, workspace=ws)
            direct = timed(trailing)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                trailing()
            replay = timed(graph.replay)
            print('bin width {0}: trailing, {1:2d} chunks: direct {2:.4f} ms, as a hipGraph {3:.4f} ms'.format(width, chunks, direct, replay))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
            two()
        print('bin width {0}: two calls as a hipGraph   {1:.4f} ms'.format(width, timed(graph.replay)))
        assert not streams.status.cpu().numpy().any()


if __name__ == '__main__':
    main()
