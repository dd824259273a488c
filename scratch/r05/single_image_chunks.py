"""One Kodak image per step with the coder's chains cut into 1 / 3 / 4 / 6 / 8 launches (codec.BatchCodec(coder_chunks=...)): latency of a
step on an idle GPU (submit -> result, one at a time) and the pipelined rate (6 transform streams, 8 coder batches in flight)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import bench          # noqa: E402
import torch          # noqa: E402


def main():
    args = bench.parse_args(['--no-cpu-baseline'])
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
    variables = bench.synthetic_model(1.0)
    for width in (1.0, 0.05):
        v = bench.synthetic_model(width)
        for chunks in (1, 3, 4, 6, 8):
            os.environ['EAE_CODER_CHUNKS'] = str(chunks)
            alone = bench.run_pipeline(ctx, 1, 100, 10, v, 512, 768, coder_streams=1, transform_streams=1, use_graphs=True, serial=True)
            one = bench.run_pipeline(ctx, 1, 300, 30, v, 512, 768, coder_streams=8, transform_streams=6, use_graphs=True)
            print(json.dumps({'bin_width': width, 'coder_chunks': chunks, 'latency_ms': round(alone['elapsed']/100*1e3, 4),
                              'pipelined_ms_per_image': round(one['elapsed']/300*1e3, 4),
                              'rate_bpp': round(bench.rate_and_psnr(one['stats'], 512, 768)[0], 4)}))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
