#!/usr/bin/env python
"""bench.py -- Mpixels/s of the compression inference path (encode -> quantise -> entropy-code -> decode) on MI355X.

One STEP = one `codec.BatchCodec.submit`: one pass of the whole hot path over one batch of Kodak-sized (512x768) synthetic luminance images that
are already resident in HBM:
    conv1+GDN1 -> conv2+GDN2 -> conv3 -> [GDN3 -> centre/quantise/int16 symbols (+dead-map flags) -> IGDN4] (one kernel)
    -> exception-map histogram
    -> lossless coder ON THE DEVICE (UEG0 + binary arithmetic coder, one map per lane: encode + decode + compare, on its
       own stream, concurrent with the synthesis transforms); streams stay in HBM, per-map bit counts go to the host
    -> tconv1+IGDN5 -> tconv2+IGDN6 -> tconv3 + BT.601 cast + squared error vs the input (PSNR).
Nothing is skipped or cached between steps. Multi-GPU: one process per GPU, each rank codes its own batch (weak
scaling, no data-path collective); one RCCL all-reduce sums the rate / PSNR statistics at the end of the timed region.

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for the fields).
"""
import argparse
import gc
import json
import os
import sys
import time

# The HIP runtime multiplexes streams onto 4 hardware queues by default; streams that land on the same queue serialise.
# With a transform stream, 2-3 coder streams and optional extra transform streams that aliasing was measured to cost up to
# 30 % (and the one-image-per-step leg keeps 14 streams busy). Must be set before the runtime initialises; an explicit
# setting of the caller wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import numpy          # noqa: E402
import torch          # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from autoencoder_based_image_compression_amd import codec                    # noqa: E402
from autoencoder_based_image_compression_amd import device as dev            # noqa: E402
from autoencoder_based_image_compression_amd import pipeline                 # noqa: E402
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var   # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats   # noqa: E402
from autoencoder_based_image_compression_amd.kodak.tools import tools as tls   # noqa: E402

H_IN, W_IN = 512, 768          # Kodak luminance (datasets/kodak/kodak.py:10-83: uint8 (24, 512, 768))
IDX_MAP_EXCEPTION = 67         # lossless/results/1_10000/training_index_10/idx_map_exception.pkl
TRUNCATED_UNARY_LENGTH = 10    # collecting_stats_eae_extra.py:44
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup quota (the GPU box exposes 256 hardware
    threads behind a 16-CPU quota; OpenMP sized for 256 would only thrash)."""
    count = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            (quota, period) = f.read().split()
        if quota != 'max':
            count = min(count, max(1, int(int(quota)/int(period))))
    except (OSError, ValueError):
        pass
    return max(1, count)


def synthetic_images(seed, n, h, w):
    """RandomState(seed).randint(16, 236) low-pass filtered (3x box blur), uint8."""
    rng = numpy.random.RandomState(seed)
    x = rng.randint(16, 236, size=(n, h, w)).astype(numpy.float32)
    for _ in range(3):
        x = (x + numpy.roll(x, 1, 1) + numpy.roll(x, -1, 1) + numpy.roll(x, 1, 2) + numpy.roll(x, -1, 2))/numpy.float32(5.)
    return numpy.round(x).astype(numpy.uint8)


def synthetic_model(bin_width=1.):
    """Random-init weights of the fixed-bin-width architecture (no trained checkpoint exists in the reference mount)."""
    v = var.random_variables(bin_width, False, seed=0, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)   # leave the clip floor
    return v


def run_pipeline(args, batch, steps, warmup, device, world, rank, cores, tracing, variables, coder_streams=None,
                 transform_streams=1, use_graphs=False):
    """Builds the resident state for `batch` images per step (codec.BatchCodec: weights, tables, per-slot buffers), runs
    `warmup` untimed and `steps` timed steps, and returns what the report needs. Everything in here up to the first barrier
    is outside the timed region."""
    if world > 1:
        import torch.distributed as dist
    images = torch.from_numpy(synthetic_images(1000 + rank, batch, H_IN, W_IN)).to(device)
    map_size = (H_IN//16)*(W_IN//16)
    # statistics that feed the coder, from this build's own a26/a27 path on the first batch (lossless/stats.py:306, :13-68)
    encoder = pipeline.DeviceEncoder(variables, False, device)
    y0 = encoder(images)
    map_mean_host = dev.map_means(y0).cpu().numpy()
    probabilities = lossless_stats.compute_binary_probabilities(y0.cpu().numpy(), variables[var.BIN_WIDTHS_NAME], map_mean_host,
                                                                TRUNCATED_UNARY_LENGTH)
    del y0, encoder
    gemm_events = []            # (start, stop, launch name) around every conv_gemm launch of the timed region
    recording = [False]

    def timed_launch(name, fn):
        if not recording[0]:
            return fn()
        (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        a.record()
        out = fn()
        b.record()
        gemm_events.append((a, b, name))
        return out

    coder_mode = 'none' if os.environ.get('EAE_BENCH_NO_CODER') else args.coder      # 'none': diagnostic only
    coder_threads = args.coder_threads if args.coder_threads > 0 else max(1, cores//max(world, 1) - 2)
    the_codec = codec.BatchCodec(variables, False, variables[var.BIN_WIDTHS_NAME], map_mean_host, probabilities, IDX_MAP_EXCEPTION,
                                 batch, H_IN, W_IN, device=device, nb_in_flight=coder_streams or args.coder_streams,
                                 launch_hook=timed_launch, coder=coder_mode, host_coder_threads=coder_threads,
                                 nb_transform_streams=transform_streams, use_graphs=use_graphs)

    def barrier():
        if world > 1:
            if dist.get_backend() == 'nccl':
                dist.barrier(device_ids=[device.index])      # RCCL: name the device, no guess from the rank
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        the_codec.submit(images)
    the_codec.drain()

    # the launch thread allocates only short-lived wrappers: keep the cyclic collector (a 30 ms pause every ~75 steps) out of it
    gc.collect()
    gc.disable()
    recording[0] = True
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    tickets = []
    for _ in range(steps):
        if tracing:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_marks.append((time.perf_counter() - t0, ev))
        tickets.append(the_codec.submit(images))
    the_codec.drain()
    results = [t.result() for t in tickets]          # raises here if any map of any batch failed
    # the path's only exchange step: sum the rate / PSNR accumulators over ranks
    stats = torch.tensor([float(sum(int(r['nb_bits'].sum()) for r in results)), float(sum(int(r['sse'].sum()) for r in results)),
                          float(sum(int(r['nb_deads'].sum()) for r in results)), float(steps*batch)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    the_codec.close()
    return {'elapsed': elapsed, 'stats': stats, 'gemm_events': gemm_events, 'probabilities': probabilities,
            'map_mean_host': map_mean_host, 'host_coder': coder_mode == 'host', 'coder_threads': coder_threads,
            'step_marks': step_marks}


def main():
    global H_IN, W_IN
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=100)
    parser.add_argument('--warmup', type=int, default=10)
    parser.add_argument('--batch', type=int, default=24, help='images per GPU per step (default: the Kodak set)')
    parser.add_argument('--height', type=int, default=H_IN, help='image height (default 512: Kodak)')
    parser.add_argument('--width', type=int, default=W_IN, help='image width (default 768: Kodak); e.g. --height 256 --width 256 --batch 64 is one rank of BASELINE.json configs[3]')
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-single-image', action='store_true', help='skip the one-image-per-step side measurement')
    parser.add_argument('--coder', choices=('device', 'host'), default='device',
                        help='device: the coder kernels on side streams (default). host: one device -> host copy of the symbols '
                             'per batch and the host C-ABI coder on a thread pool (the shape BASELINE.json sketches)')
    parser.add_argument('--coder-threads', type=int, default=0, help='host coder threads (0 = usable CPUs - 2)')
    parser.add_argument('--coder-streams', type=int, default=int(os.environ.get('EAE_CODER_STREAMS', '2')),
                        help='batches whose entropy coding may be in flight at once (each on its own HIP stream)')
    parser.add_argument('--transform-streams', type=int, default=int(os.environ.get('EAE_TRANSFORM_STREAMS', '1')),
                        help='1 (default): the transforms of consecutive batches run back to back on one stream, so that the HIP '
                             'events around a launch time that kernel alone (the roofline figures). 2-3: consecutive batches '
                             'alternate between private streams and their kernels overlap (fills the tails: +6-8 %% whole-job '
                             'throughput with --coder-streams 3), but a launch then shares the GPU and its duration says little')
    parser.add_argument('--graphs', action='store_true',
                        help='replay one captured hipGraph per step instead of launching kernel by kernel (small batches: the '
                             'launch thread is the bottleneck there). No per-launch events, so no roofline figures')
    args = parser.parse_args()
    (H_IN, W_IN) = (args.height, args.width)

    # two Python threads share the GIL (kernel launches; the codec's result worker): hand it over quickly
    sys.setswitchinterval(1e-4)
    tracing = bool(os.environ.get('EAE_BENCH_TRACE'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the product path.')
    if os.environ.get('EAE_BENCH_SHARE_GPU'):      # test hook: several ranks on one GPU (gloo instead of RCCL)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo' if os.environ.get('EAE_BENCH_SHARE_GPU') else 'nccl', rank=rank, world_size=world)
    device = torch.device('cuda', local_rank)
    cores = usable_cpus()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))      # the oracle's OpenMP transforms (cpu_baseline only)

    variables = synthetic_model(1.)
    run = run_pipeline(args, args.batch, args.steps, args.warmup, device, world, rank, cores, tracing, variables,
                       transform_streams=args.transform_streams, use_graphs=args.graphs)
    (elapsed, stats, gemm_events, probabilities, map_mean_host) = (run['elapsed'], run['stats'], run['gemm_events'],
                                                                    run['probabilities'], run['map_mean_host'])
    (host_coder, coder_threads, step_marks) = (run['host_coder'], run['coder_threads'], run['step_marks'])

    # ---- derived figures (outside the timed region) ------------------------------------------------------------------
    pixels_per_step = args.batch*H_IN*W_IN
    total_pixels = pixels_per_step*args.steps*world
    value = total_pixels/elapsed/1e6
    nb_images_total = stats[3].item()
    bpp = stats[0].item()/(nb_images_total*H_IN*W_IN)
    mean_psnr = float(tls.psnr_from_sse(stats[1].item(), nb_images_total*H_IN*W_IN))   # PSNR of the pooled MSE
    flops = {'conv2_gdn2': pipeline.FLOP_PER_PIXEL['conv2_gdn2'], 'conv3': 2*1600,      # gdn_3 runs in the latent-stage kernel
             'tconv1_igdn5': pipeline.FLOP_PER_PIXEL['tconv1_igdn5'], 'tconv2_igdn6': pipeline.FLOP_PER_PIXEL['tconv2_igdn6']}
    per_launch_ms = {}
    for (a, b, name) in gemm_events:
        per_launch_ms.setdefault(name, []).append(a.elapsed_time(b))
    gemm_ms = sum(sum(v) for v in per_launch_ms.values())
    gemm_launches = sum(len(v) for v in per_launch_ms.values())
    gemm_flop = sum(flops[name]*pixels_per_step*len(v) for (name, v) in per_launch_ms.items())
    achieved = gemm_flop/(gemm_ms*1e-3)/1e12 if gemm_ms > 0 else 0.
    traffic = None
    traffic_file = os.path.join(ROOT, 'profiles', 'traffic_conv_gemm.json')
    if os.path.isfile(traffic_file):
        with open(traffic_file) as f:
            traffic = json.load(f).get('hbm_bytes_per_launch')
    line = {
        'metric': 'Mpixels/s encode+decode (Kodak 768x512 luma), bitstream bit-exact',
        'value': round(value, 3), 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed/args.steps*1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '{0}_{1}x{2}_luma_batch{3}_per_gpu_bin_width_1.0_lossless_roundtrip'.format(
                       'kodak' if (H_IN, W_IN) == (512, 768) else 'synthetic', H_IN, W_IN, args.batch),
                   'images_per_gpu_per_step': args.batch, 'height': H_IN, 'width': W_IN, 'bin_width_multiplier': 1.0,
                   'truncated_unary_length': TRUNCATED_UNARY_LENGTH, 'idx_map_exception': IDX_MAP_EXCEPTION,
                   'weights': 'random-init fixed-bin-width architecture (trained checkpoints absent from the reference)',
                   'parallelism': 'image shards, one process per GPU' if world > 1 else 'single GPU',
                   'coder': 'device, 64 maps per wavefront, encode + decode + compare' if not host_coder else
                            'host C-ABI coder, {} threads, after one device -> host copy of the symbols'.format(coder_threads)},
        'images_per_s': round(nb_images_total/elapsed, 2),
        'rate_bpp': round(bpp, 5), 'psnr_db_pooled': round(mean_psnr, 4), 'dead_maps_per_image': round(stats[2].item()/nb_images_total, 3),
        'roofline': {'bound': 'mfma', 'kernel': 'conv_gemm_wave_kernel (conv2+GDN2, conv3, tconv1+IGDN5, tconv2+IGDN6)',
                     'achieved': round(achieved, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved/PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic,
                     'avg_launch_ms': round(gemm_ms/max(gemm_launches, 1), 4),
                     'per_launch_ms': {k: round(sum(v)/len(v), 4) for (k, v) in per_launch_ms.items()},
                     'algorithmic_flop_per_launch': {k: flops[k]*pixels_per_step for k in flops}},
    }
    if step_marks:
        host = [round(m[0]*1e3, 2) for m in step_marks]
        gpu = [round(step_marks[0][1].elapsed_time(m[1]), 2) for m in step_marks]
        sys.stderr.write('TRACE host enqueue deltas (ms): {}\nTRACE gpu step deltas (ms): {}\nTRACE total ms {}\n'.format(
            [round(b - a, 1) for (a, b) in zip(host[:-1], host[1:])], [round(b - a, 1) for (a, b) in zip(gpu[:-1], gpu[1:])], round(elapsed*1e3, 2)))
    if rank == 0 and world == 1 and args.batch != 1 and not args.no_single_image and (H_IN, W_IN) == (512, 768):
        # BASELINE.json configs[1] is ONE Kodak image: the same path with one image per step (launch-bound, not the headline)
        del run, gemm_events
        one = run_pipeline(args, 1, 300, 30, device, world, rank, cores, False, variables, coder_streams=8, transform_streams=6,
                           use_graphs=True)
        line['single_image'] = {'ms_per_image': round(one['elapsed']/300*1e3, 4),
                                'mpixels_per_s': round(300*H_IN*W_IN/one['elapsed']/1e6, 2), 'steps': 300, 'warmup': 30,
                                'note': 'one 512x768 image per step; a step is a chain of short latency-bound kernels, so steps '
                                        'are pipelined: 6 transform streams, 8 coder streams, three hipGraph launches per step'}
    if rank == 0 and world == 1 and not args.no_single_image and args.transform_streams == 1 and not args.graphs:
        # the same batch in the opt-in overlapped mode (consecutive batches on alternating transform streams, three batches of
        # coder work in flight, a step replayed as three hipGraphs): whole-job rate only -- launches share the GPU, so there
        # are no per-launch durations to report
        over = run_pipeline(args, args.batch, 100, 15, device, world, rank, cores, False, variables, coder_streams=3,
                            transform_streams=2, use_graphs=True)
        line['overlapped_mode'] = {'value': round(100*args.batch*H_IN*W_IN/over['elapsed']/1e6, 3), 'unit': 'Mpixels/s',
                                   'ms_per_step': round(over['elapsed']/100*1e3, 4), 'steps': 100, 'warmup': 15,
                                   'flags': '--transform-streams 2 --coder-streams 3 --graphs'}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(variables, probabilities, map_mean_host, cores)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(variables, probabilities, map_mean, cores):
    """The same path on the host cores, on a BOUNDED sample (checker code, timed only here, never shipped):
    transforms = oracle/transforms_oracle.c (plain-C restatement, OpenMP over all cores);
    coder = the reference's own C++ coder compiled into oracle/_ref (single thread, as the reference runs it),
    falling back to the oracle's C restatement when the reference build is absent; numpy quantiser / PSNR.
    One image calibrates, then as many images as fit in about 20 s of CPU work (2..64) are timed together."""
    from oracle import coder as oracle_coder
    from oracle import transforms as oracle_transforms
    import ctypes
    try:        # libgomp was initialised when torch was imported: set the team size for this thread explicitly
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(int(cores))
    except OSError:
        pass
    bw = variables[var.BIN_WIDTHS_NAME]
    kind_coder = 'ref' if oracle_coder.available('ref') else 'oracle'
    lib = oracle_coder.CoderLib(kind_coder)

    def run(n_img):
        x = synthetic_images(999, n_img, H_IN, W_IN)
        t = {}
        t0 = time.perf_counter()
        y = oracle_transforms.encoder(x.astype(numpy.float32)[..., None], variables, False)
        t['encoder_oracle_c_openmp'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        tiled = numpy.tile(bw.reshape(1, 1, 1, 128), y.shape[:3] + (1,))
        cq = tiled*numpy.round((y - map_mean)/tiled)
        sym = numpy.round(cq/tiled).astype(numpy.int16)
        t['quantiser_numpy'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        bits = 0
        for j in range(n_img):
            for c in range(128):
                if c == IDX_MAP_EXCEPTION:
                    continue
                (rec, nb) = lib.compress_lossless(numpy.ascontiguousarray(sym[j, :, :, c]).reshape(-1), probabilities[c])
                bits += nb
        t['coder_{}_single_thread'.format('reference_cpp' if kind_coder == 'ref' else 'oracle_c')] = time.perf_counter() - t0
        t0 = time.perf_counter()
        rec = oracle_transforms.decoder(cq + map_mean, variables, False)[..., 0]
        rec_u8 = numpy.round(rec.clip(min=16., max=235.)).astype(numpy.uint8)
        mse = numpy.mean((x.astype(numpy.float64) - rec_u8.astype(numpy.float64))**2)
        t['decoder_oracle_c_openmp_plus_psnr'] = time.perf_counter() - t0
        return (t, bits, float(mse))

    (t1, _, _) = run(1)
    n_img = int(max(2, min(64, round(20./max(sum(t1.values()), 1e-3)))))
    (t, bits, mse) = run(n_img)
    total = sum(t.values())
    return {'value': round(n_img*H_IN*W_IN/total/1e6, 4), 'unit': 'Mpixels/s', 'cores': cores, 'kind': 'port',
            'sample': ('{0} synthetic {1}x{2} images, encode+quantise+code(enc+dec)+decode+PSNR, {3:.1f} s of CPU work; '
                       'transforms OpenMP on the {4} usable CPUs, coder single-threaded like the reference'
                       ).format(n_img, H_IN, W_IN, total, cores),
            'seconds': {k: round(val, 3) for (k, val) in t.items()}, 'bits': int(bits), 'mse': round(mse, 4)}


if __name__ == '__main__':
    main()
